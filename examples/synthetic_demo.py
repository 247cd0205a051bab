#!/usr/bin/env python
"""End-to-end walk through the public API on synthetic data (no reference checkout, no CONCH weights needed):

  1. slides on the host (fp32 `.pt`-like tensors, several per patient)  ->  resident bf16 arena in HBM (one upload)
  2. the text side: ordinal rank prompts (`RankPromptLearner`) through a (small, random-weight) CoCa-style text tower
     (`CONCHPromptEncoder`, HIP forward + backward) -> the K text features; evaluation: 32 patients per launch through
     `VLSA.forward_bags` (persistent multi-bag HIP kernels)
  3. a few optimizer steps through `TrainStep`: text tower once per step, batched HIP forward + backward of the aggregation, IF-MLE +
     EMD objective in one kernel, one-launch Adam on the context / rank embeddings, the text queries' residual and the logit scale; a
     batch that repeats is replayed as one hipGraph
  4. interpretation of one slide (`calc_text_img_similarity`)

    python examples/synthetic_demo.py [--patients 64] [--steps 5]
"""
import argparse
import os
import re
import sys
import time
import zlib

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from vlsa_amd.inference import calc_text_img_similarity  # noqa: E402
from vlsa_amd.ingest import ArenaLayout, DeviceBagArena  # noqa: E402
from vlsa_amd.losses import SurvObjective  # noqa: E402
from vlsa_amd.optim import FusedAdam  # noqa: E402
from vlsa_amd.prompt_adapter import PromptAdapter  # noqa: E402
from vlsa_amd.prompt_encoder import CONCHPromptEncoder  # noqa: E402
from vlsa_amd.prompt_learner import RankPromptLearner  # noqa: E402
from vlsa_amd.train_step import TrainStep  # noqa: E402
from vlsa_amd.vlsa import VLSA  # noqa: E402


class ToyTokenizer:
    """Word / punctuation tokenizer with hashed ids -- stands in for the CONCH tokenizer wrapper (model/utils_vl.py:19-75) with
    the same call contract: a full row is <bos> ids.. <eos> <pad>.. ; 'raw' rows drop <bos> and stop at the longest sentence."""
    bos_token_id, eos_token_id, pad_token_id = 1, 2, 0

    def __init__(self, vocab, length=128):
        self.vocab, self.length = vocab, length

    def __call__(self, text, return_raw_tokens=True, return_num_tokens=True):
        texts = [text] if isinstance(text, str) else list(text)
        ids = [[3 + zlib.crc32(w.lower().encode()) % (self.vocab - 3) for w in re.findall(r"\w+|[^\w\s]", t)] for t in texts]
        rows = torch.full((len(texts), self.length), self.pad_token_id, dtype=torch.long)
        for i, r in enumerate(ids):
            rows[i, 0] = self.bos_token_id
            rows[i, 1:1 + len(r)] = torch.tensor(r, dtype=torch.long)
            rows[i, 1 + len(r)] = self.eos_token_id
        cnt = torch.tensor([len(r) for r in ids])
        out = rows[:, 1:int(cnt.max()) + 1] if return_raw_tokens else rows
        if isinstance(text, str):
            out, cnt = out[0], cnt[0]
        return (out, cnt) if return_num_tokens else out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--patients", type=int, default=64)
    ap.add_argument("--steps", type=int, default=8)
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
    g = torch.Generator().manual_seed(0)
    P, K = 12, 4

    # 1. ingest -------------------------------------------------------------------------------------------------------
    patients = {}
    for pid in range(a.patients):
        n_slides = 1 + pid % 3
        patients[pid] = [torch.randn(int(torch.randint(500, 6000, (1,), generator=g)), 512, generator=g) for _ in range(n_slides)]
    rows = {pid: sum(s.shape[0] for s in slides) for pid, slides in patients.items()}
    arena = DeviceBagArena(ArenaLayout.rows_needed(rows.values()), dev)
    t0 = time.perf_counter()
    for pid, slides in patients.items():
        arena.add(pid, slides)                     # multi-slide patients are packed back to back: no host concat
    arena.wait()
    print(f"[ingest] {len(patients)} patients, {sum(rows.values())} patches -> {arena.data.numel() * 2 / 1e6:.0f} MB bf16 arena "
          f"in {(time.perf_counter() - t0) * 1e3:.1f} ms")

    # 2. model + evaluation -------------------------------------------------------------------------------------------
    torch.manual_seed(0)
    tower = CONCHPromptEncoder(width=256, heads=4, layers=3, vocab_size=2000, output_dim=512).to(dev)   # random weights; CONCH: 768 / 12 / 12
    for p_ in tower.parameters():
        p_.requires_grad_(False)                                           # vlsa_txt_encoder_frozen: True (cfg_vlsa_conch.yaml:69)
    learner = RankPromptLearner(dict(max_num_tokens=127, embedding_dim=256, embedding_dtype=torch.float32), ToyTokenizer(2000),
                                tower.token_embedding, num_base_ranks=2, num_ranks=K, num_tokens_per_rank=2, num_context_tokens=8,
                                init_context="a histopathology slide of a patient whose survival is",
                                init_rank_names=["very short", "very long"])
    prompt_features = torch.randn(P, 512, generator=g)                     # frozen text prototypes of the PromptAdapter
    cfg = dict(name="VLFAN", dim_in=512, use_feat_proj=False, num_query=P, query="Text", query_pooling="mean", pred_head="default")
    qnet = PromptAdapter(method="TaskRes", num_prompts=P, pretrained_prompt_features=prompt_features, res_ratio=0.5)
    net = VLSA.from_modules(cfg, prompt_learner=learner, prompt_encoder=tower, query_network=qnet).to(dev)
    with torch.no_grad():
        tf = net.forward_text_only()
    print(f"[text]   {K} rank prompts x {int(learner.pseudo_sentence_tokens[0].max())} tokens -> text features {tuple(tf.shape)} "
          f"(|t| = {tf.norm(dim=-1).mean().item():.2f}); cached until a prompt parameter changes")
    pids = list(patients)
    net.eval()
    with torch.no_grad():                                                  # first pass: library load, plans, clock ramp
        torch.cat([net.forward_bags(bags)[0] for bags in arena.batches(pids, 32)])
    torch.cuda.synchronize(); t0 = time.perf_counter()
    with torch.no_grad():
        logits = torch.cat([net.forward_bags(bags)[0] for bags in arena.batches(pids, 32)])
    torch.cuda.synchronize()
    print(f"[eval]   logits {tuple(logits.shape)} in {(time.perf_counter() - t0) * 1e3:.2f} ms "
          f"({sum(rows.values()) / (time.perf_counter() - t0) / 1e6:.0f} M patches/s incl. host)")

    # 3. training steps -----------------------------------------------------------------------------------------------
    t_bin = torch.randint(0, K, (len(pids),), generator=g).to(dev)
    event = (torch.rand(len(pids), generator=g) < 0.5).float().to(dev)
    objective = SurvObjective()
    trainable = [p for p in net.parameters() if p.requires_grad]
    # the handler's optimizer (optim_factory.py:25-37: no weight decay on 1-D parameters) as ONE launch; the step itself owned by TrainStep:
    # a batch seen for the third time is replayed as one hipGraph (forward_bags, objective, backward, Adam)
    opt = FusedAdam([{"params": [p for p in trainable if p.dim() < 2], "weight_decay": 0.0},
                     {"params": [p for p in trainable if p.dim() >= 2], "weight_decay": 1e-5}], lr=2e-4)
    stepper = TrainStep(net, objective, opt)
    net.train()
    groups = [torch.randperm(len(pids), generator=g)[:32] for _ in range(2)]          # two fixed 32-patient batches, visited in turn
    batches = [([arena.bag(pids[int(i)]) for i in idx], t_bin[idx.to(dev)], event[idx.to(dev)]) for idx in groups]
    for step in range(a.steps):
        bags, tb, ev = batches[step % 2]
        loss = stepper.step(bags, tb, ev)
        print(f"[train]  step {step}: loss {loss.item():.4f}")
    print(f"[train]  {stepper.describe()}")

    # 4. interpretation -----------------------------------------------------------------------------------------------
    net.eval()
    _, A, cottn, probs, probs2, dec, shap = calc_text_img_similarity(net, arena.bag(pids[0])[None])
    print(f"[interp] attention {tuple(A.shape)}, incidence {probs.numpy().round(3).tolist()}, prototype SHAP sum {shap.sum().item():+.4f}")
    print("demo ok")


if __name__ == "__main__":
    main()
