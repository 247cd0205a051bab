#!/usr/bin/env python
"""End-to-end walk through the public API on synthetic data (no reference checkout, no CONCH weights needed):

  1. slides on the host (fp32 `.pt`-like tensors, several per patient)  ->  resident bf16 arena in HBM (one upload)
  2. evaluation: 32 patients per launch through `VLSA.forward_bags` (persistent multi-bag HIP kernels)
  3. a few optimizer steps: batched HIP forward + backward of the aggregation, IF-MLE + EMD loss in one kernel, Adam
  4. interpretation of one slide (`calc_text_img_similarity`)

    python examples/synthetic_demo.py [--patients 64] [--steps 5]
"""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from vlsa_amd.inference import calc_text_img_similarity  # noqa: E402
from vlsa_amd.ingest import ArenaLayout, DeviceBagArena  # noqa: E402
from vlsa_amd.losses import SurvObjective  # noqa: E402
from vlsa_amd.prompt_adapter import PromptAdapter  # noqa: E402
from vlsa_amd.vlsa import VLSA  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--patients", type=int, default=64)
    ap.add_argument("--steps", type=int, default=5)
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
    text_features = torch.randn(K, 512, generator=g)                       # stands in for the CONCH prompt encoder's output
    prompt_features = torch.randn(P, 512, generator=g)                     # frozen text prototypes of the PromptAdapter
    cfg = dict(name="VLFAN", dim_in=512, use_feat_proj=False, num_query=P, query="Text", query_pooling="mean", pred_head="default")
    qnet = PromptAdapter(method="TaskRes", num_prompts=P, pretrained_prompt_features=prompt_features, res_ratio=0.5)
    net = VLSA(cfg, pretrained_text_features=text_features, query_network=qnet).to(dev)
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
    opt = torch.optim.Adam([p for p in net.parameters() if p.requires_grad], lr=2e-4)
    net.train()
    for step in range(a.steps):
        idx = torch.randperm(len(pids), generator=g)[:32].tolist()
        out = net.forward_bags([arena.bag(pids[i]) for i in idx])[0]      # batched HIP forward, saved for the batched backward
        loss = objective(out, t_bin[idx], event[idx], net.get_logit_scale())
        opt.zero_grad()
        loss.backward()
        opt.step()
        print(f"[train]  step {step}: loss {loss.item():.4f}")

    # 4. interpretation -----------------------------------------------------------------------------------------------
    net.eval()
    _, A, cottn, probs, probs2, dec, shap = calc_text_img_similarity(net, arena.bag(pids[0])[None])
    print(f"[interp] attention {tuple(A.shape)}, incidence {probs.numpy().round(3).tolist()}, prototype SHAP sum {shap.sum().item():+.4f}")
    print("demo ok")


if __name__ == "__main__":
    main()
