"""One optimizer step of the reference's training loop on the GPU, end to end (runner/vlsa_handler.py:260-289 with
cfg_vlsa_conch.yaml: 32 bags per step, VLFAN encoder with TaskRes text queries, ordinal RANK PROMPT LEARNER through the CONCH-size
text tower, IF-MLE + EMD loss, Adam): text side (once per step) + 32-bag batched aggregation forward + backward + loss + optimizer.
Bag sizes: TCGA-like (2k-12k patches, bf16 resident) and the north-star 50k."""
import os
import sys
import time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import torch
import text_cases as TC
from vlsa_amd.losses import SurvObjective
from vlsa_amd.prompt_adapter import PromptAdapter
from vlsa_amd.prompt_encoder import CONCHPromptEncoder
from vlsa_amd.prompt_learner import RankPromptLearner
from vlsa_amd.vlsa import VLSA

dev = "cuda"
import gc; gc.collect(); gc.freeze()   # torch's ~10^6 imported objects out of the collector's way: a gen-2 pass otherwise stalls one call by ~40 ms (profiles/README.md)
K = P = 12
c = TC.TOWERS["conch"]
enc = CONCHPromptEncoder(width=c["width"], heads=c["heads"], layers=c["layers"], vocab_size=c["vocab"], output_dim=c["out_dim"])
for p_ in enc.parameters():
    p_.requires_grad_(False)
enc = enc.to(dev)
table, ctx_key, names = TC.synthetic_prompt_table(c["vocab"], 1)
pl = RankPromptLearner(dict(max_num_tokens=127, embedding_dim=768, embedding_dtype=torch.float32), TC.ReplayTokenizer(table),
                       enc.token_embedding, num_base_ranks=4, num_ranks=K, num_tokens_per_rank=4, num_context_tokens=8,
                       init_context=ctx_key, init_rank_names=names)
qnet = PromptAdapter(method="TaskRes", num_prompts=P, pretrained_prompt_features=torch.randn(P, 512), res_ratio=0.5)
cfg = dict(name="VLFAN", dim_in=512, use_feat_proj=False, num_query=P, query="Text", query_pooling="mean", pred_head="default")
net = VLSA.from_modules(cfg, prompt_learner=pl, prompt_encoder=enc, query_network=qnet).to(dev).train()
params = [p_ for p_ in net.parameters() if p_.requires_grad]
opt = torch.optim.Adam(params, lr=2e-4)
objective = SurvObjective()
g = torch.Generator().manual_seed(0)
for label, sizes in (("TCGA-like 2k-12k", [int(x) for x in torch.randint(2000, 12000, (32,), generator=g)]), ("50k", [50000] * 32)):
    bags = [torch.randn(n, 512, device=dev).to(torch.bfloat16) for n in sizes]
    t = torch.randint(0, K, (32,), device=dev)
    e = (torch.rand(32, device=dev) < 0.45).float()

    def step():
        logits = net.forward_bags(bags)[0]
        loss = objective(logits, t, e, net.get_logit_scale())
        opt.zero_grad(set_to_none=True)
        loss.backward()
        opt.step()
        return loss
    for _ in range(10):
        step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    R = 30
    for _ in range(R):
        step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / R

    if "--only-batched" in sys.argv:      # (for rocprofv3: nothing but the batched step in the trace)
        print(f"{label}: {dt * 1e3:.2f} ms per optimizer step")
        continue
    # the same step over a checked BagSet (vlsa_amd.functional.BagSet: per-bag validation and descriptor rows once per split);
    # every step takes its own 32 bags out of the set, as a training loop over a resident split does
    from vlsa_amd.functional import BagSet
    pool = BagSet(bags)
    gperm = torch.Generator().manual_seed(1)

    def step_set():
        logits = net.forward_bags(pool.take(torch.randperm(32, generator=gperm).tolist()))[0]
        loss = objective(logits, t, e, net.get_logit_scale())
        opt.zero_grad(set_to_none=True)
        loss.backward()
        opt.step()
        return loss
    for _ in range(10):
        step_set()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(R):
        step_set()
    torch.cuda.synchronize()
    dts = (time.perf_counter() - t0) / R
    th0 = time.perf_counter()
    for _ in range(R):
        step_set()
    th_set = (time.perf_counter() - th0) / R          # host side alone (no synchronisation inside)
    torch.cuda.synchronize()

    # ... and with torch's single-kernel Adam (`fused=True`): the default foreach implementation is 7 multi-tensor launches of 8-21 us
    # each for this handful of small parameters (profiles/r04_step_kernel_stats.csv) -- the reference's handler builds its optimizer itself
    # (runner/vlsa_handler.py), so this is a hint for its config, not something the drop-in can change
    opt_f = torch.optim.Adam(params, lr=2e-4, fused=True)

    def step_fused():
        logits = net.forward_bags(bags)[0]
        loss = objective(logits, t, e, net.get_logit_scale())
        opt_f.zero_grad(set_to_none=True)
        loss.backward()
        opt_f.step()
        return loss
    for _ in range(10):
        step_fused()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(R):
        step_fused()
    torch.cuda.synchronize()
    dtf = (time.perf_counter() - t0) / R

    def text_only():
        f = net.prompt_encoder(prompts_embedding=pl(), prompts_pseudo_tokens=pl.pseudo_sentence_tokens, shared_prefix_len=pl.shared_prefix_len)   # as VLSA calls it
        f.sum().backward()
    for _ in range(5):
        text_only()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20):
        text_only()
    torch.cuda.synchronize()
    tt = (time.perf_counter() - t0) / 20
    npatch = sum(sizes)

    # the handler's own loop shape (runner/vlsa_handler.py:267-289): one net(X) per bag, cat, loss, ONE backward
    def hstep():
        logits = torch.cat([net(x[None])[0] for x in bags], dim=0)
        loss = objective(logits, t, e, net.get_logit_scale())
        opt.zero_grad(set_to_none=True)
        loss.backward()
        opt.step()
        return loss
    for _ in range(5):
        hstep()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(R):
        hstep()
    torch.cuda.synchronize()
    dth = (time.perf_counter() - t0) / R
    print(f"{label}: handler-shaped step (32 x net(X), cat, one backward): {dth * 1e3:.2f} ms")
    net.defer_training_calls = True          # vlsa_amd/deferred.py: the same loop, its calls served by ONE forward_bags at torch.cat
    for _ in range(5):
        hstep()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(R):
        hstep()
    torch.cuda.synchronize()
    dthd = (time.perf_counter() - t0) / R
    net.defer_training_calls = False
    print(f"{label}: the same handler-shaped loop with net.defer_training_calls = True (what patch_reference() sets): {dthd * 1e3:.2f} ms")
    if "--profile-handler" in sys.argv:
        import cProfile, pstats
        pr = cProfile.Profile(); pr.enable()
        for _ in range(10):
            hstep()
        pr.disable(); torch.cuda.synchronize()
        pstats.Stats(pr).sort_stats("tottime").print_stats(40)
    if "--profile" in sys.argv:       # host side of one step (the GPU work of a small-bag step is ~2.4 ms: is the host the limit?)
        import cProfile, pstats
        t0 = time.perf_counter()
        for _ in range(R):
            step()
        th = (time.perf_counter() - t0) / R
        torch.cuda.synchronize()
        print(f"host-only per step {th * 1e3:.2f} ms")
        pr = cProfile.Profile(); pr.enable()
        for _ in range(20):
            step()
        pr.disable(); torch.cuda.synchronize()
        pstats.Stats(pr).sort_stats("tottime").print_stats(30)
    print(f"{label}: {npatch} patches in 32 bags: {dt * 1e3:.2f} ms per optimizer step ({npatch / dt / 1e9:.2f} G patches/s trained), of which the "
          f"text side (rank prompts -> CONCH-size tower, forward + backward) {tt * 1e3:.2f} ms; the reference runs the tower 32x per step on top of "
          f"the bag path (1.44 s per call on its CPU path, BASELINE.md)")
    print(f"{label}: the same step over a BagSet (bags checked once): {dts * 1e3:.2f} ms per optimizer step, host side alone {th_set * 1e3:.2f} ms")
    print(f"{label}: the same step with torch.optim.Adam(fused=True): {dtf * 1e3:.2f} ms per optimizer step")
