"""Once-per-step cost of the text side on the GPU (rank prompts -> CoCa text tower, CONCH size: 12 layers x 768, K = 12
prompts of 11 tokens) against the reference's 1.44 s per call on the CPU: forward (inference), forward +
backward (training), and the CPU oracle on this host for reference."""
import os
import sys
import time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import torch
import text_cases as TC
import text_helpers as TH
from test_text_modules_cpu import build_learner
from test_gpu_text_tower import build_encoder
import gc; gc.collect(); gc.freeze()   # torch's ~10^6 imported objects out of the collector's way: a gen-2 pass otherwise stalls one call by ~40 ms (profiles/README.md)

case = TC.RANK_CASES[0]
inp = TH.rank_case_inputs(case)
enc = build_encoder(case[1], case[2])
pl = build_learner(case, inp).cuda()
PREFIX = 0 if "--no-prefix" in sys.argv else pl.shared_prefix_len      # rows of the sentences' shared prefix evaluated once


def timed(fn, n=50, warm=10):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n


def fwd():
    with torch.no_grad():
        return enc(prompts_embedding=pl(), prompts_pseudo_tokens=pl.pseudo_sentence_tokens, shared_prefix_len=PREFIX)


def fwd_bwd():
    pl.zero_grad(set_to_none=True)
    f = enc(prompts_embedding=pl(), prompts_pseudo_tokens=pl.pseudo_sentence_tokens, shared_prefix_len=PREFIX)
    f.sum().backward()


t_f, t_fb = timed(fwd), timed(fwd_bwd)
with torch.no_grad():
    sent = pl()
t_tower = timed(lambda: enc(prompts_embedding=sent, prompts_pseudo_tokens=pl.pseudo_sentence_tokens, shared_prefix_len=PREFIX))
rows = enc._plan(pl.pseudo_sentence_tokens, sent.device, PREFIX).M
print(f"K=12 rank prompts, CONCH-size tower (12 x 768, 85 M weights fp32), {rows} compact rows of 1536 (shared prefix: {PREFIX} positions)")
print(f"GPU forward (learner + tower, no grad): {t_f * 1e6:.0f} us;  tower alone: {t_tower * 1e6:.0f} us;  forward + backward: {t_fb * 1e6:.0f} us")
if "--cpu" in sys.argv:
    from oracle import text_oracle as TO
    torch.set_num_threads(min(32, os.cpu_count() or 1))
    t0 = time.perf_counter(); TH.oracle_rank_case(case); t1 = time.perf_counter() - t0
    print(f"CPU oracle (full 128 positions, torch {torch.get_num_threads()} threads): {t1 * 1e3:.0f} ms per call; the reference runs this once per BAG (BASELINE.md: 1.44 s on 8 cores)")
