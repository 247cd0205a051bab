"""Scores alone (vlsa_gated_scores), scores + pooling in one launch (vlsa_gated_scores_pool) and the two-launch route (scores, then
vlsa_scored_pool) on rotating bf16 bags: us per bag."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vlsa_amd import functional as F
dev = "cuda"
import gc; gc.collect(); gc.freeze()
def t(fn, n=100):
    for i in range(30): fn(i)
    torch.cuda.synchronize()
    us = 1e30
    for _ in range(4):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(n): fn(i)
        e1.record(); torch.cuda.synchronize()
        us = min(us, e0.elapsed_time(e1) * 1e3 / n)
    return us
for gated in (True, False):
    Wa = torch.randn(256, 512, device=dev) / 22; ba = torch.randn(256, device=dev) * 0.05
    Wg = torch.randn(256, 512, device=dev) / 22 if gated else None; bg = torch.randn(256, device=dev) * 0.05 if gated else None
    w2 = torch.randn(1, 256, device=dev) / 16; c = torch.randn(1, device=dev)
    W = (Wa, ba, Wg, bg, w2, c)
    fs = F.FusedAttnScores()
    for n in ([int(x) for x in sys.argv[1:]] or (20000, 50000, 100000, 400000)):
        torch.cuda.empty_cache()
        bags = [torch.randn(n, 512, device=dev).to(torch.bfloat16) for _ in range(4 if n > 100000 else 16)]
        a = t(lambda i: fs(bags[i % len(bags)], *W))
        b = t(lambda i: fs.scores_and_pool(bags[i % len(bags)], *W))
        c2 = t(lambda i: F.scored_pool(bags[i % len(bags)], fs(bags[i % len(bags)], *W)))
        fl = 2.0 * n * 512 * 256 * (2 if gated else 1)
        print(f"gated={gated} N={n:7d}: scores {a:7.2f} us ({fl / a / 1e6 / 2500 * 100:4.1f} % algorithmic)   scores + pooling, one launch {b:7.2f}   two launches {c2:7.2f}")
