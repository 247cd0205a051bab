"""Tile-height sweep of the attention-score kernel in ONE process (VLSA_GS_ROWS is read per call): us per bag for every
(N, rows per tile); one launch per bag (VLSA_GS_SPLIT=0).  argv: gated(0/1) [shape]"""
import sys, os
gated = bool(int(sys.argv[1])) if len(sys.argv) > 1 else False
os.environ["VLSA_GS_HG2"] = sys.argv[2] if len(sys.argv) > 2 else "1"
os.environ["VLSA_GS_SPLIT"] = "0"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from vlsa_amd import functional as F
import gc; gc.collect(); gc.freeze()
dev = "cuda"
Wa = torch.randn(256, 512, device=dev) / 22; ba = torch.randn(256, device=dev) * 0.05
Wg = torch.randn(256, 512, device=dev) / 22 if gated else None; bg = torch.randn(256, device=dev) * 0.05 if gated else None
w2 = torch.randn(1, 256, device=dev) / 16; c = torch.randn(1, device=dev)
fs = F.FusedAttnScores()
max_rows = int(os.environ.get("SWEEP_MAX_ROWS", "128"))
heights = list(range(16, max_rows + 1, 16))
print("N".rjust(7) + "".join(f"{h:8d}" for h in heights) + "   default")
for n in (2798, 5000, 10000, 15000, 20000, 30000, 40000, 50000, 60000, 70000, 85000, 100000, 150000):
    bags = [torch.randn(n, 512, device=dev).bfloat16() for _ in range(8)]
    row = []
    for h in heights + [0]:
        if h: os.environ["VLSA_GS_ROWS"] = str(h)
        else: os.environ.pop("VLSA_GS_ROWS", None)
        for i in range(30): fs(bags[i % 8], Wa, ba, Wg, bg, w2, c)
        torch.cuda.synchronize()
        us = 1e30
        for _ in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for i in range(60): fs(bags[i % 8], Wa, ba, Wg, bg, w2, c)
            e1.record(); torch.cuda.synchronize()
            us = min(us, e0.elapsed_time(e1) * 1e3 / 60)
        row.append(us)
    print(f"{n:7d}" + "".join(f"{u:8.1f}" for u in row), flush=True)
    del bags; torch.cuda.empty_cache()
