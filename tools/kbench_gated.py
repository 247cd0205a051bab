"""Fused attention-score kernel alone: time per bag and MFMA rate (algorithmic FLOP = 2 * N * 512 * 256 per branch;
executed MFMA FLOP = 2x that for the hi + lo weight split)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vlsa_amd import functional as F
dev = "cuda"
import gc; gc.collect(); gc.freeze()   # torch's ~10^6 imported objects out of the collector's way: a gen-2 pass otherwise stalls one call by ~40 ms (profiles/README.md)
for gated in (True, False):
    Wa = torch.randn(256, 512, device=dev) / 22; ba = torch.randn(256, device=dev) * 0.05
    Wg = torch.randn(256, 512, device=dev) / 22 if gated else None; bg = torch.randn(256, device=dev) * 0.05 if gated else None
    w2 = torch.randn(1, 256, device=dev) / 16; c = torch.randn(1, device=dev)
    fs = F.FusedAttnScores()
    for n, dt in ((50000, torch.bfloat16), (40000, torch.bfloat16), (70000, torch.bfloat16), (20000, torch.bfloat16), (10000, torch.bfloat16), (2798, torch.bfloat16), (400000, torch.bfloat16),
                  (50000, torch.float32), (10000, torch.float32), (2798, torch.float32)):
        torch.cuda.empty_cache()   # every configuration allocates from fresh allocator segments
        bags = [torch.randn(n, 512, device=dev).to(dt) for _ in range(4 if n > 100000 else 16)]
        for i in range(60): fs(bags[i % len(bags)], Wa, ba, Wg, bg, w2, c)
        torch.cuda.synchronize()
        us = 1e30
        for _ in range(4):                      # best of 4 chunks of 100: one host hiccup does not end up in the committed figure
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for i in range(100): fs(bags[i % len(bags)], Wa, ba, Wg, bg, w2, c)
            e1.record(); torch.cuda.synchronize()
            us = min(us, e0.elapsed_time(e1) * 1e3 / 100)
        fl = 2.0 * n * 512 * 256 * (2 if gated else 1)
        terms = 2 if dt == torch.bfloat16 else 3
        print(f"gated={gated} {str(dt)[6:]:8s} N={n:7d}: {us:8.2f} us/bag  algorithmic {fl / us / 1e6:7.1f} TFLOP/s  = {fl / us / 1e6 / 2500 * 100:4.1f}% of 2.5 PFLOP/s; executed ({terms} split terms) "
              f"{terms * fl / us / 1e6:7.1f} TFLOP/s = {terms * fl / us / 1e6 / 2500 * 100:5.1f}%")
