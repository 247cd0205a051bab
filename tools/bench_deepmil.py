"""DeepMIL encoder forward (attention / gated-attention pooling over N patches): time per bag and the
MFMA-side rate of the hidden projections (2 * N * 512 * 256 FLOP per branch)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vlsa_amd.deepmil import DeepMIL
dev = "cuda"
import gc; gc.collect(); gc.freeze()   # torch's ~10^6 imported objects out of the collector's way: a gen-2 pass otherwise stalls one call by ~40 ms (profiles/README.md)
for pooling, branches in (("attention", 1), ("gated_attention", 2)):
    m = DeepMIL(dim_in=512, dim_hid=256, use_feat_proj=False, pooling=pooling, pred_head="Adapter").to(dev).eval()
    for n, dt in ((50000, torch.bfloat16), (50000, torch.float32), (10000, torch.float32), (2798, torch.float32)):
        torch.cuda.empty_cache()   # every configuration allocates from fresh allocator segments
        bags = [torch.randn(1, n, 512, device=dev).to(dt) for _ in range(8)]
        with torch.no_grad():
            for i in range(40): m(bags[i % 8])
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for i in range(200): m(bags[i % 8])
            torch.cuda.synchronize()
        us = (time.perf_counter() - t0) / 200 * 1e6
        # the same encoder over 32 bags per call: one score launch + one pooling launch (DeepMIL.pool_bags) + batched head
        flat = [b[0] for b in (bags * 4)]
        with torch.no_grad():
            for i in range(5): m.visual_adapter(m.pool_bags(flat))
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for i in range(20): m.visual_adapter(m.pool_bags(flat))
            torch.cuda.synchronize()
        usb = (time.perf_counter() - t0) / 20 / 32 * 1e6
        fl = 2.0 * n * 512 * 256 * branches
        print(f"{pooling:16s} N={n:6d} {str(dt)[6:]:9s}: {us:8.1f} us/bag  {n / us:8.1f} M patches/s  projections {fl / us / 1e6:7.1f} TFLOP/s   pool_bags(32) {usb:7.2f} us/bag ({fl / usb / 1e6:6.1f} TFLOP/s)")
