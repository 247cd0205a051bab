"""Zero-shot path (FeatMIL identity + logit pooling, model/vlsa.py:185-196, model/deepmil.py:16-37): wall time per bag."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vlsa_amd.vlsa import VLSA
dev = "cuda"
import gc; gc.collect(); gc.freeze()   # torch's ~10^6 imported objects out of the collector's way: a gen-2 pass otherwise stalls one call by ~40 ms (profiles/README.md)
K = 4
for pooling in ("logit_mean", "logit_max", "logit_top10"):
    cfg = dict(name="FeatMIL", dim_in=512, pooling=pooling)
    net = VLSA.from_modules(cfg, pretrained_text_features=torch.randn(K, 512)).to(dev).eval()
    for n, dt in ((50000, torch.bfloat16), (50000, torch.float32), (2798, torch.float32)):
        torch.cuda.empty_cache()   # every configuration allocates from fresh allocator segments
        bags = [torch.randn(1, n, 512, device=dev).to(dt) for _ in range(8)]
        with torch.no_grad():
            for i in range(40): net(bags[i % 8])
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for i in range(200): net(bags[i % 8])
            torch.cuda.synchronize(); us = (time.perf_counter() - t0) / 200 * 1e6
            flat = [b[0] for b in bags] * 4                        # 32 bags through forward_bags: one score launch + one pooling launch
            for i in range(10): net.forward_bags(flat)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for i in range(40): net.forward_bags(flat)
            torch.cuda.synchronize(); usb = (time.perf_counter() - t0) / 40 / 32 * 1e6
        print(f"{pooling:12s} N={n:6d} {str(dt)[6:]:9s}: net(X) {us:7.1f} us/bag  {n / us:8.1f} M patches/s   forward_bags(32) {usb:7.2f} us/bag  {n / usb:8.1f} M patches/s")
