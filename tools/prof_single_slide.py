"""Where one `net(X)` call goes (the reference handler's per-slide pattern, runner/vlsa_handler.py:322-330): wall per call for a tiny and
a 50k bag, host-only time of the same loop (no sync inside), the bare C call, and a cProfile of the Python side."""
import cProfile
import ctypes
import os
import pstats
import sys
import time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from vlsa_amd.vlsa import VLSA

dev = torch.device("cuda", 0)
D, P, K = 512, 12, 4
cfg = dict(name="VLFAN", dim_in=D, use_feat_proj=False, query="Parameter", num_query=P, query_pooling="mean")
net = VLSA.from_modules(cfg, pretrained_text_features=torch.randn(K, D, generator=torch.Generator().manual_seed(99))).to(dev).eval()
import gc; gc.collect(); gc.freeze()
for rows in ((50000,) if "50k" in sys.argv else (2798, 50000)):
    g = torch.Generator(device=dev).manual_seed(700)
    bags = [torch.randn(rows, D, device=dev, generator=g).to(torch.bfloat16)[None] for _ in range(32)]
    with torch.no_grad():
        for i in range(200):
            net(bags[i % 32])
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(640):
            net(bags[i % 32])
        host = (time.perf_counter() - t0) / 640 * 1e6
        torch.cuda.synchronize()
        wall = (time.perf_counter() - t0) / 640 * 1e6
        print(f"N={rows}: wall {wall:.1f} us per call, host side of the same loop {host:.1f} us per call")
        if "prof" in sys.argv:
            pr = cProfile.Profile()
            pr.enable()
            for i in range(2000):
                net(bags[i % 32])
            pr.disable()
            torch.cuda.synchronize()
            pstats.Stats(pr).sort_stats("tottime").print_stats(18)
