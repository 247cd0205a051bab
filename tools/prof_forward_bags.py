import sys, os, time, cProfile, pstats
sys.path.insert(0, os.getcwd())
import torch
from vlsa_amd.vlsa import VLSA
dev = "cuda"
P, K = 12, 4
cfg = dict(name="VLFAN", dim_in=512, use_feat_proj=False, query="Parameter", num_query=P, query_pooling="mean")
net = VLSA(cfg, pretrained_text_features=torch.randn(K, 512)).to(dev).eval()
n = 2798
base = torch.randn(32 * n, 512, device=dev).to(torch.bfloat16)
bags = [base[i * n:(i + 1) * n][None] for i in range(32)]
with torch.no_grad():
    for i in range(20): net.forward_bags(bags)
    torch.cuda.synchronize()
    # GPU time alone
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e0.record()
    for i in range(100): net.forward_bags(bags)
    e1.record(); torch.cuda.synchronize()
    print("wall per call us", (time.perf_counter() - t0) / 100 * 1e6, "gpu span per call us", e0.elapsed_time(e1) * 10)
    t0 = time.perf_counter()
    for i in range(100): net.forward_bags(bags)
    print("host-only per call us (no sync)", (time.perf_counter() - t0) / 100 * 1e6)
    torch.cuda.synchronize()
    pr = cProfile.Profile(); pr.enable()
    for i in range(200): net.forward_bags(bags)
    pr.disable(); torch.cuda.synchronize()
    pstats.Stats(pr).sort_stats("tottime").print_stats(22)
