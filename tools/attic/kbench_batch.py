import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from vlsa_amd import functional as F
dev = "cuda"
n = int(sys.argv[1]) if len(sys.argv) > 1 else 50000
for B in (1, 2, 4, 8, 16, 32):
    bags = [torch.randn(n, 512, device=dev).to(torch.bfloat16) for _ in range(max(B, 8))]
    Q = torch.randn(12, 512, device=dev); T = torch.randn(4, 512, device=dev)
    W = torch.randn(512, 512, device=dev) / 22; b = torch.randn(512, device=dev); ls = torch.tensor(4.03, device=dev)
    plan = F.VlfanBatchPlan(B, 12, 4, dev)
    plan.set_bags(bags[:B])
    for _ in range(3): plan.run(Q, T, ls, W, b)
    torch.cuda.synchronize()
    s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
    reps = max(2, 32 // B)
    with torch.cuda.stream(s):
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            for i in range(reps): plan.run(Q, T, ls, W, b)
    torch.cuda.current_stream().wait_stream(s)
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): g.replay()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / (10 * reps * B)
    print(f"B={B:2d} N={n}: {us:7.2f} us/bag  {n/us:8.1f} M patches/s  {n*1024/us/1e3:7.1f} GB/s (whole step incl. prep/merge/head)")
