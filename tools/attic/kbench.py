"""Kernel-only timing of the streaming kernel via hipGraph of back-to-back launches (no python overhead)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from vlsa_amd import functional as F

def bench(n, dt, kern, reps=40, bags=8):
    dev = "cuda"
    xs = [torch.randn(n, 512, device=dev).to(dt) for _ in range(bags)]
    Q = torch.randn(12, 512, device=dev); T = torch.randn(4, 512, device=dev)
    W = torch.randn(512, 512, device=dev) / 22; b = torch.randn(512, device=dev); ls = torch.tensor(4.03, device=dev)
    plan = F.VlfanInferencePlan(n, 512, 12, 4, dev, kernel=kern)
    plan.run(xs[0], Q, T, ls, W, b)
    torch.cuda.synchronize()
    s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            for i in range(reps):
                plan.run_partial_only(xs[i % bags])
    torch.cuda.current_stream().wait_stream(s)
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): g.replay()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / (5 * reps)
    nbytes = n * 512 * xs[0].element_size()
    print(f"N={n} {str(dt)[6:]} kernel={kern}: {us:7.2f} us/launch (incl ~1.5us boundary)  {nbytes/us/1e3:7.1f} GB/s")

if __name__ == "__main__":
    for n, dt, kerns in ((50000, torch.bfloat16, (3, 2)), (200000, torch.bfloat16, (3, 2)), (10000, torch.float32, (2,)), (10000, torch.bfloat16, (3,)), (2798, torch.float32, (2,))):
        for k in kerns:
            bench(n, dt, k)
