"""Forward + backward timing of the aggregation (training path) per bag."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vlsa_amd import functional as F
dev = "cuda"
import gc; gc.collect(); gc.freeze()   # torch's ~10^6 imported objects out of the collector's way: a gen-2 pass otherwise stalls one call by ~40 ms (profiles/README.md)
for n, dt in ((50000, torch.bfloat16), (50000, torch.float32), (10000, torch.float32), (2798, torch.float32)):
    torch.cuda.empty_cache()   # every configuration allocates from fresh allocator segments
    bags = [torch.randn(n, 512, device=dev).to(dt) for _ in range(8)]
    Q = torch.randn(12, 512, device=dev, requires_grad=True)
    G = torch.randn(12, 512, device=dev)
    def step(i):
        out, _ = F.vlfan_cross_attention(bags[i % 8], Q)
        (out * G).sum().backward()
    for i in range(5): step(i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(50): step(i)
    torch.cuda.synchronize()
    us = (time.perf_counter() - t0) / 50 * 1e6
    print(f"N={n} {str(dt)[6:]}: fwd+bwd {us:8.1f} us/bag (eager, python included)")

# 32 bags per optimizer step through the persistent batch kernels (forward + backward)
for n, dt in ((50000, torch.bfloat16), (10000, torch.bfloat16), (2798, torch.bfloat16), (10000, torch.float32)):
    torch.cuda.empty_cache()   # every configuration allocates from fresh allocator segments
    base = torch.randn(32 * n + 4096, 512, device=dev).to(dt)
    bags = [base[i * n:(i + 1) * n] for i in range(32)]
    Q = torch.randn(12, 512, device=dev, requires_grad=True)
    G = torch.randn(32, 12, 512, device=dev)
    def bstep():
        out = F.vlfan_cross_attention_bags(bags, Q)
        (out * G).sum().backward()
    for i in range(3): bstep()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(20): bstep()
    torch.cuda.synchronize()
    us = (time.perf_counter() - t0) / 20 / 32 * 1e6
    print(f"N={n} {str(dt)[6:]} x32 bags: batched fwd+bwd {us:8.2f} us/bag")
