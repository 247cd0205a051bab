import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vlsa_amd import functional as F
dev = "cuda"
import gc; gc.collect(); gc.freeze()   # torch's ~10^6 imported objects out of the collector's way: a gen-2 pass otherwise stalls one call by ~40 ms (profiles/README.md)
for n, dt, B in ((50000, torch.float32, 32), (10000, torch.float32, 32), (2798, torch.float32, 32), (50000, torch.bfloat16, 32), (10000, torch.bfloat16, 32)):
    torch.cuda.empty_cache()   # every configuration allocates from fresh allocator segments
    bags = [torch.randn(n, 512, device=dev).to(dt) for _ in range(B)]
    Q = torch.randn(12, 512, device=dev); T = torch.randn(4, 512, device=dev)
    W = torch.randn(512, 512, device=dev) / 22; b = torch.randn(512, device=dev); ls = torch.tensor(4.03, device=dev)
    plan = F.VlfanBatchPlan(B, 12, 4, dev); plan.set_bags(bags)
    for _ in range(20): plan.run(Q, T, ls, W, b)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    us = 1e30
    for _ in range(3):          # best of 3 chunks of 30 launches (settled clocks)
        e0.record()
        for _ in range(30): plan.run_partial_only()
        e1.record(); torch.cuda.synchronize()
        us = min(us, e0.elapsed_time(e1) * 1e3 / 30)
    nbytes = B * n * 512 * bags[0].element_size()
    e0.record()
    for _ in range(10): plan.run(Q, T, ls, W, b)
    e1.record(); torch.cuda.synchronize()
    us2 = e0.elapsed_time(e1) * 1e3 / 10
    print(f"N={n} {str(dt)[6:]} B={B}: stream kernel {us:8.1f} us/launch = {us/B:6.2f} us/bag {nbytes/us/1e3:7.1f} GB/s ({nbytes/us/8e6*100:.1f}% of 8 TB/s); whole step {us2/B:6.2f} us/bag {n*B/us2:8.1f} M patches/s")
