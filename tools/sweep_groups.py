"""Streaming-kernel time per bag vs bag size and number of bags in flight (groups), bf16 and fp32."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vlsa_amd import functional as F
dev = "cuda"
import gc; gc.collect(); gc.freeze()   # torch's ~10^6 imported objects out of the collector's way: a gen-2 pass otherwise stalls one call by ~40 ms (profiles/README.md)
Q = torch.randn(12, 512, device=dev); T = torch.randn(4, 512, device=dev)
W = torch.randn(512, 512, device=dev) / 22; b = torch.randn(512, device=dev); ls = torch.tensor(4.03, device=dev)
for dt in (torch.bfloat16, torch.float32):
    for n, B in ((2798, 64), (2798, 32), (10000, 32), (20000, 32), (50000, 32)):
        torch.cuda.empty_cache()   # every configuration allocates from fresh allocator segments
        if dt == torch.float32 and n == 50000:
            B = 16
        base = torch.randn(B * n, 512, device=dev).to(dt)
        bags = [base[i * n:(i + 1) * n] for i in range(B)]
        plan = F.VlfanBatchPlan(B, 12, 4, dev)
        plan.set_bags(bags)
        auto = plan.groups
        plan.run(Q, T, ls, W, b)
        res = []
        for S in (8, 16, 32, 64):
            if S > B:
                continue
            plan.groups = S
            for _ in range(30): plan.run_partial_only()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(50): plan.run_partial_only()
            e1.record(); torch.cuda.synchronize()
            us = e0.elapsed_time(e1) * 1e3 / 50 / B
            res.append(f"S={S}: {us:6.2f} us/bag ({n * 512 * base.element_size() / us / 1e6:5.2f} TB/s)")
        print(f"{str(dt)[6:]:9s} N={n:6d} B={B:2d} auto S={auto:2d} | " + " | ".join(res))
        del base, bags, plan
