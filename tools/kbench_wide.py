"""Bags per forward launch for slide-sized bags (round 4: up to 256): streaming kernel and whole step per bag, by bag size, launch width B
and bags in flight S.  `python tools/kbench_wide.py`"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vlsa_amd import functional as F
dev = "cuda"
import gc; gc.collect(); gc.freeze()
Q = torch.randn(12, 512, device=dev); T = torch.randn(4, 512, device=dev)
W = torch.randn(512, 512, device=dev) / 22; b = torch.randn(512, device=dev); ls = torch.tensor(4.03, device=dev)


def t(fn, n=40):
    for _ in range(10): fn()
    torch.cuda.synchronize()
    best = 1e30
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n): fn()
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1e3 / n)
    return best


for dt in (torch.bfloat16, torch.float32):
    for n in ((700, 2798, 6000, 12000) if len(sys.argv) < 2 else [int(a) for a in sys.argv[1:]]):
        torch.cuda.empty_cache()
        base = torch.randn(256 * n, 512, device=dev).to(dt)
        for B in (32, 64, 128, 256) if n >= 20000 else (64, 128, 256):
            bags = [base[i * n:(i + 1) * n] for i in range(B)]
            plan = F.VlfanBatchPlan(B, 12, 4, dev)
            plan.set_bags(bags)
            auto = plan.groups
            plan.run(Q, T, ls, W, b)
            res = []
            for S in (0, 8, 16, 32, 64, 128, 256):
                if S > B:
                    continue
                plan.groups = S or auto
                us = t(plan.run_partial_only) / B
                res.append(f"S={S or auto:3d}{'*' if not S else ' '}: {us:5.2f} ({n * 512 * base.element_size() / us / 1e6:4.2f} TB/s)")
            plan.groups = auto
            whole = t(lambda: plan.run(Q, T, ls, W, b)) / B
            print(f"{str(dt)[6:]:9s} N={n:6d} B={B:3d} | kernel us/bag: " + " | ".join(res) + f" | whole step {whole:5.2f} us/bag = {n / whole / 1e3:5.2f} G patches/s")
            del plan, bags
        del base
