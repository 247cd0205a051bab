"""Eager launch sequence vs hipGraph replay of a batch, small and large bags."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from vlsa_amd import functional as F
dev = "cuda"
Q = torch.randn(12, 512, device=dev); T = torch.randn(4, 512, device=dev)
W = torch.randn(512, 512, device=dev) / 22; b = torch.randn(512, device=dev); ls = torch.tensor(4.03, device=dev)
for n, B in ((300, 64), (2798, 64), (10000, 32), (50000, 32)):
    base = torch.randn(B * n, 512, device=dev).to(torch.bfloat16)
    plan = F.VlfanBatchPlan(B, 12, 4, dev)
    plan.set_bags([base[i * n:(i + 1) * n] for i in range(B)])
    g = plan.capture(Q, T, ls, W, b)
    def timeit(f, reps=300):
        for _ in range(50): f()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(reps): f()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / reps * 1e6
    te = timeit(lambda: plan.run(Q, T, ls, W, b)); tg = timeit(g.replay)
    print(f"N={n:6d} B={B}: eager {te:7.1f} us/launch ({te / B:6.2f} us/bag)   graph replay {tg:7.1f} us/launch ({tg / B:6.2f} us/bag)")
