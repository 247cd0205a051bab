"""Feat_Projecter over all rows of a bag: the fused HIP kernel (vlsa_feat_project) vs the torch modules (rocBLAS GEMM + LayerNorm)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vlsa_amd.layers import Feat_Projecter
dev = "cuda"
import gc; gc.collect(); gc.freeze()   # torch's ~10^6 imported objects out of the collector's way: a gen-2 pass otherwise stalls one call by ~40 ms (profiles/README.md)
m = Feat_Projecter(512, 512).to(dev)


def timed(fn, reps=50):
    for _ in range(10): fn()
    torch.cuda.synchronize()
    best = 1e30
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps): fn()
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1e3 / reps)
    return best


for n, dt in ((50000, torch.bfloat16), (10000, torch.bfloat16), (2798, torch.bfloat16), (50000, torch.float32), (10000, torch.float32), (2798, torch.float32)):
    torch.cuda.empty_cache()   # every configuration allocates from fresh allocator segments
    X = torch.randn(n, 512, device=dev).to(dt)
    with torch.no_grad():
        t_hip = timed(lambda: m(X))
        Xf = X.float()
        t_torch = timed(lambda: m.projecter(Xf))
    terms = 2 if dt == torch.bfloat16 else 3
    fl = 2.0 * n * 512 * 512
    print(f"N={n:6d} {str(dt)[6:]:9s}: fused {t_hip:8.1f} us/bag ({terms * fl / t_hip / 1e6:7.1f} TFLOP/s executed, {fl / t_hip / 1e6:6.1f} algorithmic; "
          f"{n * 2048 / t_hip / 1e3:6.0f} GB/s written)   torch modules on fp32 {t_torch:8.1f} us/bag")
