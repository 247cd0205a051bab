import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vlsa_amd import functional as F, _native
raw = ctypes.CDLL(_native.lib_path())
dev = "cuda"
B, n = 8, 50000
bags = [torch.randn(n, 512, device=dev).to(torch.bfloat16) for _ in range(B)]
Q = torch.randn(12, 512, device=dev); T = torch.randn(4, 512, device=dev)
W = torch.randn(512, 512, device=dev) / 22; b = torch.randn(512, device=dev); ls = torch.tensor(4.03, device=dev)
plan = F.VlfanBatchPlan(B, 12, 4, dev); plan.set_bags(bags)
for _ in range(3): plan.run(Q, T, ls, W, b)
torch.cuda.synchronize()
buf = (ctypes.c_longlong * 64)()
raw.vlsa_debug_read_batch_cycles(buf)
t = [x for x in list(buf) if x > 0]
print("deltas (cycles) block 3:", [t[i] - t[i-1] for i in range(1, len(t))])
print("total", t[-1] - t[0], "stamps", len(t))
