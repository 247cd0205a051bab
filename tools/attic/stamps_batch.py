import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from vlsa_amd import functional as F, _native
raw = ctypes.CDLL(_native.lib_path())
dev = "cuda"
B, n = (32, 50000) if len(sys.argv) < 2 else (int(sys.argv[1]), int(sys.argv[2]))
ALIAS = len(sys.argv) > 3
base = torch.randn(n, 512, device=dev).to(torch.bfloat16)
bags = [base if ALIAS else torch.randn(n, 512, device=dev).to(torch.bfloat16) for _ in range(B)]
Q = torch.randn(12, 512, device=dev); T = torch.randn(4, 512, device=dev)
W = torch.randn(512, 512, device=dev) / 22; b = torch.randn(512, device=dev); ls = torch.tensor(4.03, device=dev)
plan = F.VlfanBatchPlan(B, 12, 4, dev); plan.set_bags(bags)
for _ in range(3): plan.run(Q, T, ls, W, b)
torch.cuda.synchronize()
buf = (ctypes.c_longlong * 64)()
raw.vlsa_debug_read_batch_cycles(buf)
t = [x for x in list(buf)[:40] if x > 0]
print("iteration ends, deltas (cycles) block 3:", [t[i] - t[i-1] for i in range(1, len(t))])
print("total", t[-1] - t[0], "stamps", len(t))
it = list(buf)[40:48]
names = ["top", "tile landed (vmcnt)", "scores MFMA done", "barrier 1", "exchange written + barrier 2", "exchange read, T ready", "exp2 / split done", "weighted-sum MFMA done"]
rt = list(buf)[62:64]
print(f"shader clock over stamps 1..39: {(list(buf)[39] - list(buf)[1]) / max(rt[1] - rt[0], 1) * 100:.0f} MHz (s_memtime cycles per 100 MHz s_memrealtime tick)")
print("inside own tile 12 (cycles since its top):")
for nme, v in zip(names, it):
    print(f"   {nme:32s} {v - it[0]:6d}")
