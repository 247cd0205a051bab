import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from vlsa_amd import functional as F, _native
lib = _native.load()
raw = ctypes.CDLL(_native.lib_path())
raw.vlsa_debug_read_cycles.argtypes = [ctypes.c_void_p]
qp = F.prepare_queries(torch.randn(12, 512, device="cuda"))
names = ["start", "after setup", "qf retired", "tile0 landed", "scores done", "exchanged", "softmax done", "loop end", "merge barrier", "stores issued", "stores done"]
for n in (32, 8192, 50000):
    x = torch.randn(n, 512, device="cuda").to(torch.bfloat16)
    for _ in range(3):
        F.vlfan_partial(x, qp, kernel=3)
    torch.cuda.synchronize()
    buf = (ctypes.c_longlong * 32)()
    raw.vlsa_debug_read_cycles(buf)
    t = list(buf)[:11]
    print("N =", n, " (cycles of the 100 MHz? clock -> deltas)")
    for i in range(1, 11):
        print(f"   {names[i]:>16s}: +{t[i]-t[i-1]:6d}  (total {t[i]-t[0]:6d})")
