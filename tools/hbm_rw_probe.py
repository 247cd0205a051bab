"""What a 4.7 % write stream next to the read stream costs THE MACHINE (tools/probes/hbm_read_probe.hip, round 5): the production
access pattern + LDS-DMA ring with NO arithmetic over 64 x 50k x 512 bf16 rows (3.28 GB), without and with the production kernel's
score lines (a [12, N] fp32 matrix: 48 B per 1 KiB row, whole 128-byte lines per store instruction) -- next to the product with and
without its score output.  If the bare probe slows down by what the product slows down, the attention-weight leg is at the
machine's bound for this read / write mix."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from vlsa_amd import functional as F
lib = ctypes.CDLL(os.path.join(ROOT, "tools", "probes", "libhbm_read_probe.so"))
lib.hbm_rw_probe_launch.argtypes = [ctypes.c_void_p, ctypes.c_longlong, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]
dev = "cuda"
B, n = 64, 50000
base = torch.randn(B * n + 4096, 512, device=dev).to(torch.bfloat16)
bags = [base[i * n:(i + 1) * n] for i in range(B)]
nbytes = B * n * 1024
sink = torch.zeros(4, dtype=torch.int32, device=dev)
wout = torch.empty(12, B * n + 4096, dtype=torch.float32, device=dev)
Q = torch.randn(12, 512, device=dev); T = torch.randn(4, 512, device=dev)
W = torch.randn(512, 512, device=dev) / 22; b = torch.randn(512, device=dev); ls = torch.tensor(4.03, device=dev)
plans = {}
for attn in (False, True):
    pl = F.VlfanBatchPlan(B, 12, 4, dev, want_attn=attn)
    pl.set_bags(bags)
    pl.run(Q, T, ls, W, b)
    plans[attn] = pl


def timed(fn, reps=30, warm=30):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for e0, e1 in ev:
        e0.record(); fn(); e1.record()
    torch.cuda.synchronize()
    ts = sorted(e0.elapsed_time(e1) for e0, e1 in ev)
    return sum(ts) / len(ts) * 1e3, ts[0] * 1e3


s = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
rows = {}
MODE = 8 + (48 << 4)         # production pattern + phase structure + LDS + MFMA load (hbm_read_probe.py's last row)
for rnd in range(2):
    for name, mode, w, wm in (("probe, no arithmetic: reads only", 0, None, 0), ("probe, no arithmetic: reads + score lines (write-back stores)", 0, wout, 0),
                              ("probe, no arithmetic: reads + score lines (nontemporal stores)", 0, wout, 1),
                              ("probe with the product's phase structure + LDS + MFMA load: reads only", MODE, None, 0),
                              ("probe with the product's phase structure + LDS + MFMA load: reads + score lines", MODE, wout, 0)):
        fn = lambda: lib.hbm_rw_probe_launch(ctypes.c_void_p(base.data_ptr()), nbytes, mode, ctypes.c_void_p(sink.data_ptr()),
                                             None if w is None else ctypes.c_void_p(w.data_ptr()), wm, s)
        rows.setdefault(name, []).append(timed(fn))
    rows.setdefault("k_vlfan_partial_dma_batch<false> (the product)", []).append(timed(plans[False].run_partial_only))
    rows.setdefault("k_vlfan_partial_dma_batch<true>  (the product + its score lines)", []).append(timed(plans[True].run_partial_only))
ref = None
for name, v in rows.items():
    avg = sum(a for a, _ in v) / len(v); mn = min(m for _, m in v)
    print(f"{name:92s}: {avg:7.1f} us avg {mn:7.1f} min = {nbytes / avg / 1e6:5.2f} TB/s of rows")
