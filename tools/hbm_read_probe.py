"""HBM read ceiling of this box next to the production streaming kernel (tools/probes/hbm_read_probe.hip): the same 1.6384 GB
(32 x 50k x 512 bf16 rows) read by (0) the production access pattern + LDS-DMA ring with no arithmetic, (1) the ring on contiguous
pieces, (2) plain nontemporal dwordx4 loads, and by k_vlfan_partial_dma_batch itself."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from vlsa_amd import functional as F
lib = ctypes.CDLL(os.path.join(ROOT, "tools", "probes", "libhbm_read_probe.so"))
lib.hbm_read_probe_launch.argtypes = [ctypes.c_void_p, ctypes.c_longlong, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
dev = "cuda"
B, n = 32, 50000
base = torch.randn(B * n + 4096, 512, device=dev).to(torch.bfloat16)
bags = [base[i * n:(i + 1) * n] for i in range(B)]
nbytes = B * n * 1024
sink = torch.zeros(4, dtype=torch.int32, device=dev)
Q = torch.randn(12, 512, device=dev); T = torch.randn(4, 512, device=dev)
W = torch.randn(512, 512, device=dev) / 22; b = torch.randn(512, device=dev); ls = torch.tensor(4.03, device=dev)
plan = F.VlfanBatchPlan(B, 12, 4, dev)
plan.set_bags(bags)
plan.run(Q, T, ls, W, b)


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
for rnd in range(2):
    for mode, name in ((0, "LDS-DMA ring, production pattern (32 rows x 256 B per wave tile), no arithmetic"),
                       (1, "LDS-DMA ring, contiguous 8 KiB per wave tile"), (2, "plain nontemporal dwordx4 loads"),
                       (4, "production pattern + the two workgroup barriers per iteration, no arithmetic"),
                       (3, "4 independent waves per workgroup, whole rows: 16 rows x 1 KiB per wave tile"),
                       (4 + (2 << 4), "production pattern + barriers + slot held ~0.2 us per iteration"),
                       (4 + (4 << 4), "production pattern + barriers + slot held ~0.4 us per iteration"),
                       (4 + (8 << 4), "production pattern + barriers + slot held ~0.85 us per iteration"),
                       (4 + (12 << 4), "production pattern + barriers + slot held ~1.3 us per iteration"),
                       (4 + (16 << 4), "production pattern + barriers + the product's LDS read volume (32 x ds_read_b128 per wave tile)"),
                       (4 + (32 << 4), "production pattern + barriers + the product's MFMA count (48 per wave tile)"),
                       (4 + (48 << 4), "production pattern + barriers + both"),
                       (8 + (48 << 4), "production pattern + the product's phase structure (work | barrier | exchange | barrier | work) + LDS + MFMA")):
        avg, mn = timed(lambda: lib.hbm_read_probe_launch(ctypes.c_void_p(base.data_ptr()), nbytes, mode, ctypes.c_void_p(sink.data_ptr()), s))
        rows.setdefault(name, []).append((avg, mn))
    avg, mn = timed(plan.run_partial_only)
    rows.setdefault("k_vlfan_partial_dma_batch (the product: scores, softmax, weighted row sums)", []).append((avg, mn))
for name, v in rows.items():
    avg = sum(a for a, _ in v) / len(v); mn = min(m for _, m in v)
    print(f"{name:95s}: {avg:7.1f} us avg {mn:7.1f} min = {nbytes / avg / 1e6:5.2f} TB/s avg, {nbytes / mn / 1e6:5.2f} TB/s best")
