"""Batched forward WITH and WITHOUT attention weights: 32 x 50k bf16 bags per launch, P = 12, K = 4.
Algorithmic bytes per patch: 1024 B read, + 4 P = 48 B written when A is requested."""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vlsa_amd import functional as F

dev = "cuda"
import gc; gc.collect(); gc.freeze()   # torch's ~10^6 imported objects out of the collector's way: a gen-2 pass otherwise stalls one call by ~40 ms (profiles/README.md)
B, n, P, K = int(sys.argv[3]) if len(sys.argv) > 3 else 32, int(sys.argv[1]) if len(sys.argv) > 1 else 50000, 12, 4
DT = torch.float32 if (len(sys.argv) > 2 and sys.argv[2] == "fp32") else torch.bfloat16
NB = max(B, 32)        # distinct bags in HBM (>= 1.6 GB at 50k: nothing is served from the Infinity Cache by accident)
allbags = [torch.randn(n, 512, device=dev).to(DT) for _ in range(NB)]
bags = allbags[:B]
Q = torch.randn(P, 512, device=dev); T = torch.randn(K, 512, device=dev)
W = torch.randn(512, 512, device=dev) / 22; b = torch.randn(512, device=dev); ls = torch.tensor(4.03, device=dev)
streams = [torch.cuda.Stream(), torch.cuda.Stream()]
for want in (False, True):
    # as bench.py: launches alternate over two streams, the tail of one overlaps the next one's stream; with B < 32 the plans
    # walk through all NB distinct bags (plan j owns bags j B .. j B + B - 1)
    plans = [F.VlfanBatchPlan(B, P, K, dev, want_attn=want) for _ in range(max(2, NB // B))]
    for j, pl_ in enumerate(plans):
        pl_.set_bags(allbags[(j * B) % NB:(j * B) % NB + B])
    plan = plans[0]
    NPL = len(plans)

    def go(R):
        for st in streams:
            st.wait_stream(torch.cuda.current_stream())
        for i in range(R):
            with torch.cuda.stream(streams[i & 1]):
                plans[i % NPL].run(Q, T, ls, W, b)
        for st in streams:
            torch.cuda.current_stream().wait_stream(st)
    go(60 * 32 // B)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    R = 200 * 32 // B
    go(R)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / R
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ts = []
    for _ in range(30):
        e0.record(); plan.run_partial_only(); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
    kern = sorted(ts)[len(ts) // 2] * 1e-3
    bytes_pp = 512 * bags[0].element_size() + (4 * P if want else 0)
    print(f"{str(DT)[6:]} want_attn={want}: step {dt * 1e6:.1f} us per {B} bags = {dt / B * 1e6:.2f} us/bag = {B * n / dt / 1e9:.2f} G patches/s; "
          f"whole-step {B * n * bytes_pp / dt / 1e12:.2f} TB/s = {B * n * bytes_pp / dt / 8e12 * 100:.1f} % of the ({bytes_pp} B/patch) HBM roofline; "
          f"streaming kernel alone {kern * 1e6:.1f} us (event pair included)")
