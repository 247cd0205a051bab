"""DeepMIL module call on a slide-sized bag: host time per call (no sync inside 200 calls -- the launch queue absorbs them) next to the
event-timed rate and the cProfile top of the host side.  Run under `rocprofv3 --kernel-trace --stats` for the GPU chain."""
import sys, os, time, cProfile, pstats
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from vlsa_amd.deepmil import DeepMIL
dev = "cuda"
torch.manual_seed(1)
m = DeepMIL(dim_in=512, dim_hid=256, use_feat_proj=False, pooling="gated_attention", pred_head="Adapter").to(dev).eval()
bags = [torch.randn(2798, 512, device=dev).to(torch.bfloat16)[None] for _ in range(16)]
with torch.no_grad():
    for i in range(50): m(bags[i % 16])
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(5):
        t0 = time.perf_counter()
        for i in range(200): m(bags[i % 16])
        t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
        best = min(best, (t1 - t0) / 200 * 1e6)
        print(f"host {1e6 * (t1 - t0) / 200:6.1f} us per call, with the drain {1e6 * (t2 - t0) / 200:6.1f}")
    if "prof" in sys.argv:
        pr = cProfile.Profile(); pr.enable()
        for i in range(2000): m(bags[i % 16])
        pr.disable(); torch.cuda.synchronize()
        pstats.Stats(pr).sort_stats("tottime").print_stats(14)
