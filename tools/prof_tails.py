"""The launch chains behind the streaming kernels, for rocprofv3 --kernel-trace --stats: (a) one slide per call through the drop-in
module (`net(X)`, 50k x 512 bf16), (b) 256 slide-sized bags (2 798 patches) per forward launch at plan level, (c) 64 x 50k per launch.
`python tools/prof_tails.py [single|wide|batch ...]`; prints wall time per call next to the trace."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vlsa_amd import functional as F
from vlsa_amd.vlsa import VLSA
dev = "cuda"
import gc; gc.collect(); gc.freeze()
which = sys.argv[1:] or ["single", "wide", "batch"]
g = torch.Generator(device=dev).manual_seed(5)
Q = torch.randn(12, 512, device=dev, generator=g); T = torch.randn(4, 512, device=dev, generator=g)
W = torch.randn(512, 512, device=dev, generator=g) / 22; b = torch.randn(512, device=dev, generator=g); ls = torch.tensor(4.03, device=dev)


def wall(fn, n):
    for _ in range(max(8, n // 4)): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e6


if "single" in which:
    cfg = dict(name="VLFAN", dim_in=512, use_feat_proj=False, query="Parameter", num_query=12, query_pooling="mean")
    net = VLSA.from_modules(cfg, pretrained_text_features=torch.randn(4, 512)).to(dev).eval()
    bags = [torch.randn(50_000, 512, device=dev, generator=g).to(torch.bfloat16)[None] for _ in range(32)]
    i = [0]
    def one():
        i[0] += 1
        net(bags[i[0] % 32])
    with torch.no_grad():
        print(f"single slide net(X), 50k bf16: {wall(one, 320):.1f} us per call", flush=True)
if "wide" in which:
    base = torch.randn(256 * 2798, 512, device=dev, generator=g).to(torch.bfloat16)
    plan = F.VlfanBatchPlan(256, 12, 4, dev)
    plan.set_bags(F.BagSet([base[i * 2798:(i + 1) * 2798] for i in range(256)]))
    us = wall(lambda: plan.run(Q, T, ls, W, b, params_key=0), 100)
    usk = wall(plan.run_partial_only, 100)
    print(f"256 x 2798 bf16 per launch: whole {us:.1f} us ({us / 256:.3f} per bag), streaming kernel alone {usk:.1f} us", flush=True)
if "batch" in which:
    bags = [torch.randn(50_000, 512, device=dev, generator=g).to(torch.bfloat16) for _ in range(64)]
    plan = F.VlfanBatchPlan(64, 12, 4, dev)
    plan.set_bags(F.BagSet(bags))
    us = wall(lambda: plan.run(Q, T, ls, W, b, params_key=0), 40)
    usk = wall(plan.run_partial_only, 40)
    print(f"64 x 50k bf16 per launch: whole {us:.1f} us ({us / 64:.3f} per bag), streaming kernel alone {usk:.1f} us", flush=True)
