"""Do the small tail kernels co-run with a resident persistent streaming kernel?  (timeline experiment for rocprofv3)"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from vlsa_amd import functional as F
dev = "cuda"
B, n = 32, 50000
bags = [torch.randn(n, 512, device=dev).to(torch.bfloat16) for _ in range(B)]
Q = torch.randn(12, 512, device=dev); T = torch.randn(4, 512, device=dev)
W = torch.randn(512, 512, device=dev) / 22; b = torch.randn(512, device=dev); ls = torch.tensor(4.03, device=dev)
plan = F.VlfanBatchPlan(B, 12, 4, dev); plan.set_bags(bags); plan.run(Q, T, ls, W, b)
pm = torch.zeros(32, 16, device=dev); pl = torch.ones(32, 16, device=dev); pacc = torch.randn(32, 12, 512, device=dev)
sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
torch.cuda.synchronize()
mode = sys.argv[1]
for rep in range(4):
    with torch.cuda.stream(sa):
        plan.run_partial_only()
        if mode == "two":
            pass
    if mode == "two":
        with torch.cuda.stream(sb):
            plan.run_partial_only()      # queued behind the first one (LDS-limited: one workgroup per CU)
    time.sleep(0.0001)
    with torch.cuda.stream(sb if mode == "one" else sa):
        F.vlfan_merge(pm, pl, pacc)      # small merge kernel, independent data
        F.normalize_rows(T)
    torch.cuda.synchronize()
