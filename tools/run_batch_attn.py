"""A few full batched runs WITH attention weights (streaming kernel with score stores + merge + in-place normalise) for
rocprofv3 PMC passes: python tools/run_batch_attn.py [B] [N]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vlsa_amd import functional as F
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
n = int(sys.argv[2]) if len(sys.argv) > 2 else 50000
dev = "cuda"
bags = [torch.randn(n, 512, device=dev).to(torch.bfloat16) for _ in range(B)]
Q = torch.randn(12, 512, device=dev); T = torch.randn(4, 512, device=dev)
W = torch.randn(512, 512, device=dev) / 22; b = torch.randn(512, device=dev); ls = torch.tensor(4.03, device=dev)
plan = F.VlfanBatchPlan(B, 12, 4, dev, want_attn=True)
plan.set_bags(bags)
for _ in range(24):
    plan.run(Q, T, ls, W, b)
torch.cuda.synchronize()
