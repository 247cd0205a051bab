"""Launch only the persistent multi-bag streaming kernel a few times (for rocprofv3 PMC passes)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vlsa_amd import functional as F
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
n = int(sys.argv[2]) if len(sys.argv) > 2 else 50000
dev = "cuda"
DT = torch.float32 if (len(sys.argv) > 4 and sys.argv[4] == "f32") else torch.bfloat16
bags = [torch.randn(n, 512, device=dev).to(DT) for _ in range(B)]
Q = torch.randn(12, 512, device=dev); T = torch.randn(4, 512, device=dev)
W = torch.randn(512, 512, device=dev) / 22; b = torch.randn(512, device=dev); ls = torch.tensor(4.03, device=dev)
R = int(sys.argv[3]) if len(sys.argv) > 3 else 0    # CUs without a streaming workgroup (bench.py default at N = 1)
plan = F.VlfanBatchPlan(B, 12, 4, dev, reserved_cus=R)
plan.set_bags(bags)
plan.run(Q, T, ls, W, b)
for _ in range(40):
    plan.run_partial_only()
torch.cuda.synchronize()
