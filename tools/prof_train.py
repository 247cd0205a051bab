"""One 32-bag training step shape for rocprofv3: batched forward + backward of the aggregation (50k bf16 bags)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vlsa_amd import functional as F
dev = "cuda"
n = int(sys.argv[1]) if len(sys.argv) > 1 else 50000
dt = torch.float32 if (len(sys.argv) > 3 and sys.argv[3] == "fp32") else torch.bfloat16
base = torch.randn(32 * n, 512, device=dev).to(dt)
bags = [base[i * n:(i + 1) * n] for i in range(32)]
Q = torch.randn(12, 512, device=dev, requires_grad=True)
G = torch.randn(32, 12, 512, device=dev)
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 60
for i in range(iters):
    out = F.vlfan_cross_attention_bags(bags, Q)
    (out * G).sum().backward()
torch.cuda.synchronize()
