"""Launch only the streaming kernel a few times (for rocprofv3 PMC passes)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from vlsa_amd import functional as F
n = int(sys.argv[1]) if len(sys.argv) > 1 else 50000
dt = torch.bfloat16 if (len(sys.argv) < 3 or sys.argv[2] == "bf16") else torch.float32
bags = [torch.randn(n, 512, device="cuda").to(dt) for _ in range(8)]
qp = F.prepare_queries(torch.randn(12, 512, device="cuda"))
for i in range(16):
    F.vlfan_partial(bags[i % 8], qp, kernel=2)
torch.cuda.synchronize()
