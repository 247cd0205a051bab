import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from vlsa_amd import functional as F
kern = int(sys.argv[1]) if len(sys.argv) > 1 else 3
qp = F.prepare_queries(torch.randn(12, 512, device="cuda"))
for n in (32, 4096, 8192, 16384, 32768, 50000, 100000, 200000):
    bags = [torch.randn(n, 512, device="cuda").to(torch.bfloat16) for _ in range(4)]
    for i in range(12):
        F.vlfan_partial(bags[i % 4], qp, kernel=kern)
    torch.cuda.synchronize()
