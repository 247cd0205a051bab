import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vlsa_amd import functional as F
torch.manual_seed(0)
for N in (1, 5, 33, 100):
    X = torch.randn(N, 512).cuda(); Q = torch.randn(4, 512).cuda()
    o2, A2, _ = F.vlfan_aggregate(X, Q, kernel=2, want_attn=True)
    o1, A1, _ = F.vlfan_aggregate(X, Q, kernel=1, want_attn=True)
    print(N, "out diff", (o1 - o2).abs().max().item(), "A diff", (A1 - A2).abs().max().item())
