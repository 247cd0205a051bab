import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from vlsa_amd import functional as F
torch.manual_seed(0)
for dt in (torch.bfloat16, torch.float32):
    for N in (1, 5, 33, 100, 1000, 5000, 50000):
        X = torch.randn(N, 512).to(dt).cuda(); Q = torch.randn(12, 512).cuda()
        o1, A1, _ = F.vlfan_aggregate(X, Q, kernel=1, want_attn=True)
        res = []
        for k in ((2, 3) if dt == torch.bfloat16 else (2,)):
            o2, A2, _ = F.vlfan_aggregate(X, Q, kernel=k, want_attn=True)
            o3, _, _ = F.vlfan_aggregate(X, Q, kernel=k, want_attn=False)
            res.append((k, (o1 - o2).abs().max().item(), (A1 - A2).abs().max().item(), (o3 - o2).abs().max().item()))
        print(dt, N, " ".join(f"k{k}: out {a:.2e} A {b:.2e} noattn {c:.2e}" for k, a, b, c in res))
