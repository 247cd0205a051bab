"""Launch only the fused attention-score kernel (for rocprofv3 PMC passes)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vlsa_amd import functional as F
dev = "cuda"
n = int(sys.argv[1]) if len(sys.argv) > 1 else 400000
gated = not (len(sys.argv) > 2 and sys.argv[2] == "ungated")
Wa = torch.randn(256, 512, device=dev) / 22; ba = torch.randn(256, device=dev) * 0.05
Wg = torch.randn(256, 512, device=dev) / 22 if gated else None; bg = torch.randn(256, device=dev) * 0.05 if gated else None
w2 = torch.randn(1, 256, device=dev) / 16; c = torch.randn(1, device=dev)
X = torch.randn(n, 512, device=dev).to(torch.bfloat16)
fs = F.FusedAttnScores()
for _ in range(30): fs(X, Wa, ba, Wg, bg, w2, c)
torch.cuda.synchronize()
