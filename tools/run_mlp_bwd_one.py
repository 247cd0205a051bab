"""Only the gated attention-score backward kernel, 12 launches (PMC passes): python tools/run_mlp_bwd_one.py [N] [mode]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vlsa_amd import functional as F
n = int(sys.argv[1]) if len(sys.argv) > 1 else 50000
mode = sys.argv[2] if len(sys.argv) > 2 else "gated"
dev = "cuda"
X = torch.randn(n, 512, device=dev).to(torch.bfloat16)
if mode in ("gated", "tanh"):
    gated = mode == "gated"
    Wa = (torch.randn(256, 512, device=dev) / 22).requires_grad_(True); ba = (torch.randn(256, device=dev) * 0.05).requires_grad_(True)
    Wg = (torch.randn(256, 512, device=dev) / 22).requires_grad_(True) if gated else None
    bg = (torch.randn(256, device=dev) * 0.05).requires_grad_(True) if gated else None
    w2 = (torch.randn(1, 256, device=dev) / 16).requires_grad_(True); c = torch.randn(1, device=dev).requires_grad_(True)
    fs = F.FusedAttnScores()
    G = torch.randn(n, device=dev)
    for _ in range(12):
        F.attn_scores_autograd(X, fs, Wa, ba, Wg, bg, w2, c).backward(G)
else:
    W = (torch.randn(512, 512, device=dev) / 22).requires_grad_(True); b = torch.zeros(512, device=dev, requires_grad=True)
    gm = torch.ones(512, device=dev, requires_grad=True); bt = torch.zeros(512, device=dev, requires_grad=True)
    fp = F.FusedFeatProjecter()
    G = torch.randn(n, 512, device=dev)
    for _ in range(12):
        fp.autograd(X, W, b, gm, bt, 1e-5).backward(G)
torch.cuda.synchronize()
