import sys, os, cProfile, pstats
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from vlsa_amd import functional as F
dev = "cuda"
n = 2798
X = torch.randn(n, 512, device=dev).to(torch.bfloat16)
Wa = (torch.randn(256, 512, device=dev) / 22).requires_grad_(True); ba = (torch.randn(256, device=dev) * 0.05).requires_grad_(True)
Wg = (torch.randn(256, 512, device=dev) / 22).requires_grad_(True); bg = (torch.randn(256, device=dev) * 0.05).requires_grad_(True)
w2 = (torch.randn(1, 256, device=dev) / 16).requires_grad_(True); c = torch.randn(1, device=dev).requires_grad_(True)
fs = F.FusedAttnScores()
G = torch.randn(n, device=dev)
def it():
    a = F.attn_scores_autograd(X, fs, Wa, ba, Wg, bg, w2, c)
    a.backward(G)
for _ in range(20): it()
torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable()
for _ in range(200): it()
torch.cuda.synchronize()
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(22)
