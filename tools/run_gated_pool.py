"""Launch scores + attention pooling of rotating bf16 bags (for rocprofv3 PMC passes): `one` = vlsa_gated_scores_pool (one launch of
k_scores_tile_p<.., POOL> + the fold), `two` = score kernel, then pooling partials + merge.  python tools/run_gated_pool.py N one|two"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vlsa_amd import functional as F
dev = "cuda"
n = int(sys.argv[1]) if len(sys.argv) > 1 else 50000
one = not (len(sys.argv) > 2 and sys.argv[2] == "two")
Wa = torch.randn(256, 512, device=dev) / 22; ba = torch.randn(256, device=dev) * 0.05
Wg = torch.randn(256, 512, device=dev) / 22; bg = torch.randn(256, device=dev) * 0.05
w2 = torch.randn(1, 256, device=dev) / 16; c = torch.randn(1, device=dev)
bags = [torch.randn(n, 512, device=dev).to(torch.bfloat16) for _ in range(12)]     # 12 x 51 MB at 50k: more than the 256 MB MALL
fs = F.FusedAttnScores()
for i in range(36):
    X = bags[i % 12]
    if one:
        fs.scores_and_pool(X, Wa, ba, Wg, bg, w2, c)
    else:
        F.scored_pool(X, fs(X, Wa, ba, Wg, bg, w2, c))
torch.cuda.synchronize()
