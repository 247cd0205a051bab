"""Size sweep of the plain score launch (12k .. 400k bf16 rows, 16 bags in rotation): run once with VLSA_GS_TILE=1 (k_scores_tile_p at every
size) and once with VLSA_GS_TILE=0 (k_gated_scores): the dispatch policy gs_tile_use() in gated_scores.hip comes from this.
python tools/gs_sweep.py [gated]"""
import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from vlsa_amd import functional as F
dev = "cuda"
gated = len(sys.argv) > 1 and sys.argv[1] == "gated"
Wa = torch.randn(256, 512, device=dev) / 22; ba = torch.randn(256, device=dev) * 0.05
Wg = torch.randn(256, 512, device=dev) / 22 if gated else None; bg = torch.randn(256, device=dev) * 0.05 if gated else None
w2 = torch.randn(1, 256, device=dev) / 16; c = torch.randn(1, device=dev)
fs = F.FusedAttnScores()
out = []
for n in (12000, 16384, 20000, 24000, 28000, 32768, 40000, 50000, 60000, 70000, 85000, 100000, 150000, 200000, 400000):
    torch.cuda.empty_cache()
    bags = [torch.randn(n, 512, device=dev).to(torch.bfloat16) for _ in range(4 if n > 100000 else 16)]
    for i in range(40): fs(bags[i % len(bags)], Wa, ba, Wg, bg, w2, c)
    torch.cuda.synchronize(); us = 1e30
    for _ in range(4):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(80): fs(bags[i % len(bags)], Wa, ba, Wg, bg, w2, c)
        e1.record(); torch.cuda.synchronize(); us = min(us, e0.elapsed_time(e1) * 1e3 / 80)
    out.append(f"{n}:{us:.1f}")
print(" ".join(out))
