import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vlsa_amd import functional as F
dev = "cuda"
gated = True
Wa = torch.randn(256, 512, device=dev) / 22; ba = torch.randn(256, device=dev) * 0.05
Wg = torch.randn(256, 512, device=dev) / 22; bg = torch.randn(256, device=dev) * 0.05
w2 = torch.randn(1, 256, device=dev) / 16; c = torch.randn(1, device=dev)
fs = F.FusedAttnScores()
n = int(sys.argv[1])
bags = [torch.randn(n, 512, device=dev).to(torch.bfloat16) for _ in range(8)]
for i in range(40): fs(bags[i % 8], Wa, ba, Wg, bg, w2, c)
torch.cuda.synchronize()
us = 1e30
for _ in range(4):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(100): fs(bags[i % 8], Wa, ba, Wg, bg, w2, c)
    e1.record(); torch.cuda.synchronize()
    us = min(us, e0.elapsed_time(e1) * 1e3 / 100)
print(f"N={n} rows={os.environ.get('VLSA_GS_ROWS','default')}: {us:.1f} us")
