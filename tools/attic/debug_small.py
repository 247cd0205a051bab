import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from vlsa_amd import functional as F
torch.set_printoptions(linewidth=200, precision=4, sci_mode=False)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1
X = (torch.arange(512).float()[None, :] * 0.25 + torch.arange(N).float()[:, None] * 1000).cuda()
Q = torch.zeros(4, 512); Q[0, 0] = 1; Q[1, 1] = 1; Q[2, 130] = 1; Q[3, 511] = 1
qp = F.prepare_queries(Q.cuda())
for kern in (2, 1):
    pm, pl, pacc, sc = F.vlfan_partial(X, qp, kernel=kern, want_scores=True)
    torch.cuda.synchronize()
    print("kernel", kern, "pm", pm[0, :4].cpu(), "pl", pl[0, :4].cpu())
    print(" scores", sc[:, :4].cpu())
    a = pacc[0].cpu()
    print(" acc row0 [0:20]", a[0, :20])
    print(" acc row0 [120:140]", a[0, 120:140])
    print(" acc row1 [500:512]", a[1, 500:512])
