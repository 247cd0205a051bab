"""DeepMIL(gated_attention) forward + backward, 20 iterations (kernel census with rocprofv3 --kernel-trace --stats)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from vlsa_amd.deepmil import DeepMIL
n = int(sys.argv[1]) if len(sys.argv) > 1 else 50000
enc = DeepMIL(dim_in=512, dim_hid=256, num_cls=512, use_feat_proj=False, pooling="gated_attention", pred_head="Adapter").cuda().eval()
X = torch.randn(1, n, 512, device="cuda").to(torch.bfloat16)
G = torch.randn(1, 512, device="cuda")
for _ in range(20):
    for p in enc.parameters(): p.grad = None
    enc(X).backward(G)
torch.cuda.synchronize()
