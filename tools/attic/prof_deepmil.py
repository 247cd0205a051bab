import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from vlsa_amd.deepmil import DeepMIL
dev = "cuda"
m = DeepMIL(dim_in=512, dim_hid=256, use_feat_proj=False, pooling="gated_attention", pred_head="Adapter").to(dev).eval()
bags = [torch.randn(1, 50000, 512, device=dev).to(torch.bfloat16) for _ in range(8)]
with torch.no_grad():
    for i in range(100): m(bags[i % 8])
torch.cuda.synchronize()
