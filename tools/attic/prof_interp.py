import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from vlsa_amd.vlsa import VLSA
from vlsa_amd.inference import calc_text_img_similarity
dev = "cuda"
cfg = dict(name="VLFAN", dim_in=512, use_feat_proj=False, query="Parameter", num_query=8, query_pooling="mean")
net = VLSA.from_modules(cfg, pretrained_text_features=torch.randn(8, 512)).to(dev).eval()
X = torch.randn(1, 50000, 512, device=dev).to(torch.bfloat16 if (len(sys.argv) < 2 or sys.argv[1] == "bf16") else torch.float32)
with torch.no_grad():
    for _ in range(5): calc_text_img_similarity(net, X)
torch.cuda.synchronize()
