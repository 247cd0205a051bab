import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vlsa_amd.vlsa import VLSA
from vlsa_amd.inference import calc_text_img_similarity
dev = "cuda"
cfg = dict(name="VLFAN", dim_in=512, use_feat_proj=False, query="Parameter", num_query=8, query_pooling="mean")
net = VLSA.from_modules(cfg, pretrained_text_features=torch.randn(8, 512)).to(dev).eval()
for rep in range(2):
    for dt in (torch.bfloat16, torch.float32):
        X = torch.randn(1, 50000, 512, device=dev).to(dt)
        with torch.no_grad():
            for _ in range(3): calc_text_img_similarity(net, X)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(10): calc_text_img_similarity(net, X)
            torch.cuda.synchronize()
        print(dt, (time.perf_counter() - t0) / 10 * 1e3, "ms")
