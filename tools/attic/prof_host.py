"""Host-side profile (cProfile) of one batched module call: python tools/prof_host.py zeroshot|deepmil|vlfan [N]"""
import sys, os, time, cProfile, pstats
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from vlsa_amd.vlsa import VLSA
dev = "cuda"
which = sys.argv[1] if len(sys.argv) > 1 else "zeroshot"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 2798
K = 4
if which == "zeroshot":
    cfg = dict(name="FeatMIL", pooling="logit_top10")
elif which == "deepmil":
    cfg = dict(name="DeepMIL", dim_in=512, dim_hid=256, num_cls=512, use_feat_proj=False, pooling="gated_attention", pred_head="Adapter")
else:
    cfg = dict(name="VLFAN", dim_in=512, use_feat_proj=False, query="Parameter", num_query=12, query_pooling="mean")
net = VLSA.from_modules(cfg, pretrained_text_features=torch.randn(K, 512)).to(dev).eval()
base = torch.randn(32 * n, 512, device=dev).to(torch.bfloat16)
bags = [base[i * n:(i + 1) * n] for i in range(32)]
with torch.no_grad():
    for i in range(20): net.forward_bags(bags)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(100): net.forward_bags(bags)
    th = (time.perf_counter() - t0) / 100 * 1e6
    torch.cuda.synchronize()
    tw = (time.perf_counter() - t0) / 100 * 1e6
    print(which, n, "host-only per call us", round(th, 1), "wall", round(tw, 1))
    pr = cProfile.Profile(); pr.enable()
    for i in range(200): net.forward_bags(bags)
    pr.disable(); torch.cuda.synchronize()
    pstats.Stats(pr).sort_stats("tottime").print_stats(18)
