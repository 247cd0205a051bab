import cProfile, os, pstats, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from vlsa_amd.vlsa import VLSA
import gc; gc.collect(); gc.freeze()
cfg = dict(name="VLFAN", dim_in=512, use_feat_proj=False, query="Parameter", num_query=12, query_pooling="mean")
net = VLSA.from_modules(cfg, pretrained_text_features=torch.randn(4, 512)).cuda().eval()
base = torch.randn(64 * 2798, 512, device="cuda").to(torch.bfloat16)
bags = [base[i * 2798:(i + 1) * 2798] for i in range(64)]
with torch.no_grad():
    for _ in range(20): net.forward_bags(bags)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(100): net.forward_bags(bags)
    torch.cuda.synchronize(); print(f"{(time.perf_counter() - t0) / 100 / 64 * 1e6:.2f} us per bag (list of 64 x 2798)")
    pr = cProfile.Profile(); pr.enable()
    for _ in range(50): net.forward_bags(bags)
    pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(12)
