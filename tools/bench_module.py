"""Per-bag wall time of the drop-in module call net(X) in eval mode (what the reference's test loop does,
runner/vlsa_handler.py:315-345), bags resident in HBM, vs the batched call net.forward_bags(32 bags)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vlsa_amd.vlsa import VLSA
dev = "cuda"
import gc; gc.collect(); gc.freeze()   # torch's ~10^6 imported objects out of the collector's way: a gen-2 pass otherwise stalls one call by ~40 ms (profiles/README.md)
P, K = 12, 4
cfg = dict(name="VLFAN", dim_in=512, use_feat_proj=False, query="Parameter", num_query=P, query_pooling="mean")
net = VLSA.from_modules(cfg, pretrained_text_features=torch.randn(K, 512)).to(dev).eval()
for n, dt in ((50000, torch.bfloat16), (10000, torch.bfloat16), (2798, torch.bfloat16), (2798, torch.float32), (10000, torch.float32), (50000, torch.float32)):
    torch.cuda.empty_cache()   # every configuration allocates from fresh allocator segments
    base = torch.randn(32 * n, 512, device=dev).to(dt)
    bags = [base[i * n:(i + 1) * n][None] for i in range(32)]
    with torch.no_grad():
        for i in range(64): net(bags[i % 32])
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for i in range(320): net(bags[i % 32])
        torch.cuda.synchronize(); t1 = (time.perf_counter() - t0) / 320 * 1e6
        for i in range(10): net.forward_bags(bags)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for i in range(40): net.forward_bags(bags)
        torch.cuda.synchronize(); t2 = (time.perf_counter() - t0) / 40 / 32 * 1e6
        bags2 = bags + bags                      # 64 bags per call = one persistent launch
        for i in range(10): net.forward_bags(bags2)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for i in range(40): net.forward_bags(bags2)
        torch.cuda.synchronize(); t3 = (time.perf_counter() - t0) / 40 / 64 * 1e6
    print(f"N={n:6d} {str(dt)[6:]:9s}: net(X) {t1:7.1f} us/bag   forward_bags(32) {t2:7.2f} us/bag   forward_bags(64) {t3:7.2f} us/bag")
