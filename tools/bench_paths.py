"""Wall time per bag of every module-level path (eval, bag resident), to spot slow kernels: N = 50k."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vlsa_amd.vlsa import VLSA
from vlsa_amd.inference import calc_text_img_similarity
dev = "cuda"
import gc; gc.collect(); gc.freeze()   # torch's ~10^6 imported objects out of the collector's way: a gen-2 pass otherwise stalls one call by ~40 ms (profiles/README.md)
K, P, n = 4, 12, 50000
def timeit(f, reps=100):
    with torch.no_grad():
        for _ in range(20): f()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(reps): f()
        torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e6
T = torch.randn(K, 512)
for dt in (torch.bfloat16, torch.float32):
    torch.cuda.empty_cache()   # every configuration allocates from fresh allocator segments
    X = torch.randn(1, n, 512, device=dev).to(dt)
    rows = []
    for qp in ("mean", "max", "weight", "attention", "gated_attention"):
        for gq in (False, True):
            cfg = dict(name="VLFAN", dim_in=512, dim_hid=256, use_feat_proj=False, query="Parameter", num_query=P, gated_query=gq, query_pooling=qp)
            net = VLSA.from_modules(cfg, pretrained_text_features=T).to(dev).eval()
            rows.append((f"VLFAN pool={qp} gated_query={gq}", timeit(lambda: net(X))))
            if qp in ("mean", "attention") and not gq:
                rows.append((f"  + mil_encoder(X, ret_with_attn=True)", timeit(lambda: net.mil_encoder(X, ret_with_attn=True))))
    for pool in ("mean", "max"):
        net = VLSA.from_modules(dict(name="FeatMIL", dim_in=512, pooling=pool), pretrained_text_features=T).to(dev).eval()
        rows.append((f"FeatMIL pool={pool}", timeit(lambda: net(X))))
    for pool in ("mean", "max", "attention", "gated_attention"):
        net = VLSA.from_modules(dict(name="DeepMIL", dim_in=512, dim_hid=256, num_cls=512, use_feat_proj=False, pooling=pool, pred_head="Adapter"),
                   pretrained_text_features=T).to(dev).eval()
        rows.append((f"DeepMIL pool={pool} Adapter", timeit(lambda: net(X))))
    cfg = dict(name="VLFAN", dim_in=512, use_feat_proj=False, query="Parameter", num_query=8, query_pooling="mean")
    net = VLSA.from_modules(cfg, pretrained_text_features=torch.randn(8, 512)).to(dev).eval()
    rows.append(("calc_text_img_similarity (P=8, K=8)", timeit(lambda: calc_text_img_similarity(net, X), reps=10)))
    print(f"---- {str(dt)[6:]} N={n}")
    for name, us in rows:
        print(f"{name:48s} {us:9.1f} us/bag")
