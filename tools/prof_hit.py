"""Host cost of a look-ahead HIT (net(X) served from a window): cProfile over one pass of 512 slide-sized resident items."""
import cProfile, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vlsa_amd.ingest import ResidentBags
from vlsa_amd.vlsa import VLSA
import gc; gc.collect(); gc.freeze()


class Items(torch.utils.data.Dataset):
    def __init__(self, n_items, n):
        g = torch.Generator().manual_seed(5)
        self.x = [torch.randn(n, 512, generator=g) for _ in range(8)]
        self.n_items = n_items
    def __len__(self): return self.n_items
    def __getitem__(self, i): return torch.tensor([i]), (self.x[i % 8], torch.zeros(1)), torch.ones(2)


cfg = dict(name="VLFAN", dim_in=512, use_feat_proj=False, query="Parameter", num_query=12, query_pooling="mean")
net = VLSA.from_modules(cfg, pretrained_text_features=torch.randn(4, 512)).cuda().eval()
n_items, n = 512, int(sys.argv[1]) if len(sys.argv) > 1 else 2798
rb = ResidentBags(Items(n_items, n), dtype=torch.bfloat16)
items = [torch.utils.data.default_collate([rb[i]])[1][0] for i in range(n_items)]
with torch.no_grad():
    for _ in range(2):
        for X in items: net(X)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5):
        for X in items: net(X)
    torch.cuda.synchronize()
    print(f"N={n}: {(time.perf_counter() - t0) / 5 / n_items * 1e6:.2f} us per net(X)")
    pr = cProfile.Profile(); pr.enable()
    for X in items: net(X)
    pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(14)
