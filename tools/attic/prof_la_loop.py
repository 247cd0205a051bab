"""cProfile of the handler-shaped eval loop (DataLoader + net(X) + softmax + 2 x .cpu()) over fp32 10k-patch resident bags with look-ahead"""
import sys, os, time, cProfile, pstats
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from vlsa_amd.vlsa import VLSA
from vlsa_amd.ingest import ResidentBags
import gc; gc.collect(); gc.freeze()
dev = "cuda"
P, K = 12, 4
cfg = dict(name="VLFAN", dim_in=512, use_feat_proj=False, query="Parameter", num_query=P, query_pooling="mean")
net = VLSA.from_modules(cfg, pretrained_text_features=torch.randn(K, 512)).to(dev).eval()
n, dt, n_items = int(sys.argv[1]) if len(sys.argv) > 1 else 10000, torch.float32 if (len(sys.argv) < 3 or sys.argv[2] == "fp32") else torch.bfloat16, 256


class _Items(torch.utils.data.Dataset):
    def __init__(self):
        g = torch.Generator().manual_seed(5)
        self.x = [torch.randn(n, 512, generator=g).to(dt) for _ in range(8)]
    def __len__(self):
        return n_items
    def __getitem__(self, i):
        return torch.Tensor([i]).to(torch.int), (self.x[i % 8].float(), torch.Tensor([0])), torch.Tensor([1.0, 1.0])


rb = ResidentBags(_Items(), dtype=dt)
loader = torch.utils.data.DataLoader(rb, batch_size=1, shuffle=False, num_workers=0)
for i in range(n_items):
    rb[i]
torch.cuda.synchronize()


def loop():
    out = []
    for data_idx, data_x, data_y in loader:
        X = data_x[0].cuda()
        raw, *_ = net(X)
        pred = torch.softmax(raw, dim=-1)
        out.append(raw.detach().cpu()); out.append(pred.detach().cpu())


with torch.no_grad():
    for la in (0, 64):
        net.lookahead_bags = la
        loop()
        t0 = time.perf_counter(); loop(); print(la, (time.perf_counter() - t0) / n_items * 1e6, "us/bag")
    pr = cProfile.Profile(); pr.enable(); loop(); pr.disable()
    pstats.Stats(pr).sort_stats("tottime").print_stats(12)
