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
        from vlsa_amd.functional import BagSet
        bs = BagSet(bags2)                       # the same 64 bags, checked once: descriptor rows and bags-in-flight choice kept
        for i in range(10): net.forward_bags(bs)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for i in range(40): net.forward_bags(bs)
        torch.cuda.synchronize(); t4 = (time.perf_counter() - t0) / 40 / 64 * 1e6
        t5 = None
        if n <= 12000:                          # slide-sized bags: up to 256 per forward launch (round 4); distinct rows per bag
            wide = torch.randn(256 * n, 512, device=dev).to(dt)
            bs = BagSet([wide[i * n:(i + 1) * n] for i in range(256)])
            for i in range(10): net.forward_bags(bs)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for i in range(40): net.forward_bags(bs)
            torch.cuda.synchronize(); t5 = (time.perf_counter() - t0) / 40 / 256 * 1e6
            del bs, wide
    print(f"N={n:6d} {str(dt)[6:]:9s}: net(X) {t1:7.1f} us/bag   forward_bags(32) {t2:7.2f} us/bag   forward_bags(64) {t3:7.2f} us/bag   forward_bags(BagSet of 64) {t4:7.2f} us/bag"
          + (f"   forward_bags(BagSet of 256) {t5:7.2f} us/bag = {n / t5 / 1e3:5.2f} G patches/s" if t5 else ""))


# ---- the handler's evaluation loop (runner/vlsa_handler.py:315-345) over a ResidentBags dataset: net(X) once per bag, look-ahead
# windows of <= 64 bags behind it (vlsa_amd/vlsa.py::_lookahead).  Row 1: the model calls alone (as the rows above); row 2: the
# whole loop as the handler writes it (DataLoader(batch_size=1) + default_collate + .cuda() + softmax + two .cpu() per bag).
from vlsa_amd.ingest import ResidentBags


class _Items(torch.utils.data.Dataset):
    def __init__(self, n_items, n, dt):
        g = torch.Generator().manual_seed(5)
        self.x = [torch.randn(n, 512, generator=g).to(dt) for _ in range(8)]
        self.n_items = n_items

    def __len__(self):
        return self.n_items

    def __getitem__(self, i):
        return torch.Tensor([i]).to(torch.int), (self.x[i % 8].float(), torch.Tensor([0])), torch.Tensor([1.0, 1.0])


for n, dt, n_items in ((50000, torch.bfloat16, 256), (2798, torch.bfloat16, 512), (10000, torch.float32, 256)):
    torch.cuda.empty_cache()
    rb = ResidentBags(_Items(n_items, n, dt), dtype=dt)
    loader = torch.utils.data.DataLoader(rb, batch_size=1, shuffle=False, num_workers=0)
    items = [torch.utils.data.default_collate([rb[i]])[1][0] for i in range(n_items)]       # uploads; tagged [1, N, 512] views
    res = {}
    calls = []
    orig = net._forward_bags_fused
    net._forward_bags_fused = lambda bags, tf, **kw: (calls.append(len(bags)), orig(bags, tf, **kw))[1]

    def whole_loop():
        out = []
        for data_idx, data_x, data_y in loader:
            X = data_x[0].cuda()
            raw, *_ = net(X)
            pred = torch.softmax(raw, dim=-1)
            out.append(raw.detach().cpu()); out.append(pred.detach().cpu())

    with torch.no_grad():
        for la in (0, 64):          # the whole loop first (every iteration synchronises), then the model calls alone
            net.lookahead_bags = la
            whole_loop()
            calls.clear()
            gc.collect()            # (a generation-2 pass inside the timed loop would be ~40 ms: profiles/README.md)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            whole_loop(); whole_loop()
            t_loop = (time.perf_counter() - t0) / 2 / n_items * 1e6
            res[la] = [0.0, t_loop, list(calls)]
        for la in (0, 64):
            net.lookahead_bags = la
            for _ in range(2):
                for X in items: net(X)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(3):
                for X in items: net(X)
            torch.cuda.synchronize(); res[la][0] = (time.perf_counter() - t0) / 3 / n_items * 1e6
    print(f"N={n:6d} {str(dt)[6:]:9s}: handler eval loop over {n_items} resident items: net(X) {res[64][0]:6.2f} us/bag with look-ahead "
          f"({n / res[64][0] / 1e3:.2f} G patches/s), {res[0][0]:6.2f} without;   whole loop incl. DataLoader + softmax + 2 x .cpu(): "
          f"{res[64][1]:6.1f} us/bag with, {res[0][1]:6.1f} without  (windows of the last two passes: {res[64][2]})")
    net._forward_bags_fused = orig
    del rb, loader, items
