"""The handler's evaluation loop over 256 distinct resident 50k x 512 bf16 bags (bench.py's `eval_loop_lookahead` leg) alone, for a
kernel trace: `rocprofv3 --kernel-trace --output-format csv -d <dir> -- python tools/prof_eval_loop.py`; then
`python tools/prof_eval_loop.py gaps <kernel_trace.csv>` prints the busy / idle split of the GPU timeline of the last pass."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 2 and sys.argv[1] == "gaps":
    import csv
    rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(sys.argv[2]))]
    rows.sort()
    rows = rows[-int(len(rows) * 0.3):]                  # the last pass
    t0, t1 = rows[0][0], max(r[1] for r in rows)
    busy, cur_end, gaps = 0, rows[0][0], []
    for s, e, n in rows:
        if s > cur_end:
            gaps.append((s - cur_end, n))
            cur_end = s
        if e > cur_end:
            busy += e - cur_end
            cur_end = e
    print(f"span {(t1 - t0) / 1e3:.0f} us, busy {busy / 1e3:.0f} us ({busy / (t1 - t0):.1%}); {len(gaps)} gaps, the largest:")
    for g, n in sorted(gaps, reverse=True)[:12]:
        print(f"  {g / 1e3:8.1f} us before {n[:60]}")
    sys.exit(0)
import torch
from vlsa_amd.ingest import ResidentBags
from vlsa_amd.vlsa import VLSA
import gc; gc.collect(); gc.freeze()
rows, n_items = 50_000, 256


class Items(torch.utils.data.Dataset):
    def __init__(self):
        g = torch.Generator().manual_seed(321)
        self.base = torch.randn(rows + 4 * n_items, 512, generator=g).to(torch.bfloat16)
    def __len__(self): return n_items
    def __getitem__(self, i): return torch.tensor([i]), (self.base[4 * i:4 * i + rows], torch.zeros(1)), torch.ones(2)


cfg = dict(name="VLFAN", dim_in=512, use_feat_proj=False, query="Parameter", num_query=12, query_pooling="mean")
net = VLSA.from_modules(cfg, pretrained_text_features=torch.randn(4, 512)).cuda().eval()
rb = ResidentBags(Items(), dtype=torch.bfloat16)
items = [torch.utils.data.default_collate([rb[i]])[1][0] for i in range(n_items)]
with torch.no_grad():
    for _ in range(2):
        for X in items: net(X)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(3):
        for X in items: net(X)
    torch.cuda.synchronize()
print(f"{(time.perf_counter() - t0) / 3 / n_items * 1e6:.2f} us per bag")
