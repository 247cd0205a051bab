"""Does the streaming kernel's bandwidth depend on the DATA (the chip clocks to its power budget: zeroed operands toggle fewer
wires)?  32 x 50k bf16 bags of N(0,1) values / of zeros / of one repeated row, same process, interleaved rounds."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from vlsa_amd import functional as F
dev = "cuda"
B, n = 32, 50000
Q = torch.randn(12, 512, device=dev); T = torch.randn(4, 512, device=dev)
W = torch.randn(512, 512, device=dev) / 22; b = torch.randn(512, device=dev); ls = torch.tensor(4.03, device=dev)
kinds = {}
kinds["normal"] = torch.randn(B * n, 512, device=dev).to(torch.bfloat16)
kinds["zeros"] = torch.zeros(B * n, 512, device=dev, dtype=torch.bfloat16)
kinds["one row"] = torch.randn(1, 512, device=dev).to(torch.bfloat16).repeat(B * n, 1).contiguous()
plans = {}
for k, base in kinds.items():
    p = F.VlfanBatchPlan(B, 12, 4, dev)
    p.set_bags([base[i * n:(i + 1) * n] for i in range(B)]); p.run(Q, T, ls, W, b)
    plans[k] = p
res = {k: [] for k in kinds}
for rnd in range(4):
    for k, p in plans.items():
        for _ in range(40): p.run_partial_only()
        torch.cuda.synchronize()
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(40)]
        for e0, e1 in ev:
            e0.record(); p.run_partial_only(); e1.record()
        torch.cuda.synchronize()
        ts = [e0.elapsed_time(e1) * 1e3 for e0, e1 in ev]
        res[k].append(sum(ts) / len(ts))
for k, v in res.items():
    avg = sum(v) / len(v)
    print(f"{k:8s}: {avg:7.1f} us avg over {len(v)} rounds ({', '.join(f'{x:.1f}' for x in v)})  = {B * n * 1024 / avg / 1e6:5.2f} TB/s")
