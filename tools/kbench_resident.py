"""The streaming kernel on cache-resident rows: all 32 bags of a launch are the SAME tensor, so that what bounds an iteration with
the memory system out of the way (the dependent chain LDS read -> MFMA -> barrier -> exchange -> barrier -> exp2 -> MFMA) shows."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vlsa_amd import functional as F
dev = "cuda"


def run(B, n, alias):
    base = torch.randn((1 if alias else B) * n + 4096, 512, device=dev).to(torch.bfloat16)
    bags = [base[0:n] if alias else base[i * n:(i + 1) * n] for i in range(B)]
    Q = torch.randn(12, 512, device=dev); T = torch.randn(4, 512, device=dev)
    W = torch.randn(512, 512, device=dev) / 22; b = torch.randn(512, device=dev); ls = torch.tensor(4.03, device=dev)
    plan = F.VlfanBatchPlan(B, 12, 4, dev)
    plan.set_bags(bags); plan.run(Q, T, ls, W, b)
    for _ in range(30): plan.run_partial_only()
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(30)]
    for e0, e1 in ev:
        e0.record(); plan.run_partial_only(); e1.record()
    torch.cuda.synchronize()
    ts = sorted(e0.elapsed_time(e1) for e0, e1 in ev)
    avg = sum(ts) / len(ts) * 1e3
    print(f"B={B} N={n} {'ONE tensor for all bags' if alias else 'distinct bags':24s} groups={plan.groups}: {avg:7.1f} us avg {ts[0]*1e3:7.1f} min  "
          f"{B*n*1024/avg/1e6:5.2f} TB/s (algorithmic) avg")


for rnd in range(2):
    run(32, 50000, False); run(32, 50000, True); run(32, 2048, True); run(32, 16384, True)
