import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from vlsa_amd import functional as F
dev = "cuda"
def run(B, n):
    base = torch.randn(B * n + 4096, 512, device=dev).to(torch.bfloat16)
    bags = [base[i * n:(i + 1) * n] for i in range(B)]
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
    print(f"B={B} N={n} groups={plan.groups}: {avg:7.1f} us avg {ts[0]*1e3:7.1f} min  {B*n*1024/avg/1e6:5.2f} TB/s avg {B*n*1024/ts[0]/1e9:5.2f} best")
for rnd in range(2):
    run(32, 50000); run(32, 51200); run(32, 49152); run(8, 204800); run(64, 51200)
