import os, sys, runpy
sys.argv = ["bench_step.py", "--only-batched"]
src = open(os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "tools", "bench_step.py")).read()
# run the setup part only (up to the loop over bag sizes), then profile the deferred handler loop
head = src[:src.index("for label, sizes in")]
g = {"__name__": "prof", "__file__": os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "tools", "bench_step.py")}
exec(compile(head, "bench_step_head", "exec"), g)
torch, net, opt, objective, dev, K = g["torch"], g["net"], g["opt"], g["objective"], g["dev"], g["K"]
gen = torch.Generator().manual_seed(0)
sizes = [int(x) for x in torch.randint(2000, 12000, (32,), generator=gen)]
bags = [torch.randn(n, 512, device=dev).to(torch.bfloat16) for n in sizes]
t = torch.randint(0, K, (32,), device=dev); e = (torch.rand(32, device=dev) < 0.45).float()
net.defer_training_calls = True
def hstep():
    logits = torch.cat([net(x[None])[0] for x in bags], dim=0)
    loss = objective(logits, t, e, net.get_logit_scale())
    opt.zero_grad(set_to_none=True); loss.backward(); opt.step()
for _ in range(10): hstep()
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
for _ in range(20): hstep()
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("tottime").print_stats(22)
