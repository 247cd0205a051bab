"""host profile of the evaluation pass / training epoch of tools/bench_epoch.py"""
import os, sys
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.argv = ["bench_epoch.py", "256"]
src = open(os.path.join(ROOT, "tools", "bench_epoch.py")).read()
head = src[:src.index("torch.cuda.synchronize(); t0 = time.perf_counter()\ntrain_epoch()")]
g = {"__name__": "prof", "__file__": os.path.join(ROOT, "tools", "bench_epoch.py")}
exec(compile(head, "bench_epoch_head", "exec"), g)
import cProfile, pstats, torch
g["train_epoch"](); g["net"].defer_training_calls = True; g["train_epoch"](); g["eval_pass"](); g["eval_pass"]()
for name in ("eval_pass", "train_epoch"):
    pr = cProfile.Profile(); pr.enable()
    for _ in range(3): g[name]()
    pr.disable(); torch.cuda.synchronize()
    print("=====", name)
    pstats.Stats(pr).sort_stats("tottime").print_stats(18)
