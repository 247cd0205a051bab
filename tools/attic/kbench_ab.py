"""Same-box A/B of two builds of the library: VLSA_AB_LIB=<path of the other .so> python tools/attic/kbench_ab.py -- the streaming kernel
over 32 x 50k bf16 bags, alternating rounds (each library in its own process would not share the box's momentary state)."""
import ctypes, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 1 and sys.argv[1] == "child":
    sys.path.insert(0, ROOT)
    from vlsa_amd import _native
    if os.environ.get("VLSA_AB_LIB"):
        _native._LIB_PATH = os.environ["VLSA_AB_LIB"]
    import torch
    from vlsa_amd import functional as F
    dev = "cuda"
    B, n = 32, 50000
    base = torch.randn(B * n + 4096, 512, device=dev).to(torch.bfloat16)
    bags = [base[i * n:(i + 1) * n] for i in range(B)]
    Q = torch.randn(12, 512, device=dev); T = torch.randn(4, 512, device=dev)
    W = torch.randn(512, 512, device=dev) / 22; b = torch.randn(512, device=dev); ls = torch.tensor(4.03, device=dev)
    plan = F.VlfanBatchPlan(B, 12, 4, dev)
    plan.set_bags(bags); plan.run(Q, T, ls, W, b)
    for _ in range(60): plan.run_partial_only()
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(60)]
    for e0, e1 in ev:
        e0.record(); plan.run_partial_only(); e1.record()
    torch.cuda.synchronize()
    ts = sorted(e0.elapsed_time(e1) for e0, e1 in ev)
    avg = sum(ts) / len(ts) * 1e3
    print(f"{os.path.basename(_native._LIB_PATH):28s}: {avg:7.1f} us avg {ts[len(ts)//2]*1e3:7.1f} median {ts[0]*1e3:7.1f} min  {B*n*1024/avg/1e6:5.2f} TB/s avg")
else:
    other = os.environ.get("VLSA_AB_LIB_OTHER", os.path.join(ROOT, "vlsa_amd", "_lib", "libvlsa_hip_head.so"))
    for rnd in range(3):
        for lib in ("", other):
            env = dict(os.environ, VLSA_AB_LIB=lib)
            subprocess.run([sys.executable, __file__, "child"], env=env)
