"""Timing-only ablations and schedule variants of k_vlfan_partial_f32_batch (vlfan_batch_f32.hip), one library per value of
-DVLSA_F32_ABL (vlsa_amd/_lib/variants/libvlsa_f32_<bits>.so; results of the ablation libraries 1 / 2 / 4 are WRONG by construction,
16 and 32 are re-orderings with the same results).  `python tools/f32_ablate.py build` compiles them (CPU container, after
`python -m vlsa_amd.build`); `python tools/f32_ablate.py` times every variant in its own process (VLSA_HIP_LIB) on the same box."""
import glob, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIBD = os.path.join(ROOT, "vlsa_amd", "_lib")
NAMES = {"old": "the previous build (copy it to variants/libvlsa_f32_old.so by hand)", 0: "the product", 1: "weighted sum: splits, no MFMAs", 2: "weighted sum: LDS reads only", 4: "no score MFMAs", 6: "neither contraction",
         16: "weighted sum term-major (independent MFMAs back to back)", 32: "four score accumulators", 48: "both re-orderings"}
if len(sys.argv) > 1 and sys.argv[1] == "build":
    os.makedirs(os.path.join(LIBD, "variants"), exist_ok=True)
    objs = [o for o in glob.glob(os.path.join(LIBD, "obj", "*.o")) if not o.endswith("vlfan_batch_f32.o")]
    for bits in NAMES:
        if bits in (0, "old"):
            continue
        o = f"/tmp/f32_abl{bits}.o"
        subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", f"-DVLSA_F32_ABL={bits}", "-c",
                               os.path.join(ROOT, "vlsa_amd", "csrc", "vlfan_batch_f32.hip"), "-o", o])
        subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", *objs, o, "-o", os.path.join(LIBD, "variants", f"libvlsa_f32_{bits}.so")])
    sys.exit(0)
if len(sys.argv) > 1 and sys.argv[1] == "child":
    sys.path.insert(0, ROOT)
    import torch
    from vlsa_amd import functional as F
    dev = "cuda"
    import gc; gc.collect(); gc.freeze()
    out = []
    for n, B in ((50000, 32), (10000, 32)):
        torch.cuda.empty_cache()
        bags = [torch.randn(n, 512, device=dev) for _ in range(B)]
        Q = torch.randn(12, 512, device=dev); T = torch.randn(4, 512, device=dev)
        W = torch.randn(512, 512, device=dev) / 22; b = torch.randn(512, device=dev); ls = torch.tensor(4.03, device=dev)
        plan = F.VlfanBatchPlan(B, 12, 4, dev); plan.set_bags(bags)
        for _ in range(20): plan.run(Q, T, ls, W, b)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        us = 1e30
        for _ in range(3):
            e0.record()
            for _ in range(30): plan.run_partial_only()
            e1.record(); torch.cuda.synchronize()
            us = min(us, e0.elapsed_time(e1) * 1e3 / 30)
        out.append(f"{n} x {B}: {us:7.1f} us = {B * n * 2048 / us / 8e6 * 100:5.1f} %")
    print(" | ".join(out))
    sys.exit(0)
for bits, name in NAMES.items():
    lib = os.path.join(LIBD, "libvlsa_hip.so" if bits == 0 else f"variants/libvlsa_f32_{bits}.so")
    if not os.path.exists(lib):
        continue
    r = subprocess.run([sys.executable, os.path.abspath(__file__), "child"], env=dict(os.environ, VLSA_HIP_LIB=lib), capture_output=True, text=True)
    print(f"ABL={bits!s:>3s} {name:58s}: {r.stdout.strip().splitlines()[-1] if r.stdout.strip() else r.stderr[-200:]}", flush=True)
