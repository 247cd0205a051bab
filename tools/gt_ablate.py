"""Timing-only ablations of k_scores_tile (gated_scores_tile.hip), one library per value of -DVLSA_GT_ABL
(vlsa_amd/_lib/variants/libvlsa_gt<bits>.so; results of those libraries are WRONG by construction).
`python tools/gt_ablate.py build` compiles the variants (CPU container, after `python -m vlsa_amd.build`);
`python tools/gt_ablate.py` times every variant in its own process (VLSA_HIP_LIB) on the same box."""
import glob, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIBD = os.path.join(ROOT, "vlsa_amd", "_lib")
NAMES = {0: "the product", 16: "no activations", 32: "no MFMAs (operands still arrive)", 48: "neither"}
if os.environ.get("GT_ONLY"):
    NAMES = {int(b): NAMES.get(int(b), "?") for b in os.environ["GT_ONLY"].split(",")}
EXTRA = os.environ.get("GT_EXTRA", "").split()
if len(sys.argv) > 1 and sys.argv[1] == "build":
    os.makedirs(os.path.join(LIBD, "variants"), exist_ok=True)
    objs = [o for o in glob.glob(os.path.join(LIBD, "obj", "*.o")) if not o.endswith("gated_scores_tile.o")]
    for bits in NAMES:
        if bits == 0:
            continue
        o = f"/tmp/gt_abl{bits}.o"
        subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", f"-DVLSA_GT_ABL={bits}", *EXTRA, "-c",
                               os.path.join(ROOT, "vlsa_amd", "csrc", "gated_scores_tile.hip"), "-o", o])
        subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", *objs, o, "-o", os.path.join(LIBD, "variants", f"libvlsa_gt{bits}.so")])
    sys.exit(0)
if len(sys.argv) > 1 and sys.argv[1] == "child":
    sys.path.insert(0, ROOT)
    import torch
    from vlsa_amd import functional as F
    dev = "cuda"
    import gc; gc.collect(); gc.freeze()
    out = []
    for gated in (True, False):
        Wa = torch.randn(256, 512, device=dev) / 22; ba = torch.randn(256, device=dev) * 0.05
        Wg = torch.randn(256, 512, device=dev) / 22 if gated else None; bg = torch.randn(256, device=dev) * 0.05 if gated else None
        w2 = torch.randn(1, 256, device=dev) / 16; c = torch.randn(1, device=dev)
        fs = F.FusedAttnScores()
        for n in (393216, 50000):
            bags = [torch.randn(n, 512, device=dev).to(torch.bfloat16) for _ in range(4)]
            for i in range(40): fs(bags[i % 4], Wa, ba, Wg, bg, w2, c)
            torch.cuda.synchronize()
            us = 1e30
            for _ in range(3):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for i in range(60): fs(bags[i % 4], Wa, ba, Wg, bg, w2, c)
                e1.record(); torch.cuda.synchronize()
                us = min(us, e0.elapsed_time(e1) * 1e3 / 60)
            out.append(f"{'gated' if gated else 'ungated'} {n}: {us:7.1f}")
    print(" | ".join(out))
    sys.exit(0)
for bits, name in NAMES.items():
    lib = os.path.join(LIBD, "libvlsa_hip.so" if bits == 0 else f"variants/libvlsa_gt{bits}.so")
    if not os.path.exists(lib):
        continue
    r = subprocess.run([sys.executable, os.path.abspath(__file__), "child"], env=dict(os.environ, VLSA_HIP_LIB=lib), capture_output=True, text=True)
    print(f"ABL={bits!s:>4s} {name:36s}: us per bag  {r.stdout.strip().splitlines()[-1] if r.stdout.strip() else r.stderr[-200:]}", flush=True)
