"""Timing-only ablations of k_gated_scores, one library per bit set (vlsa_amd/_lib/variants/libvlsa_abl<bits>.so, built with
-DVLSA_GS_ABL=<bits>; results of those libraries are WRONG by construction): 400k- and 50k-patch bf16 bags, gated and ungated.
`python tools/gs_ablate.py` runs every variant in its own process (VLSA_HIP_LIB) on the same box.

Building the variants (in the CPU container, after `python -m vlsa_amd.build`):
    cd vlsa_amd/csrc; mkdir -p ../_lib/variants
    for a in 2 4 6 32 1 8; do
      hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DVLSA_GS_ABL=$a -c gated_scores.hip -o /tmp/gs_abl$a.o
      hipcc --offload-arch=gfx950 -shared -fPIC $(ls ../_lib/obj/*.o | grep -v gated_scores) /tmp/gs_abl$a.o -o ../_lib/variants/libvlsa_abl$a.so
    done
(the variant libraries are not kept in the tree: 2.3 MB each)."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
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
NAMES = {0: "the product", 2: "no weight loads after step 1", 4: "no X loads / publication", 6: "neither weights nor X", 32: "no MFMAs (operands still arrive)",
         1: "no A-fragment reads from LDS", 8: "no per-step barrier"}
for bits, name in NAMES.items():
    lib = os.path.join(ROOT, "vlsa_amd", "_lib", "libvlsa_hip.so" if bits == 0 else f"variants/libvlsa_{bits}.so" if isinstance(bits, str) else f"variants/libvlsa_abl{bits}.so")
    if not os.path.exists(lib):
        continue
    r = subprocess.run([sys.executable, os.path.abspath(__file__), "child"], env=dict(os.environ, VLSA_HIP_LIB=lib), capture_output=True, text=True)
    print(f"ABL={bits!s:>4s} {name:36s}: us per bag  {r.stdout.strip().splitlines()[-1] if r.stdout.strip() else r.stderr[-200:]}", flush=True)
