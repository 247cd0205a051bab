"""Same-box A/B of the attention-score kernel: variants (environment settings, e.g. another build through VLSA_HIP_LIB, or
VLSA_GS_HG2=0 = the ungated module on the 8-wave shape) in alternating child processes."""
import os, subprocess, sys
HERE = os.path.dirname(os.path.abspath(__file__))
CHILD = r'''
import sys, os
sys.path.insert(0, os.path.dirname(%r))
import torch
from vlsa_amd import functional as F
import gc; gc.collect(); gc.freeze()
dev = "cuda"
for gated in ((False,) if os.environ.get('ONLY_UNGATED') else (True, False)):
    Wa = torch.randn(256, 512, device=dev) / 22; ba = torch.randn(256, device=dev) * 0.05
    Wg = torch.randn(256, 512, device=dev) / 22 if gated else None; bg = torch.randn(256, device=dev) * 0.05 if gated else None
    w2 = torch.randn(1, 256, device=dev) / 16; c = torch.randn(1, device=dev)
    fs = F.FusedAttnScores()
    out = []
    for n in [int(v) for v in os.environ.get('NS', '32768,65536,50000,20000,2798,400000').split(',')]:
        bags = [torch.randn(n, 512, device=dev).to(torch.float32 if os.environ.get('DTYPE') == 'f32' else torch.bfloat16) for _ in range(4 if n > 100000 else 16)]
        for i in range(40): fs(bags[i %% len(bags)], Wa, ba, Wg, bg, w2, c)
        torch.cuda.synchronize()
        us = 1e30
        for _ in range(4):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for i in range(100): fs(bags[i %% len(bags)], Wa, ba, Wg, bg, w2, c)
            e1.record(); torch.cuda.synchronize()
            us = min(us, e0.elapsed_time(e1) * 1e3 / 100)
        out.append(f"{n}: {us:7.2f}")
        del bags; torch.cuda.empty_cache()
    print(f"{os.environ.get('VARIANT','-'):28s} gated={int(gated)}  " + "   ".join(out), flush=True)
''' % HERE
# every argument is one variant: comma-separated NAME=VALUE environment settings ("-" = none), e.g.
#   python tools/kbench_gated_ab.py - VLSA_GS_HG2=0          (four-wave vs 8-wave ungated shape)
#   python tools/kbench_gated_ab.py - VLSA_HIP_LIB=/root/repo/vlsa_amd/_lib/libvlsa_hip_alt.so
for rep in range(2):
    for spec in (sys.argv[1:] or ["-", "VLSA_GS_HG2=0"]):
        env = dict(os.environ, VARIANT=os.path.basename(spec))
        if spec != "-":
            env.update(kv.split("=", 1) for kv in spec.split(","))
        subprocess.run([sys.executable, "-c", CHILD], env=env, check=False)
