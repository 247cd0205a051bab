"""Same-box A/B of the ungated attention-score kernel's two workgroup shapes (VLSA_GS_HG2=0: 8 waves x 256 rows; default: four waves x
128 rows, three workgroups per CU), alternating child processes."""
import os, subprocess, sys
HERE = os.path.dirname(os.path.abspath(__file__))
CHILD = r'''
import sys, os
sys.path.insert(0, os.path.dirname(%r))
import torch
from vlsa_amd import functional as F
import gc; gc.collect(); gc.freeze()
dev = "cuda"
for gated in (False,):   # the gated module has one shape
    Wa = torch.randn(256, 512, device=dev) / 22; ba = torch.randn(256, device=dev) * 0.05
    Wg = torch.randn(256, 512, device=dev) / 22 if gated else None; bg = torch.randn(256, device=dev) * 0.05 if gated else None
    w2 = torch.randn(1, 256, device=dev) / 16; c = torch.randn(1, device=dev)
    fs = F.FusedAttnScores()
    out = []
    for n in (32768, 65536, 50000, 20000, 2798, 400000):
        bags = [torch.randn(n, 512, device=dev).bfloat16() for _ in range(4 if n > 100000 else 16)]
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
    print(f"HG2={os.environ.get('VLSA_GS_HG2','0')} gated={int(gated)}  " + "   ".join(out), flush=True)
''' % HERE
for rep in range(2):
    for hg2 in (sys.argv[1:] or ["0", "1"]):
        env = dict(os.environ, VLSA_GS_HG2=hg2)
        subprocess.run([sys.executable, "-c", CHILD], env=env, check=False)
