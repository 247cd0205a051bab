"""Backward kernels of the N-sized layers (mlp_backward.hip, vlfan_dx.hip): time per bag through the autograd functions, and the
DeepMIL / VLFAN-with-projecter modules forward vs forward + backward."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vlsa_amd import functional as F
dev = "cuda"
import gc; gc.collect(); gc.freeze()


def timeit(fn, reps=40, chunks=3):
    for _ in range(10): fn()
    torch.cuda.synchronize()
    best = 1e30
    for _ in range(chunks):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps): fn()
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1e3 / reps)
    return best


for gated in (True, False):
    Wa = (torch.randn(256, 512, device=dev) / 22).requires_grad_(True); ba = (torch.randn(256, device=dev) * 0.05).requires_grad_(True)
    Wg = (torch.randn(256, 512, device=dev) / 22).requires_grad_(True) if gated else None
    bg = (torch.randn(256, device=dev) * 0.05).requires_grad_(True) if gated else None
    w2 = (torch.randn(1, 256, device=dev) / 16).requires_grad_(True); c = torch.randn(1, device=dev).requires_grad_(True)
    fs = F.FusedAttnScores()
    for n, dt in ((50000, torch.bfloat16), (10000, torch.bfloat16), (2798, torch.bfloat16), (50000, torch.float32), (10000, torch.float32)):
        X = torch.randn(n, 512, device=dev).to(dt)
        G = torch.randn(n, device=dev)
        fwd = timeit(lambda: fs(X, Wa, ba, Wg, bg, w2, c))
        def both():
            a = F.attn_scores_autograd(X, fs, Wa, ba, Wg, bg, w2, c)
            a.backward(G)
        tot = timeit(both)
        fl = 2.0 * n * 512 * 256 * (2 if gated else 1)
        print(f"attn scores gated={gated} {str(dt)[6:]:8s} N={n:6d}: fwd {fwd:7.1f} us, fwd+bwd {tot:7.1f} us (bwd {tot - fwd:7.1f} us = "
              f"{2 * fl / (tot - fwd) / 1e6:6.1f} TFLOP/s algorithmic [2x fwd FLOP], x{tot / fwd:4.2f} of fwd)")

W = (torch.randn(512, 512, device=dev) / 22).requires_grad_(True); b = torch.zeros(512, device=dev, requires_grad=True)
gm = torch.ones(512, device=dev, requires_grad=True); bt = torch.zeros(512, device=dev, requires_grad=True)
fp = F.FusedFeatProjecter()
for n, dt in ((50000, torch.bfloat16), (10000, torch.bfloat16), (2798, torch.bfloat16), (50000, torch.float32)):
    X = torch.randn(n, 512, device=dev).to(dt)
    G = torch.randn(n, 512, device=dev)
    fwd = timeit(lambda: fp(X, W, b, gm, bt, 1e-5))
    def both():
        y = fp.autograd(X, W, b, gm, bt, 1e-5)
        y.backward(G)
    tot = timeit(both)
    print(f"feat projecter {str(dt)[6:]:8s} N={n:6d}: fwd {fwd:7.1f} us, fwd+bwd {tot:7.1f} us (bwd {tot - fwd:7.1f} us, x{tot / fwd:4.2f} of fwd)")

Q = torch.randn(12, 512, device=dev, requires_grad=True)
for n in (50000, 10000, 2798):
    X = torch.randn(n, 512, device=dev)
    Xg = X.clone().requires_grad_(True)
    G = torch.randn(12, 512, device=dev)
    def q_only():
        o, _ = F.vlfan_cross_attention(X, Q); o.backward(G)
    def with_dx():
        o, _ = F.vlfan_cross_attention(Xg, Q); o.backward(G)
    t0, t1 = timeit(q_only), timeit(with_dx)
    print(f"cross attention fp32 N={n:6d}: fwd+bwd(dQ) {t0:7.1f} us, + dX {t1:7.1f} us (dX kernel ~{t1 - t0:6.1f} us = "
          f"{n * 4096 / (t1 - t0) / 1e6:5.2f} TB/s of 4 KB/row)")

from vlsa_amd.deepmil import DeepMIL, VLFAN
for pooling in ("gated_attention", "attention"):
    enc = DeepMIL(dim_in=512, dim_hid=256, num_cls=512, use_feat_proj=False, pooling=pooling, pred_head="Adapter").cuda().eval()
    for n, dt in ((50000, torch.bfloat16), (2798, torch.bfloat16), (50000, torch.float32)):
        X = torch.randn(1, n, 512, device=dev).to(dt)
        G = torch.randn(1, 512, device=dev)
        def f():
            with torch.no_grad(): enc(X)
        def fb():
            enc(X).backward(G)
        t0, t1 = timeit(f), timeit(fb)
        print(f"DeepMIL({pooling}) {str(dt)[6:]:8s} N={n:6d}: forward {t0:7.1f} us, forward + backward {t1:7.1f} us = x{t1 / t0:4.2f}")
enc = VLFAN(dim_in=512, use_feat_proj=True, query="Parameter", num_query=12, query_pooling="mean", pred_head="default").cuda().train()
for n, dt in ((50000, torch.bfloat16), (2798, torch.bfloat16)):
    X = torch.randn(1, n, 512, device=dev).to(dt)
    G = torch.randn(1, 512, device=dev)
    def f():
        with torch.no_grad(): enc(X)
    def fb():
        enc(X).backward(G)
    t0, t1 = timeit(f), timeit(fb)
    print(f"VLFAN(use_feat_proj, trainable) {str(dt)[6:]:8s} N={n:6d}: forward {t0:7.1f} us, forward + backward {t1:7.1f} us = x{t1 / t0:4.2f}")
