"""Randomised cross-check of the per-bag training node (VF.slide_train through VLSA.forward) against the general autograd route:
random N, P, K, dtype, gated query, identity / Linear adapter, row stride; logits and every gradient.  python tools/fuzz_slide_train.py [n]"""
import os, sys, random
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import numpy as np
import torch
import cases, helpers as H
from test_gpu_modules import build_vlsa
from vlsa_amd.vlsa import VLSA

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
rng = random.Random(1234)
worst = 0.0
orig = VLSA._slide_train
for it in range(n_cases):
    N = rng.choice([1, 2, 3, 15, 16, 17, 31, 32, 33, 63, 64, 65, 127, 129, 255, 257, 1000, 2798, 4097, rng.randint(1, 9000), rng.randint(9000, 60000)])
    gated = rng.random() < 0.3
    P = rng.randint(1, 15 if gated else 16)
    K = rng.choice([1, 2, 4, 8, 12, 33, 64])
    head = rng.choice(["default", "Identity"])
    dtype = rng.choice([torch.float32, torch.bfloat16])
    case = ("fz", N, P, K, "mean", head, gated, "iid", 5000 + it, True)
    X, params, pool = H.vlfan_case_inputs(case)
    model, tp = build_vlsa(case, params, pool)
    model.train()
    Xd = X.to(dtype).cuda()
    if rng.random() < 0.3:                       # strided rows
        wide = torch.zeros(N, 768, dtype=dtype, device="cuda"); wide[:, :512] = Xd; Xd = wide[:, :512]
    G = torch.randn(1, K, generator=torch.Generator().manual_seed(it)).cuda()
    named = dict(model.named_parameters()); named["T"] = tp.T

    def run():
        out = model(Xd[None])
        (out[0] * G).sum().backward()
        g = {k: p.grad.detach().clone() for k, p in named.items() if p.grad is not None}
        model.zero_grad(set_to_none=True); tp.zero_grad(set_to_none=True)
        return out[0].detach(), type(out[0].grad_fn).__name__, g
    VLSA._slide_train = orig
    l1, fn1, g1 = run()
    VLSA._slide_train = lambda self, X, T: None
    l2, fn2, g2 = run()
    VLSA._slide_train = orig
    assert fn1 == "_SlideTrainFnBackward" and fn2 != fn1, (fn1, fn2)
    dl = float((l1 - l2).abs().max())
    assert dl < 1e-4, (it, case, dtype, dl)
    assert set(g1) == set(g2)
    for k in g2:
        a, b = g1[k].float(), g2[k].float()
        err = float((a - b).abs().max()); ref = float(b.abs().max())
        rel = err / (ref + 1e-12)
        worst = max(worst, rel if ref > 1e-6 else 0.0)
        assert err <= 2e-3 * ref + 2e-6, (it, case, dtype, k, err, ref)
print(f"{n_cases} random cases: per-bag node == general route; worst relative gradient difference {worst:.2e}")
