"""Edge cases of the C-ABI path: strided rows, non-contiguous inputs, extreme P / K, other feature dims (generic kernel),
huge-magnitude and zero rows, large N, error codes."""
import ctypes

import numpy as np
import pytest
import torch

import cases
from oracle import vlsa_oracle as O

pytestmark = pytest.mark.gpu
TOL = 1e-4


def _ref(X, Q, gated=False):
    return O.vlfan_forward(X.float().cpu(), Q.cpu(), gated_query=gated)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_strided_and_noncontiguous_bags(dtype):
    from vlsa_amd import functional as F
    N = 700
    base = cases.make_bag(N, 41, d=1024).to(dtype).cuda()          # rows of 1024, we use the first 512 columns
    Q = torch.randn(9, 512, generator=cases.gen(42)).cuda()
    Xs = base[:, :512]                                              # row stride 1024 elements: consumed in place
    assert Xs.stride(0) == 1024
    out, A, _ = F.vlfan_aggregate(Xs, Q, want_attn=True)
    ref = _ref(Xs, Q)
    assert (out.cpu() - ref["out"]).abs().max().item() < TOL * max(1, ref["out"].abs().max().item())
    assert (A.cpu() - ref["A"]).abs().max().item() < TOL
    Xt = base[:, 1:513]                                             # misaligned start: library must copy, not crash
    out2, _, _ = F.vlfan_aggregate(Xt, Q)
    ref2 = _ref(Xt, Q)
    assert (out2.cpu() - ref2["out"]).abs().max().item() < TOL * max(1, ref2["out"].abs().max().item())
    Xr = base[::2, :512]                                            # every other row
    out3, _, _ = F.vlfan_aggregate(Xr, Q)
    ref3 = _ref(Xr, Q)
    assert (out3.cpu() - ref3["out"]).abs().max().item() < TOL * max(1, ref3["out"].abs().max().item())


@pytest.mark.parametrize("P,K,gated", [(1, 1, False), (16, 64, False), (15, 7, True), (16, 3, False)])
def test_extreme_query_and_class_counts(P, K, gated):
    from vlsa_amd import functional as F
    N, D = 1234, 512
    X = cases.make_bag(N, 43).cuda()
    g = cases.gen(44)
    Q = torch.randn(P + (1 if gated else 0), D, generator=g)
    T = torch.randn(K, D, generator=g)
    W = torch.randn(D, D, generator=g) / D ** 0.5
    b = torch.randn(D, generator=g) / D ** 0.5
    ls = torch.tensor(2.5)
    plan = F.VlfanInferencePlan(N, D, P, K, X.device, gated=gated)
    logits = plan.run(X, Q.cuda(), T.cuda(), ls.cuda(), W.cuda(), b.cuda())
    ref = O.vlsa_vlfan_forward(X.cpu(), Q, T, ls, gated_query=gated, head_weight=W, head_bias=b)
    assert (logits.cpu() - ref["logits"][0]).abs().max().item() < TOL
    assert (plan.incidence.cpu() - ref["incidence"][0]).abs().max().item() < TOL
    with pytest.raises((ValueError, Exception)):
        F.prepare_queries(torch.randn(18, D).cuda())                # P > 16 is refused, not silently truncated


@pytest.mark.parametrize("D", [64, 256, 768, 1024])
def test_other_feature_dims_use_the_generic_kernel(D):
    from vlsa_amd import functional as F
    N, P = 333, 5
    X = cases.make_bag(N, 45, d=D).cuda()
    Q = torch.randn(P, D, generator=cases.gen(46)).cuda()
    out, A, _ = F.vlfan_aggregate(X, Q, want_attn=True)
    ref = _ref(X, Q)
    assert (out.cpu() - ref["out"]).abs().max().item() < TOL * max(1, ref["out"].abs().max().item())
    assert (A.cpu() - ref["A"]).abs().max().item() < TOL
    pooled = F.scored_pool(X, None)
    assert (pooled.cpu() - X.cpu().mean(0)).abs().max().item() < 1e-5
    assert (F.colmax(X).cpu() - X.cpu().max(0).values).abs().max().item() == 0.0


def test_zero_rows_huge_rows_and_near_saturated_scores():
    from vlsa_amd import functional as F
    N = 515
    X = cases.make_bag(N, 47, "adversarial")
    X[10] = X[10] * 1e6
    X[11] = X[11] * 1e-6
    X[20:24] = 0.0
    Q = torch.randn(12, 512, generator=cases.gen(48))
    Q[0] = X[7] * 3.0            # cosine = 1 with patch 7: score 100, every other weight of that query underflows
    Q[1] = -X[7]
    for dt in (torch.float32, torch.bfloat16):
        Xd = X.to(dt)
        ref = _ref(Xd, Q)
        for kernel in ((1, 2) if dt == torch.float32 else (1, 2, 3)):
            out, A, _ = F.vlfan_aggregate(Xd.cuda(), Q.cuda(), kernel=kernel, want_attn=True)
            rel = (out.cpu() - ref["out"]).abs() / ref["out"].abs().clamp_min(1.0)
            assert rel.max().item() < TOL, (dt, kernel)
            assert (A.cpu() - ref["A"]).abs().max().item() < TOL
            assert torch.isfinite(out).all()


def test_large_bag_one_million_patches():
    from vlsa_amd import functional as F
    N = 1_000_003
    g = torch.Generator(device="cuda").manual_seed(5)
    X = torch.randn(N, 512, device="cuda", generator=g).to(torch.bfloat16)
    Q = torch.randn(12, 512, device="cuda", generator=g)
    out3, A3, _ = F.vlfan_aggregate(X, Q, kernel=3, want_attn=True)
    out2, A2, _ = F.vlfan_aggregate(X, Q, kernel=2, want_attn=True)
    assert (out3 - out2).abs().max().item() < TOL * max(1.0, out2.abs().max().item())
    assert (A3 - A2).abs().max().item() < TOL
    assert (A3.sum(dim=1) - 1).abs().max().item() < 5e-4
    # chunk-merge invariance at this size: 7 uneven shards through the partial/merge ABI
    qp = F.prepare_queries(Q)
    cuts = [0, 1, 130_000, 130_016, 555_555, 900_000, 999_999, N]
    parts = [F.vlfan_partial(X[a:b], qp) for a, b in zip(cuts[:-1], cuts[1:])]
    _, _, outs = F.vlfan_merge(torch.cat([p[0] for p in parts]), torch.cat([p[1] for p in parts]), torch.cat([p[2] for p in parts]))
    assert (outs - out3).abs().max().item() < TOL * max(1.0, out3.abs().max().item())


def test_error_codes_not_crashes():
    from vlsa_amd import _native as nat, functional as F
    lib = nat.load()
    X = torch.randn(64, 512, device="cuda")
    qp = F.prepare_queries(torch.randn(4, 512, device="cuda"))
    pm = torch.empty(2, 16, device="cuda"); pl = torch.empty(2, 16, device="cuda"); pacc = torch.empty(2, 4, 512, device="cuda")
    s = F._stream()
    p = F._p
    # null output pointer, misaligned X, unsupported kernel / dim combination, bad dtype
    assert lib.vlsa_vlfan_partial(p(X), 0, 64, 512, 512, p(qp.buf), 4, 0, None, p(pl), p(pacc), None, s) == -1
    assert lib.vlsa_vlfan_partial(ctypes.c_void_p(X.data_ptr() + 4), 0, 63, 512, 512, p(qp.buf), 4, 0, p(pm), p(pl), p(pacc), None, s) == -1
    assert lib.vlsa_vlfan_partial(p(X), 0, 64, 512, 512, p(qp.buf), 4, 3, p(pm), p(pl), p(pacc), None, s) == -2   # DMA kernel needs bf16
    assert lib.vlsa_vlfan_partial(p(X), 7, 64, 512, 512, p(qp.buf), 4, 0, p(pm), p(pl), p(pacc), None, s) == -1
    assert lib.vlsa_vlfan_partial(p(X), 0, 64, 512, 512, p(qp.buf), 17, 0, p(pm), p(pl), p(pacc), None, s) == -1
    assert lib.vlsa_topk_mean(p(X), 1, 64, 40, 1.0, p(pm), s) == -2                                               # k > 32 and k < N
    assert lib.vlsa_error_string(-2) == b"unsupported configuration"
    torch.cuda.synchronize()


def test_examples_synthetic_demo_runs():
    """examples/synthetic_demo.py: ingest -> batched evaluation -> training steps -> interpretation, as a user would."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "examples", "synthetic_demo.py"), "--patients", "40", "--steps", "2"],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "demo ok" in r.stdout
