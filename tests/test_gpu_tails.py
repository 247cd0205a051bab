"""Round-5 tails against a float64 torch tail computed from the single-bag aggregation kernels' rows (model/deepmil.py:133-150,203-204,
model/vlsa.py:188-192): the batched plan's merge + pooling (one workgroup per bag for <= 8 partial records, the column-chunk kernel
above), the f32-MFMA adapter for >= 16 bags (and the VALU one below), the finish kernel; the single-slide plan's merge + W-partials +
finish (mean / weight pooling with a Linear adapter) and its fallback (max pooling, identity adapter); the cached query / text
preparation (`params_key`).  `tools/fuzz_tails.py` is the randomised version."""
import pytest
import torch

import cases

pytestmark = pytest.mark.gpu


def _tail64(rows, mode, pw, W, b, T, ls):
    rows = rows.double()
    if mode == "mean":
        pooled = rows.mean(dim=-2)
    elif mode == "max":
        pooled = rows.max(dim=-2).values
    else:
        pooled = (torch.softmax(pw.double(), 0)[:, None] * rows).sum(dim=-2)
    v = pooled if W is None else pooled @ W.double().t() + b.double()
    vh = v / v.norm(dim=-1, keepdim=True).clamp_min(1e-12)
    Th = T.double() / T.double().norm(dim=-1, keepdim=True).clamp_min(1e-12)
    logits = ls.double().exp() * vh @ Th.t()
    return logits, vh, torch.softmax(logits, dim=-1)


CASES = [  # B, P, K, pooling, identity adapter, gated, dtype, largest bag
    (1, 12, 4, "mean", False, False, torch.bfloat16, 50_000),
    (1, 7, 12, "weight", False, True, torch.float32, 2798),
    (1, 13, 5, "max", False, False, torch.bfloat16, 4100),
    (1, 12, 4, "mean", True, False, torch.bfloat16, 900),
    (15, 12, 4, "mean", False, False, torch.bfloat16, 30_000),       # VALU adapter (< 16 bags), 8-32 partial records per bag
    (16, 16, 64, "mean", False, False, torch.float32, 4100),          # MFMA adapter, one full bag tile
    (17, 1, 1, "weight", False, False, torch.bfloat16, 2798),         # ... a ragged second tile
    (100, 12, 8, "max", False, True, torch.bfloat16, 1200),
    (256, 12, 4, "mean", False, False, torch.bfloat16, 700),          # one workgroup and one partial record per bag
    (200, 4, 33, "weight", True, False, torch.float32, 300),
]


@pytest.mark.parametrize("B,P,K,mode,ident,gated,dtype,nmax", CASES)
def test_tails_match_a_float64_tail_of_the_single_bag_rows(B, P, K, mode, ident, gated, dtype, nmax):
    from vlsa_amd import functional as F
    dev = torch.device("cuda", 0)
    g = cases.gen(5100 + B + P + K)
    sizes = [nmax if i == 0 else [1, 17, 64, 65, 333, 700, 1023][i % 7] for i in range(B)]
    base = cases.make_bag(max(sizes) + 4 * B, 5200 + B, "clustered").to(dtype).to(dev)
    bags = [base[4 * i:4 * i + n] for i, n in enumerate(sizes)]
    Q = torch.randn(P + (1 if gated else 0), 512, generator=g).to(dev)
    T = torch.randn(K, 512, generator=g).to(dev)
    W = None if ident else (torch.randn(512, 512, generator=g) / 22).to(dev)
    b = None if ident else (torch.randn(512, generator=g) * 0.1).to(dev)
    pw = torch.randn(P, generator=g).to(dev) if mode == "weight" else None
    ls = torch.tensor(cases.LOGIT_SCALE, device=dev)
    rows = torch.stack([F.vlfan_aggregate(x, Q, gated)[0] for x in bags])
    rl, rv, ri = _tail64(rows, mode, pw, W, b, T, ls)
    plan = F.VlfanBatchPlan(B, P, K, dev, gated=gated, pool=mode, identity_head=ident)
    plan.set_bags(bags)
    for key in (None, 7, 7):                               # a fresh preparation, then the cached block twice
        plan.run(Q, T, ls, W, b, pw, params_key=key)
    torch.cuda.synchronize()
    assert (plan.logits.double() - rl).abs().max().item() < 1e-4
    assert (plan.vhat.double() - rv).abs().max().item() < 1e-5
    assert (plan.incidence.double() - ri).abs().max().item() < 1e-4
    assert (plan.out.double() - rows.double()).abs().max().item() < 1e-4 * max(1.0, rows.abs().max().item())
    # a changed query under a NEW key is seen; under the SAME key it (by contract) is not
    Q2 = Q * 1.5 + 0.1
    l_new = plan.run(Q2, T, ls, W, b, pw, params_key=8).clone()
    rows2 = torch.stack([F.vlfan_aggregate(x, Q2, gated)[0] for x in bags])
    assert (l_new.double() - _tail64(rows2, mode, pw, W, b, T, ls)[0]).abs().max().item() < 1e-4
    # the single-slide plan on the largest bag
    sp = F.VlfanInferencePlan(sizes[0], 512, P, K, dev, gated=gated, pool=mode, identity_head=ident)
    sp.run(bags[0], Q, T, ls, W, b, pw)
    torch.cuda.synchronize()
    assert (sp.logits.double() - rl[0]).abs().max().item() < 1e-4
    assert (sp.vhat.double() - rv[0]).abs().max().item() < 1e-5
    assert (sp.incidence.double() - ri[0]).abs().max().item() < 1e-4
    assert (sp.out.double() - rows[0].double()).abs().max().item() < 1e-5 * max(1.0, rows.abs().max().item())


def test_merge_head_entry_point_equals_merge_plus_head():
    """vlsa_vlfan_merge_head (two ticket-free launches) against vlsa_vlfan_merge + vlsa_head_forward on the same partial records"""
    import ctypes
    from vlsa_amd import _native as nat
    from vlsa_amd import functional as F
    lib, dev, p = nat.load(), torch.device("cuda", 0), F._p
    g = torch.Generator(device=dev).manual_seed(3)
    for G, P, K, mode in ((256, 12, 4, 0), (44, 16, 64, 2), (1, 1, 1, 0), (7, 5, 3, 1)):
        pm = torch.randn(G, 16, device=dev, generator=g) * 20
        pl = torch.rand(G, 16, device=dev, generator=g) + 0.5
        pacc = torch.randn(G, P, 512, device=dev, generator=g)
        W = torch.randn(512, 512, device=dev, generator=g) / 22
        b = torch.randn(512, device=dev, generator=g)
        That = torch.nn.functional.normalize(torch.randn(K, 512, device=dev, generator=g), dim=-1)
        ls = torch.tensor([4.03], device=dev)
        pw = torch.randn(P, device=dev, generator=g)
        outs = []
        for fused in (True, False):
            ws = torch.zeros(lib.vlsa_head_workspace_bytes(512), dtype=torch.uint8, device=dev)
            f = lambda *s: torch.empty(*s, device=dev)  # noqa: E731
            m2, l, out, pooled, v, vhat, vnorm, logits, inc = f(16), f(16), f(P, 512), f(512), f(512), f(512), f(1), f(K), f(K)
            if fused:
                nat.check(lib.vlsa_vlfan_merge_head(p(pm), p(pl), p(pacc), G, P, 512, mode, p(pw), p(W), p(b), p(That), K, p(ls), p(ws), p(m2),
                                                    p(l), p(out), p(pooled), p(v), p(vhat), p(vnorm), p(logits), p(inc), F._stream()), "merge_head")
            else:
                nat.check(lib.vlsa_vlfan_merge(p(pm), p(pl), p(pacc), G, P, 512, 1, p(m2), p(l), p(out), F._stream()), "merge")
                nat.check(lib.vlsa_head_forward(p(out), P, 512, mode, p(pw), p(W), p(b), p(That), K, p(ls), p(ws), p(pooled), p(v), p(vhat),
                                                p(vnorm), p(logits), p(inc), F._stream()), "head")
            torch.cuda.synchronize()
            outs.append([t.clone() for t in (m2[:P], l[:P], out, pooled, v, vhat, vnorm, logits, inc)])
        for name, a_, b_ in zip(("m2", "l", "out", "pooled", "v", "vhat", "vnorm", "logits", "incidence"), *outs):
            tol = 1e-5 * max(1.0, float(b_.abs().max()))
            assert float((a_ - b_).abs().max()) <= tol, (G, P, K, mode, name, float((a_ - b_).abs().max()))


@pytest.mark.parametrize("B,P,K,mode", [(32, 12, 12, 0), (5, 7, 4, 1), (70, 16, 64, 2), (2, 1, 1, 0)])
def test_head_forward_with_the_text_rows_normalised_in_the_pooling_launch(B, P, K, mode):
    """vlsa_head_forward_batch_text (round 6: the training step's head in three launches) = vlsa_normalize_rows + vlsa_head_forward_batch,
    bit for bit, for every pooling mode."""
    import ctypes
    from vlsa_amd import _native as nat
    lib = nat.load()
    dev = torch.device("cuda", 0)
    g = torch.Generator().manual_seed(4100 + B)
    D = 512
    rows = torch.randn(B, P, D, generator=g).to(dev)
    T = (torch.randn(K, D, generator=g) * 3).to(dev)
    W = (torch.randn(D, D, generator=g) * D ** -0.5).to(dev)
    b = (torch.randn(D, generator=g) * 0.1).to(dev)
    pw = torch.randn(P, generator=g).to(dev)
    ls = torch.tensor([3.1], device=dev)
    p = lambda t: ctypes.c_void_p(t.data_ptr())  # noqa: E731
    s = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    f = lambda *sh: torch.empty(*sh, dtype=torch.float32, device=dev)  # noqa: E731
    outs = []
    for fused in (False, True):
        That, tnorm, pooled, v, vhat, vnorm, logits = f(K, D), f(K), f(B, D), f(B, D), f(B, D), f(B), f(B, K)
        if fused:
            nat.check(lib.vlsa_head_forward_batch_text(p(rows), B, P, D, mode, p(pw), p(W), p(b), p(T), K, p(ls), p(That), p(tnorm), p(pooled),
                                                       p(v), p(vhat), p(vnorm), p(logits), None, s), "text")
        else:
            tk = torch.zeros(B, dtype=torch.int32, device=dev)
            nat.check(lib.vlsa_normalize_rows(p(T), K, D, p(That), p(tnorm), s), "norm")
            nat.check(lib.vlsa_head_forward_batch(p(rows), B, P, D, mode, p(pw), p(W), p(b), p(That), K, p(ls), p(tk), p(pooled), p(v), p(vhat),
                                                  p(vnorm), p(logits), None, s), "head")
        outs.append((That, tnorm, pooled, v, vhat, vnorm, logits))
    for a, c, name in zip(outs[0], outs[1], ("That", "tnorm", "pooled", "v", "vhat", "vnorm", "logits")):
        if B == 1 and name in ("pooled",):
            continue
        assert torch.equal(a, c), name
