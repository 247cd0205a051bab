"""GPU parity of the VLFAN forward path through the C ABI vs the golden vectors and the CPU oracle.

Tolerance: north_star asks for attention weights and incidence logits within 1e-4 (fp32).
"""
import numpy as np
import pytest
import torch

import cases
import helpers as H
from oracle import vlsa_oracle as O

pytestmark = pytest.mark.gpu

TOL = 1e-4
KERNELS = {"generic": 1, "mfma": 2}


def _run_case(case, kernel, dtype=torch.float32):
    from vlsa_amd import functional as F
    (name, N, P, K, pooling, head, gated, kind, seed, grads) = case
    X, params, pool = H.vlfan_case_inputs(case)
    Q = 0.5 * params["resid"] + params["prompt"]
    dev = "cuda"
    Xd = X.to(dev).to(dtype)
    out, A, (m2, l, qp) = F.vlfan_aggregate(Xd, Q.to(dev), gated=gated, kernel=kernel, want_attn=True)
    res = dict(out=out, A=A)
    if pooling in ("mean", "max", "weight"):
        That, _ = F.normalize_rows(params["T"].to(dev))
        pw = pool["weight"].to(dev) if pooling == "weight" else None
        W = params["W"].to(dev) if head != "Identity" else None
        b = params["b"].to(dev) if head != "Identity" else None
        ls = torch.tensor(cases.LOGIT_SCALE, device=dev)
        res.update(F.head_forward(out, pooling, pw, W, b, That, ls, want_incidence=True))
        res["That"] = That
    torch.cuda.synchronize()
    return {k: (v.float().cpu() if isinstance(v, torch.Tensor) else v) for k, v in res.items()}


@pytest.mark.parametrize("kernel", ["generic", "mfma"])
@pytest.mark.parametrize("case", cases.VLFAN_CASES, ids=[c[0] for c in cases.VLFAN_CASES])
def test_vlfan_forward_vs_golden(case, kernel):
    fx = H.load_fixture("vlfan_" + case[0])
    pooling = case[4]
    r = _run_case(case, KERNELS[kernel])
    ref, _ = H.oracle_vlfan_case(case)
    # aggregated rows and attention weights (every pooling variant)
    scale = max(1.0, ref["out"].abs().max().item())
    assert (r["out"] - ref["out"]).abs().max().item() < TOL * scale
    assert np.abs(r["A"].numpy() - fx["A"]).max() < TOL
    assert np.abs(r["A"].sum(dim=1).numpy() - 1).max() < 1e-4
    if pooling in ("mean", "max", "weight"):
        assert np.abs(r["logits"].numpy() - fx["logits"].ravel()).max() < TOL
        assert np.abs(r["vhat"].numpy() - fx["image_features"].ravel()).max() < 1e-5
        assert np.abs(r["That"].numpy() - fx["text_features"]).max() < 1e-6
        inc = torch.softmax(torch.from_numpy(fx["logits"]).ravel(), dim=-1).numpy()
        assert np.abs(r["incidence"].numpy() - inc).max() < TOL


@pytest.mark.parametrize("case", [c for c in cases.VLFAN_CASES if c[1] >= 16], ids=[c[0] for c in cases.VLFAN_CASES if c[1] >= 16])
def test_vlfan_forward_bf16_input(case):
    """bf16 storage: the oracle is the fp32 math evaluated on the bf16-rounded bag."""
    (name, N, P, K, pooling, head, gated, kind, seed, grads) = case
    from vlsa_amd import functional as F
    X, params, pool = H.vlfan_case_inputs(case)
    Xb = X.to(torch.bfloat16)
    Q = 0.5 * params["resid"] + params["prompt"]
    ref = O.vlfan_forward(Xb.float(), Q, gated_query=gated)
    for kernel in (1, 2):
        out, A, _ = F.vlfan_aggregate(Xb.cuda(), Q.cuda(), gated=gated, kernel=kernel, want_attn=True)
        scale = max(1.0, ref["out"].abs().max().item())
        assert (out.cpu() - ref["out"]).abs().max().item() < TOL * scale
        assert (A.cpu() - ref["A"]).abs().max().item() < TOL


@pytest.mark.parametrize("N,dtype", [(10_000, torch.float32), (50_000, torch.bfloat16), (25_001, torch.bfloat16)])
def test_full_size_properties(N, dtype):
    """BASELINE sizes: (i) generic and MFMA kernels agree, (ii) shard-merge invariance, (iii) rows of A sum
    to 1, (iv) permutation invariance of the aggregated rows."""
    from vlsa_amd import functional as F
    P = 12
    X = cases.make_bag(N, 900 + N % 7).to(dtype).cuda()
    params = cases.make_params(P, 4, 901)
    Q = (0.5 * params["resid"] + params["prompt"]).cuda()
    out_m, A_m, (m2, l, qp) = F.vlfan_aggregate(X, Q, kernel=2, want_attn=True)
    out_g, A_g, _ = F.vlfan_aggregate(X, Q, kernel=1, want_attn=True)
    scale = max(1.0, out_g.abs().max().item())
    assert (out_m - out_g).abs().max().item() < TOL * scale
    assert (A_m - A_g).abs().max().item() < TOL
    assert (A_m.sum(dim=1) - 1).abs().max().item() < 2e-4
    # shard invariance: 3 uneven shards merged == single pass
    cuts = [0, N // 5, N // 2 + 3, N]
    parts = [F.vlfan_partial(X[a:b], qp) for a, b in zip(cuts[:-1], cuts[1:])]
    pm = torch.cat([p[0] for p in parts]); pl = torch.cat([p[1] for p in parts]); pacc = torch.cat([p[2] for p in parts])
    _, _, out_s = F.vlfan_merge(pm, pl, pacc)
    assert (out_s - out_m).abs().max().item() < TOL * scale
    # permutation invariance
    perm = torch.randperm(N, device="cuda", generator=torch.Generator(device="cuda").manual_seed(1))
    out_p, _, _ = F.vlfan_aggregate(X[perm].contiguous(), Q, kernel=2)
    assert (out_p - out_m).abs().max().item() < TOL * scale
    # CPU oracle on the same values
    ref = O.vlfan_forward(X.float().cpu(), Q.cpu())
    assert (out_m.cpu() - ref["out"]).abs().max().item() < TOL * scale
    assert (A_m.cpu() - ref["A"]).abs().max().item() < TOL
