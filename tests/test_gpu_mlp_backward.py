"""HIP backward of the N-sized layers around the aggregation (vlsa_amd/csrc/mlp_backward.hip,
vlfan_dx.hip) against torch autograd through the CPU oracle's restatements of the reference modules:

  * (gated) attention scores (model/layers.py:103-122,137-153): dWa, dba, dWg, dbg, dw2, dc from the upstream dL/da
  * Feat_Projecter (model/layers.py:65-82): dW, db, dgamma, dbeta from the upstream dL/dY
  * cross attention (model/deepmil.py:187-200): dL/dX for fp32 bags (+ dQ from the existing kernels)

Gradient bar as everywhere else in the suite: 1e-4 of the tensor's largest entry.  The module-level paths through these
kernels are pinned by the reference-generated fixtures in test_gpu_modules.py / test_gpu_modules_r2.py."""
import numpy as np
import pytest
import torch

import cases
from oracle import vlsa_oracle as O

pytestmark = pytest.mark.gpu
RTOL = 1e-4       # BASELINE.md 3; observed <= 1.7e-5 for fp32 gradients (profiles/r04_grad_errors.txt)


def _close(got, ref, what, rtol=RTOL, atol=1e-6):
    got, ref = got.detach().float().cpu().numpy(), ref.detach().float().cpu().numpy()
    err = np.abs(got - ref).max()
    if rtol > 0:
        cases.record_grad_error(what, err, np.abs(ref).max(), rtol * np.abs(ref).max() + atol)
    assert err <= rtol * np.abs(ref).max() + atol, f"{what}: max abs err {err:.3e} vs max |ref| {np.abs(ref).max():.3e}"


@pytest.mark.parametrize("gated", [True, False])
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
@pytest.mark.parametrize("N", [1, 63, 64, 65, 300, 2798, 20000])
def test_attention_scores_backward_matches_autograd(gated, dtype, N):
    from vlsa_amd import functional as VF
    seed = 7100 + N
    X = cases.make_bag(N, seed, "clustered" if N % 2 else "iid", dtype=dtype)        # fp32 values (bf16-rounded when bf16)
    kind = "gated_attention" if gated else "attention"
    pp = {k: v.clone().requires_grad_(True) for k, v in cases.make_pool_params(kind, seed + 1).items()}
    G = torch.randn(N, generator=cases.gen(seed + 2))
    if gated:
        _, raw, _ = O.gated_attention_pooling(X, pp["wa"], pp["ba"], pp["wg"], pp["bg"], pp["w2"], pp["b2"])
    else:
        _, raw, _ = O.attention_pooling(X, pp["w1"], pp["b1"], pp["w2"], pp["b2"])
    (raw * G).sum().backward()
    dev = torch.device("cuda")
    Xd = X.to(dev).to(dtype)
    gp = {k: v.detach().to(dev).requires_grad_(True) for k, v in pp.items()}
    fused = VF.FusedAttnScores()
    if gated:
        a = VF.attn_scores_autograd(Xd, fused, gp["wa"], gp["ba"], gp["wg"], gp["bg"], gp["w2"], gp["b2"])
    else:
        a = VF.attn_scores_autograd(Xd, fused, gp["w1"], gp["b1"], None, None, gp["w2"], gp["b2"])
    _close(a, raw, "scores", rtol=0, atol=1e-4)
    (a * G.to(dev)).sum().backward()
    for k in pp:
        _close(gp[k].grad, pp[k].grad, f"d{k}")


def test_attention_scores_backward_through_the_deepmil_module():
    """DeepMIL(pooling='gated_attention') in eval mode under autograd: scores forward + backward in HIP, no [N, 256] GEMM."""
    from vlsa_amd.deepmil import DeepMIL
    N, seed = 5000, 7300
    X = cases.make_bag(N, seed, "clustered")
    enc = DeepMIL(dim_in=512, dim_hid=256, num_cls=512, use_feat_proj=False, pooling="gated_attention", pred_head="Adapter").cuda().eval()
    ref = {k: v.detach().cpu().clone().requires_grad_(True) for k, v in enc.state_dict().items()}
    out = enc(X[None].cuda())
    assert hasattr(enc, "_fused_scores")
    G = torch.randn(1, 512, generator=cases.gen(seed + 1))
    (out * G.cuda()).sum().backward()
    pool, _, _ = O.gated_attention_pooling(X, ref["sigma.fc1.0.weight"], ref["sigma.fc1.0.bias"], ref["sigma.score.0.weight"],
                                           ref["sigma.score.0.bias"], ref["sigma.fc2.weight"], ref["sigma.fc2.bias"])
    f = pool[None]
    r = 0.8 * f + 0.2 * O.adapter_forward(f, ref["visual_adapter.fc.0.weight"], ref["visual_adapter.fc.2.weight"])
    _close(out, r, "DeepMIL out", rtol=0, atol=1e-4)
    (r * G).sum().backward()
    got = dict(enc.named_parameters())
    for k in ("sigma.fc1.0.weight", "sigma.fc1.0.bias", "sigma.score.0.weight", "sigma.score.0.bias", "sigma.fc2.weight",
              "visual_adapter.fc.0.weight", "visual_adapter.fc.2.weight"):
        _close(got[k].grad, ref[k].grad, k)
    # fc2.bias shifts every score alike: softmax-invariant, its gradient is 0 up to rounding
    assert float(got["sigma.fc2.bias"].grad.abs().max()) < 1e-4


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
@pytest.mark.parametrize("N", [1, 31, 64, 65, 1000, 20000])
def test_feat_projecter_backward_matches_autograd(dtype, N):
    from vlsa_amd import functional as VF
    seed = 7500 + N
    X = cases.make_bag(N, seed, "iid", dtype=dtype)
    fp = {k: v.clone().requires_grad_(True) for k, v in cases.make_featproj_params(seed + 1).items()}
    G = torch.randn(N, 512, generator=cases.gen(seed + 2))
    Y = O.feat_projecter_forward(X, fp["w"], fp["b"], fp["gamma"], fp["beta"])
    (Y * G).sum().backward()
    dev = torch.device("cuda")
    gp = {k: v.detach().to(dev).requires_grad_(True) for k, v in fp.items()}
    fused = VF.FusedFeatProjecter()
    Yd = fused.autograd(X.to(dev).to(dtype), gp["w"], gp["b"], gp["gamma"], gp["beta"], 1e-5)
    _close(Yd, Y, "Y", rtol=0, atol=1e-4)
    (Yd * G.to(dev)).sum().backward()
    for k in fp:
        _close(gp[k].grad, fp[k].grad, f"d{k}")


@pytest.mark.parametrize("N,P,gated", [(1, 4, False), (17, 12, False), (64, 16, False), (300, 7, True), (2798, 12, False), (20000, 12, True)])
def test_cross_attention_dx_matches_autograd(N, P, gated):
    from vlsa_amd import functional as VF
    seed = 7700 + N
    X = cases.make_bag(N, seed, "clustered" if N > 100 else "iid").requires_grad_(True)
    params = cases.make_params(P, 4, seed + 1, gated)
    Q = (0.5 * params["resid"] + params["prompt"]).requires_grad_(True)
    G = torch.randn(P, 512, generator=cases.gen(seed + 2))
    A = torch.softmax(O.vlfan_attention_logits(X, Q, gated), dim=-1)
    out = A @ X
    (out * G).sum().backward()
    dev = torch.device("cuda")
    Xd = X.detach().to(dev).requires_grad_(True)
    Qd = Q.detach().to(dev).requires_grad_(True)
    od, _ = VF.vlfan_cross_attention(Xd, Qd, gated=gated)
    _close(od, out, "out", rtol=0, atol=1e-4 * max(1.0, float(out.detach().abs().max())))
    (od * G.to(dev)).sum().backward()
    _close(Qd.grad, Q.grad, "dQ", atol=2e-5)        # (N = 1: softmax over one patch, dQ is 0 up to rounding)
    # dX: rows with (near-)zero attention get gradients many orders of magnitude below the attended rows'; compare on the scale
    # of the whole tensor and, row by row, on each row's own scale
    _close(Xd.grad, X.grad, "dX")
    g, r = Xd.grad.cpu().numpy(), X.grad.numpy()
    row_scale = np.abs(r).max(axis=1)
    big = row_scale > 1e-3 * row_scale.max()
    assert (np.abs(g - r).max(axis=1)[big] <= 5e-3 * row_scale[big]).all()


def test_cross_attention_dx_for_a_batch_of_bags():
    from vlsa_amd import functional as VF
    sizes, P, seed = (700, 64, 3000), 12, 7900
    bags = [cases.make_bag(n, seed + i, "iid").requires_grad_(True) for i, n in enumerate(sizes)]
    params = cases.make_params(P, 4, seed + 10)
    Q = (0.5 * params["resid"] + params["prompt"]).requires_grad_(True)
    G = torch.randn(len(sizes), P, 512, generator=cases.gen(seed + 11))
    outs = torch.stack([torch.softmax(O.vlfan_attention_logits(x, Q), dim=-1) @ x for x in bags])
    (outs * G).sum().backward()
    dev = torch.device("cuda")
    bd = [x.detach().to(dev).requires_grad_(i != 1) for i, x in enumerate(bags)]     # the middle bag carries no gradient
    Qd = Q.detach().to(dev).requires_grad_(True)
    od = VF.vlfan_cross_attention_bags(bd, Qd)
    (od * G.to(dev)).sum().backward()
    _close(Qd.grad, Q.grad, "dQ")
    for i, x in enumerate(bags):
        if i == 1:
            assert bd[i].grad is None
        else:
            _close(bd[i].grad, x.grad, f"dX[{i}]")


def _dropout_keep(seed, rows, units, p):
    """The kernels' counter-based mask (vlsa_common.h: dropout_bits) re-stated with torch integer ops: [rows, units] bool."""
    M = 0xFFFFFFFF
    r = torch.arange(rows, dtype=torch.int64)[:, None]
    u = torch.as_tensor(units, dtype=torch.int64)[None, :]
    h = (seed ^ ((r * 0x9E3779B1) & M) ^ ((u * 0x85EBCA6B) & M)) & M
    h = h ^ (h >> 16)
    h = (h * 0x85EBCA6B) & M
    h = h ^ (h >> 13)
    h = (h * 0xC2B2AE35) & M
    h = h ^ (h >> 16)
    thr = max(1, int(p * 4294967296.0))
    return h >= thr


@pytest.mark.parametrize("N", [3000, 19001])       # (19 001 bf16 rows: the persistent LDS-DMA score kernel, gated_scores_tile.hip)
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
def test_gated_scores_training_dropout_forward_and_backward(dtype, N):
    """Gated_Attention_Pooling in train mode (nn.Dropout behind tanh and sigmoid, model/layers.py:94,99): the fused kernels with
    their own counter-based masks against torch autograd through the same arithmetic with the SAME masks."""
    from vlsa_amd import functional as VF
    p, seed = 0.25, 123457
    X = cases.make_bag(N, 8100, "clustered", dtype=dtype)
    pp = {k: v.clone().requires_grad_(True) for k, v in cases.make_pool_params("gated_attention", 8101).items()}
    G = torch.randn(N, generator=cases.gen(8102))
    ka = _dropout_keep(seed, N, range(256), p).float() / (1 - p)
    kg = _dropout_keep(seed, N, range(256, 512), p).float() / (1 - p)
    assert abs(float((ka > 0).float().mean()) - (1 - p)) < 0.01 and abs(float((kg > 0).float().mean()) - (1 - p)) < 0.01
    emb = torch.tanh(X @ pp["wa"].t() + pp["ba"]) * ka
    scr = torch.sigmoid(X @ pp["wg"].t() + pp["bg"]) * kg
    raw = ((emb * scr) @ pp["w2"].t() + pp["b2"]).squeeze(-1)
    (raw * G).sum().backward()
    dev = torch.device("cuda")
    gp = {k: v.detach().to(dev).requires_grad_(True) for k, v in pp.items()}
    a = VF.attn_scores_autograd(X.to(dev).to(dtype), VF.FusedAttnScores(), gp["wa"], gp["ba"], gp["wg"], gp["bg"], gp["w2"], gp["b2"],
                                drop_p=p, seed=seed)
    _close(a, raw, "scores under dropout", rtol=0, atol=2e-4)
    (a * G.to(dev)).sum().backward()
    for k in pp:
        _close(gp[k].grad, pp[k].grad, f"d{k}")


def test_deepmil_train_mode_runs_the_fused_kernels_with_dropout():
    """DeepMIL(pooling='gated_attention', drop_rate=0.25).train(): the reference's default training configuration of this encoder.
    Two forward passes draw different masks (torch's CPU generator seeds them), the mean over many draws approaches the eval-mode
    scores' pooled vector, and gradients flow to the pooling parameters through the HIP backward."""
    from vlsa_amd.deepmil import DeepMIL
    torch.manual_seed(11)
    enc = DeepMIL(dim_in=512, dim_hid=256, num_cls=512, use_feat_proj=False, drop_rate=0.25, pooling="gated_attention",
                  pred_head="Adapter").cuda().train()
    X = cases.make_bag(2000, 8200, "clustered").cuda()[None]
    torch.manual_seed(5)
    o1, o2 = enc(X), enc(X)
    assert hasattr(enc, "_fused_scores") and o1.requires_grad
    assert float((o1 - o2).detach().abs().max()) > 0                       # different dropout draws
    o1.sum().backward()
    assert enc.sigma.fc1[0].weight.grad is not None and float(enc.sigma.fc1[0].weight.grad.abs().max()) > 0
    torch.manual_seed(5)
    r1 = enc(X)
    assert float((r1 - o1).detach().abs().max()) == 0                      # torch.manual_seed governs the masks
    enc.eval()
    with torch.no_grad():
        ev = enc(X)
    enc.train()
    with torch.no_grad():
        mean = torch.stack([enc(X) for _ in range(48)]).mean(0)
    assert float((mean - ev).abs().max()) < 0.15 * float(ev.abs().max())


@pytest.mark.parametrize("gated", [True, False])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("N", [1, 33, 64, 200, 4000])
def test_attention_pooling_dx_matches_autograd(gated, dtype, N):
    """pooled = softmax(a(X)) @ X with gradients for the module's parameters AND the bag (trainable Feat_Projecter in front of a
    DeepMIL encoder): the fused node (vlsa_attn_scores_backward + vlsa_attn_scores_backward_dx) vs torch autograd."""
    from vlsa_amd import functional as VF
    seed = 8300 + N
    X = cases.make_bag(N, seed, "clustered" if N > 64 else "iid", dtype=dtype).requires_grad_(True)
    kind = "gated_attention" if gated else "attention"
    pp = {k: v.clone().requires_grad_(True) for k, v in cases.make_pool_params(kind, seed + 1).items()}
    G = torch.randn(512, generator=cases.gen(seed + 2))
    if gated:
        pooled, raw, _ = O.gated_attention_pooling(X, pp["wa"], pp["ba"], pp["wg"], pp["bg"], pp["w2"], pp["b2"])
    else:
        pooled, raw, _ = O.attention_pooling(X, pp["w1"], pp["b1"], pp["w2"], pp["b2"])
    (pooled * G).sum().backward()
    dev = torch.device("cuda")
    Xd = X.detach().to(dev).to(dtype).requires_grad_(True)
    gp = {k: v.detach().to(dev).requires_grad_(True) for k, v in pp.items()}
    w = (gp["wa"], gp["ba"], gp["wg"], gp["bg"], gp["w2"], gp["b2"]) if gated else (gp["w1"], gp["b1"], None, None, gp["w2"], gp["b2"])
    pd, ad = VF.attn_pool_autograd(Xd, VF.FusedAttnScores(), *w)
    _close(pd, pooled, "pooled", rtol=0, atol=1e-4)
    _close(ad, raw, "scores", rtol=0, atol=1e-4)
    (pd * G.to(dev)).sum().backward()
    for k in pp:
        if k != "b2":                       # softmax-invariant shift: gradient 0 up to rounding
            _close(gp[k].grad, pp[k].grad, f"d{k}", atol=2e-6)
    tol = RTOL if dtype == torch.float32 else 4.5e-3    # a bf16 bag receives a bf16 gradient: half an ulp = 2^-8 = 3.9e-3 (observed 3.4e-3)
    _close(Xd.grad, X.grad, "dX", rtol=tol, atol=1e-6)
