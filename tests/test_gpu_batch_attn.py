"""Attention weights out of the BATCHED path: the persistent streaming kernel stores its scores, one
more launch normalises them with the bag-global (m, l).  Checked against the CPU oracle, the reference-generated
fixtures (``fx["A"]``) and the single-bag path; bf16 and fp32 bags; ragged sizes; module API; gradients unaffected."""
import numpy as np
import pytest
import torch

import cases
import helpers as H
from oracle import vlsa_oracle as O

pytestmark = pytest.mark.gpu
TOL = 1e-4


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
@pytest.mark.parametrize("sizes", [[5000, 64, 2798, 33, 1, 4097], [300], [63, 65, 127, 129, 8192, 31, 32, 10_000, 17]])
@pytest.mark.parametrize("gated", [False, True])
def test_batch_plan_attention_weights_vs_oracle(sizes, dtype, gated):
    from vlsa_amd import functional as F
    dev = torch.device("cuda", 0)
    P, K = 12, 4
    bags = [cases.make_bag(n, 900 + i, "clustered" if i % 2 else "iid").to(dtype) for i, n in enumerate(sizes)]
    params = cases.make_params(P, K, 910, gated)
    Q = 0.5 * params["resid"] + params["prompt"]
    T, W, b = params["T"], params["W"], params["b"]
    ls = torch.tensor(cases.LOGIT_SCALE)
    plan = F.VlfanBatchPlan(len(sizes), P, K, dev, gated=gated, want_attn=True)
    plan.set_bags([x.to(dev) for x in bags])
    ref_plan = F.VlfanBatchPlan(len(sizes), P, K, dev, gated=gated)
    ref_plan.set_bags([x.to(dev) for x in bags])
    for _ in range(2):
        logits = plan.run(Q.to(dev), T.to(dev), ls.to(dev), W.to(dev), b.to(dev)).clone()
    ref_logits = ref_plan.run(Q.to(dev), T.to(dev), ls.to(dev), W.to(dev), b.to(dev))
    torch.cuda.synchronize()
    assert torch.equal(logits, ref_logits)                    # storing the scores does not change the aggregation
    for i, x in enumerate(bags):
        r = O.vlfan_forward(x.float(), Q, gated_query=gated)
        A = plan.attn.views[i].cpu()
        assert A.shape == r["A"].shape
        assert (A - r["A"]).abs().max().item() < TOL, (i, sizes[i])
        assert (A.sum(dim=1) - 1).abs().max().item() < 1e-4


@pytest.mark.parametrize("name", ["n2798_shipped", "n257_p12_wt", "n300_gatedq", "n4096_p13_id", "n64_adv", "n257_attn"])
def test_module_forward_bags_ret_with_attn_vs_reference_fixture(name):
    """VLSA.forward_bags(..., ret_with_attn=True) and VLFAN.forward_bags(..., ret_with_attn=True) vs fx['A'] (the
    reference's own VLFAN.forward(X, ret_with_attn=True)), no-grad (fused) and grad (autograd) routes."""
    from test_gpu_modules import build_vlsa
    case = [c for c in cases.VLFAN_CASES if c[0] == name][0]
    (_, N, P, K, pooling, head, gated, kind, seed, grads) = case
    fx = H.load_fixture("vlfan_" + name)
    X, params, pool = H.vlfan_case_inputs(case)
    model, tp = build_vlsa(case, params, pool)
    other = cases.make_bag(777, seed + 9)                       # a second, different bag in the same launch
    bags = [X[None].cuda(), other[None].cuda(), X.cuda()]
    with torch.no_grad():
        logits, feats, That, attn = model.forward_bags(bags, ret_with_attn=True)
    assert len(attn) == 3
    for idx in (0, 2):
        a = attn[idx][0] if isinstance(attn[idx], tuple) else attn[idx]
        assert tuple(a.shape) == (1, P, N)
        assert np.abs(a[0].cpu().numpy() - fx["A"]).max() < TOL
        assert np.abs(logits[idx].cpu().numpy() - fx["logits"][0]).max() < TOL
        if isinstance(attn[idx], tuple):
            assert np.abs(attn[idx][1].cpu().numpy().ravel() - fx["pool_ext"].ravel()).max() < 1e-4
    ro = O.vlfan_forward(other, (0.5 * params["resid"] + params["prompt"]), gated_query=gated)
    a1 = attn[1][0] if isinstance(attn[1], tuple) else attn[1]
    assert (a1[0].cpu() - ro["A"]).abs().max().item() < TOL
    # grad route: same weights, and the gradients still flow (A itself is detached, as in the reference's use)
    logits2, _, _, attn2 = model.forward_bags(bags, ret_with_attn=True)
    a = attn2[0][0] if isinstance(attn2[0], tuple) else attn2[0]
    assert not a.requires_grad and np.abs(a[0].cpu().numpy() - fx["A"]).max() < TOL
    if grads:
        (logits2[0:1] * H.t(fx["G"]).cuda()).sum().backward()
        g = model.mil_encoder.Q.grad if gated else model.mil_encoder.Q.residual_features.grad
        cases.check_big(fx, "grad.Q" if gated else "grad.resid", g, atol=1e-5, rtol=1e-4)
    v, attn3 = model.mil_encoder.forward_bags(bags[:2], ret_with_attn=True)
    assert v.shape[0] == 2 and len(attn3) == 2


# ---- the other encoders through forward_bags: same numbers as bag-by-bag forward and as the fixtures ----
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("pooling", ["logit_top10", "logit_max", "logit_mean", "logit_top3"])
def test_forward_bags_zeroshot(pooling, dtype):
    from vlsa_amd.vlsa import VLSA
    K = 12
    T = cases.make_params(1, K, 9900)["T"]
    model = VLSA.from_modules(dict(name="FeatMIL", pooling=pooling), pretrained_text_features=T, logit_scale_init=cases.LOGIT_SCALE).cuda().eval()
    sizes = [500, 5, 9000, 64, 1, 4097]
    bags = [cases.make_bag(n, 9910 + i).to(dtype) for i, n in enumerate(sizes)]
    with torch.no_grad():
        logits, feats, That = model.forward_bags([x.cuda() for x in bags])
        assert feats is None
        model.return_patch_features = True
        _, feats2, _ = model.forward_bags([x.cuda() for x in bags[:2]])
    assert logits.shape == (len(sizes), K) and len(feats2) == 2 and feats2[0].shape == (500, 512)
    for i, x in enumerate(bags):
        ref = O.vlsa_zeroshot_forward(x.float(), T, torch.tensor(cases.LOGIT_SCALE), pooling)[0]
        assert (logits[i].cpu() - ref[0]).abs().max().item() < TOL, (i, sizes[i])


@pytest.mark.parametrize("name", ["zs_top10", "zs_top10_n5", "zs_top3_k12", "zs_mean", "zs_max"])
def test_forward_bags_zeroshot_vs_reference_fixture(name):
    from vlsa_amd.vlsa import VLSA
    case = [c for c in cases.ZEROSHOT_CASES if c[0] == name][0]
    (_, N, K, pooling, seed) = case
    fx = H.load_fixture("zeroshot_" + name)
    X = cases.make_bag(N, seed)
    T = cases.make_params(1, K, seed + 1000)["T"]
    model = VLSA.from_modules(dict(name="FeatMIL", pooling=pooling), pretrained_text_features=T, logit_scale_init=cases.LOGIT_SCALE).cuda().eval()
    with torch.no_grad():
        logits, _, txt = model.forward_bags([X.cuda(), cases.make_bag(333, seed + 5).cuda(), X[None].cuda()])
    assert np.abs(logits[0].cpu().numpy() - fx["logits"][0]).max() < TOL
    assert np.abs(logits[2].cpu().numpy() - fx["logits"][0]).max() < TOL
    assert np.abs(txt.cpu().numpy() - fx["text_features"]).max() < 1e-6


@pytest.mark.parametrize("name", ["dm_attn", "dm_gattn", "dm_mean", "dm_max"])
def test_forward_bags_deepmil_vs_reference_fixture(name):
    from test_gpu_modules_r2 import _load_pool
    from vlsa_amd.vlsa import VLSA
    case = [c for c in cases.DEEPMIL_CASES if c[0] == name][0]
    (_, N, K, pooling, seed) = case
    fx = H.load_fixture("deepmil_" + name)
    X = cases.make_bag(N, seed)
    T = cases.make_params(1, K, seed + 1000)["T"]
    cfg = dict(name="DeepMIL", dim_in=512, dim_hid=256, num_cls=512, use_feat_proj=False, drop_rate=0.25, pooling=pooling,
               pred_head="Adapter", dim_reduction=4, keep_ratio=0.8)
    model = VLSA.from_modules(cfg, pretrained_text_features=T, logit_scale_init=cases.LOGIT_SCALE)
    ad = cases.make_adapter_params(seed + 4000)
    with torch.no_grad():
        model.mil_encoder.visual_adapter.fc[0].weight.copy_(ad["down"]); model.mil_encoder.visual_adapter.fc[2].weight.copy_(ad["up"])
    _load_pool(model.mil_encoder.sigma, cases.make_pool_params(pooling, seed + 3000), pooling)
    model = model.cuda().eval()
    with torch.no_grad():
        logits, feats, _ = model.forward_bags([X.cuda(), cases.make_bag(200, seed + 7).cuda(), X.to(torch.bfloat16).cuda().float()])
        single = model(X[None].cuda())[0]
    assert np.abs(logits[0].cpu().numpy() - fx["logits"][0]).max() < TOL
    assert np.abs(feats[0].cpu().numpy() - fx["image_features"][0]).max() < 1e-5
    assert (logits[0] - single[0]).abs().max().item() < 2e-5
