"""Round-2 GPU parity tests of the drop-in modules against reference-generated fixtures (tests/golden/make_golden_r2.py):
``use_feat_proj=True`` (forward AND gradients reaching the projecter -- ADVICE r1 high), DeepMIL ``pred_head='default'``,
every PromptAdapter method incl. the negative prompt feeding a gated-query VLFAN, ``query_div_loss`` on the device, and
the fused MFMA attention scores of DeepMIL on bf16 bags."""
import numpy as np
import pytest
import torch
import torch.nn as nn

import cases
import helpers as H
from test_oracle_golden_r2 import prompt_adapter_state

pytestmark = pytest.mark.gpu
TOL = 1e-4
GRAD_RTOL, GRAD_ATOL = 1e-4, 1e-5      # as in test_gpu_modules.py (observed <= 2.6e-5: profiles/r04_grad_errors.txt)


class TextParam(nn.Module):
    def __init__(self, T):
        super().__init__()
        self.T = nn.Parameter(T.clone())


def _load_pool(sg, pp, pooling):
    with torch.no_grad():
        if pooling == "attention":
            sg.attention[0].weight.copy_(pp["w1"]); sg.attention[0].bias.copy_(pp["b1"])
            sg.attention[2].weight.copy_(pp["w2"]); sg.attention[2].bias.copy_(pp["b2"])
        elif pooling == "gated_attention":
            sg.fc1[0].weight.copy_(pp["wa"]); sg.fc1[0].bias.copy_(pp["ba"])
            sg.score[0].weight.copy_(pp["wg"]); sg.score[0].bias.copy_(pp["bg"])
            sg.fc2.weight.copy_(pp["w2"]); sg.fc2.bias.copy_(pp["b2"])


def _pool_grads(sg):
    if type(sg).__name__ == "Attention_Pooling":
        return {"w1": sg.attention[0].weight.grad, "b1": sg.attention[0].bias.grad,
                "w2": sg.attention[2].weight.grad, "b2": sg.attention[2].bias.grad}
    if type(sg).__name__ == "Gated_Attention_Pooling":
        return {"wa": sg.fc1[0].weight.grad, "ba": sg.fc1[0].bias.grad, "wg": sg.score[0].weight.grad,
                "bg": sg.score[0].bias.grad, "w2": sg.fc2.weight.grad, "b2": sg.fc2.bias.grad}
    return {}


@pytest.mark.parametrize("case", cases.FEATPROJ_CASES, ids=[c[0] for c in cases.FEATPROJ_CASES])
def test_feat_projecter_forward_and_gradients(case):
    from vlsa_amd.vlsa import VLSA
    (name, enc_name, N, P, K, pooling, seed) = case
    fx = H.load_fixture("featproj_" + name)
    X = cases.make_bag(N, seed)
    params = cases.make_params(max(P, 1), K, seed + 1000)
    tp = TextParam(params["T"])
    if enc_name == "VLFAN":
        cfg = dict(name="VLFAN", dim_in=512, dim_hid=256, use_feat_proj=True, drop_rate=0.25, num_query=P,
                   query="Parameter", gated_query=False, query_pooling=pooling, pred_head="default")
    else:
        cfg = dict(name="DeepMIL", dim_in=512, dim_hid=256, num_cls=512, use_feat_proj=True, drop_rate=0.25,
                   pooling=pooling, pred_head="Adapter", dim_reduction=4, keep_ratio=0.8)
    model = VLSA.from_modules(cfg, text_provider=lambda: tp.T, prompt_learner=tp, logit_scale_init=cases.LOGIT_SCALE)
    enc = model.mil_encoder
    fpp = cases.make_featproj_params(seed + 5000)
    with torch.no_grad():
        enc.feat_proj.projecter[0].weight.copy_(fpp["w"]); enc.feat_proj.projecter[0].bias.copy_(fpp["b"])
        enc.feat_proj.projecter[1].weight.copy_(fpp["gamma"]); enc.feat_proj.projecter[1].bias.copy_(fpp["beta"])
        if enc_name == "VLFAN":
            enc.Q.copy_(0.5 * params["resid"] + params["prompt"])
            enc.visual_adapter.weight.copy_(params["W"]); enc.visual_adapter.bias.copy_(params["b"])
        else:
            ad = cases.make_adapter_params(seed + 4000)
            enc.visual_adapter.fc[0].weight.copy_(ad["down"]); enc.visual_adapter.fc[2].weight.copy_(ad["up"])
    if enc_name != "VLFAN":
        _load_pool(enc.sigma, cases.make_pool_params(pooling, seed + 3000), pooling)
    model = model.cuda().eval()
    Xd = X[None].cuda()
    with torch.no_grad():                      # inference: projecter (vlsa_feat_project) and aggregation by the HIP kernels
        logits, img, _ = model(Xd)
        lb, _, _ = model.forward_bags([Xd, Xd])
    assert hasattr(enc.feat_proj, "_fused")    # the fused Feat_Projecter kernel ran
    if enc_name == "VLFAN" and isinstance(pooling, str) and pooling in ("mean", "max"):
        assert any(isinstance(k, tuple) and k and k[0] == "batch" for k in model._plans)   # ... in front of the batched fused path
    assert np.abs(logits.cpu().numpy() - fx["logits"]).max() < TOL
    assert np.abs(img.cpu().numpy() - fx["image_features"]).max() < 1e-5
    assert np.abs(lb.cpu().numpy() - fx["logits"]).max() < TOL
    logits2, _, _ = model(Xd)                  # training: the bag now carries a gradient (into the projecter)
    assert np.abs(logits2.detach().cpu().numpy() - fx["logits"]).max() < TOL
    (logits2 * H.t(fx["G"]).cuda()).sum().backward()
    chk = lambda key, g: cases.check_big(fx, key, g, atol=GRAD_ATOL, rtol=GRAD_RTOL)  # noqa: E731
    fp = enc.feat_proj.projecter
    chk("grad.fp.w", fp[0].weight.grad); chk("grad.fp.b", fp[0].bias.grad)
    chk("grad.fp.gamma", fp[1].weight.grad); chk("grad.fp.beta", fp[1].bias.grad)
    chk("grad.logit_scale", model.logit_scale.grad); chk("grad.T", tp.T.grad)
    if enc_name == "VLFAN":
        chk("grad.Q", enc.Q.grad); chk("grad.W", enc.visual_adapter.weight.grad); chk("grad.b", enc.visual_adapter.bias.grad)
    else:
        chk("grad.adapter.down", enc.visual_adapter.fc[0].weight.grad); chk("grad.adapter.up", enc.visual_adapter.fc[2].weight.grad)
        for k, g in _pool_grads(enc.sigma).items():
            chk("grad.pool." + k, g)


def test_functional_api_refuses_a_bag_that_requires_grad():
    from vlsa_amd import functional as F
    X = torch.randn(64, 512, device="cuda", requires_grad=True)
    Q = torch.randn(4, 512, device="cuda", requires_grad=True)
    # fp32 [N, 512] bags get dX from vlsa_vlfan_backward_dx (round 3); anything else is still refused, never detached silently
    Xb = torch.randn(64, 512, device="cuda", dtype=torch.bfloat16, requires_grad=True)
    with pytest.raises(F.VlsaNativeError):
        F.vlfan_cross_attention(Xb, Q)
    with pytest.raises(F.VlsaNativeError):
        F.vlfan_cross_attention(torch.randn(64, 256, device="cuda", requires_grad=True), torch.randn(4, 256, device="cuda"))
    with pytest.raises(F.VlsaNativeError):
        F.scored_pool(X, torch.randn(64, device="cuda"))
    with torch.no_grad():
        F.vlfan_cross_attention(X, Q)          # fine when nothing is differentiated


@pytest.mark.parametrize("case", cases.DEEPMIL_HEAD_CASES, ids=[c[0] for c in cases.DEEPMIL_HEAD_CASES])
def test_deepmil_linear_head(case):
    from vlsa_amd.vlsa import VLSA
    (name, N, K, pooling, seed) = case
    fx = H.load_fixture("deepmil_" + name)
    X = cases.make_bag(N, seed)
    params = cases.make_params(1, K, seed + 1000)
    cfg = dict(name="DeepMIL", dim_in=512, dim_hid=256, num_cls=512, use_feat_proj=False, drop_rate=0.25,
               pooling=pooling, pred_head="default")
    model = VLSA.from_modules(cfg, pretrained_text_features=params["T"], logit_scale_init=cases.LOGIT_SCALE)
    enc = model.mil_encoder
    gp = cases.make_linear_params(seed + 4000, cases.D, cases.D)
    with torch.no_grad():
        enc.g.weight.copy_(gp["w"]); enc.g.bias.copy_(gp["b"])
    _load_pool(enc.sigma, cases.make_pool_params(pooling, seed + 3000), pooling)
    model = model.cuda().eval()
    Xd = X[None].cuda()
    with torch.no_grad():
        logits, img, _ = model(Xd)
    assert np.abs(logits.cpu().numpy() - fx["logits"]).max() < TOL
    assert np.abs(img.cpu().numpy() - fx["image_features"]).max() < 1e-5
    logits2, _, _ = model(Xd)
    (logits2 * H.t(fx["G"]).cuda()).sum().backward()
    chk = lambda key, g: cases.check_big(fx, key, g, atol=GRAD_ATOL, rtol=GRAD_RTOL)  # noqa: E731
    chk("grad.g.w", enc.g.weight.grad); chk("grad.g.b", enc.g.bias.grad)
    for k, g in _pool_grads(enc.sigma).items():
        chk("grad.pool." + k, g)


def build_prompt_adapter(case):
    from vlsa_amd.prompt_adapter import PromptAdapter
    (name, method, P, neg, seed) = case
    st = prompt_adapter_state(method, P, neg, seed)
    pa = PromptAdapter(method=method, num_prompts=P, pretrained_prompt_features=st["feats"], res_ratio=0.5, keep_ratio=0.8,
                       dim_reduction=4, load_negative_prompts=neg, pretrained_neg_prompt_features=st.get("neg"))
    with torch.no_grad():
        if method == "TaskRes":
            pa.residual_features.copy_(st["resid"])
            if neg:
                pa.neg_residual_features.copy_(st["neg_resid"])
        elif method == "Adapter":
            pa.adapter.fc[0].weight.copy_(st["adapter"]["down"]); pa.adapter.fc[2].weight.copy_(st["adapter"]["up"])
        elif method == "FC":
            pa.fc[0].weight.copy_(st["fc"])
    return pa.cuda().eval()


@pytest.mark.parametrize("case", cases.PROMPT_ADAPTER_CASES, ids=[c[0] for c in cases.PROMPT_ADAPTER_CASES])
def test_prompt_adapter_methods(case):
    (name, method, P, neg, seed) = case
    fx = H.load_fixture("padapter_" + name)
    pa = build_prompt_adapter(case)
    Q = pa()
    assert tuple(Q.shape) == fx["Q"].shape
    assert np.abs(Q.detach().cpu().numpy() - fx["Q"]).max() < 1e-5
    assert np.abs(pa.get_raw_prompt_features().cpu().numpy() - fx["raw"]).max() == 0
    if "G" in fx:
        (Q * H.t(fx["G"]).cuda()).sum().backward()
        for n, p in pa.named_parameters():
            cases.check_big(fx, "grad." + n, p.grad, atol=GRAD_ATOL, rtol=GRAD_RTOL)


def test_negative_prompt_adapter_feeds_gated_query_vlfan():
    """TaskRes + negative prompt as the query network of VLFAN(gated_query=True) (model/vlsa.py:80-94,
    model/deepmil.py:192-195): P + 1 queries, the last one subtracted; HIP path vs the CPU oracle."""
    from oracle import vlsa_oracle as O
    from vlsa_amd.vlsa import VLSA
    case = [c for c in cases.PROMPT_ADAPTER_CASES if c[0] == "pa_taskres_neg"][0]
    (name, method, P, neg, seed) = case
    pa = build_prompt_adapter(case)
    params = cases.make_params(P, 8, seed + 1000)
    cfg = dict(name="VLFAN", dim_in=512, dim_hid=256, use_feat_proj=False, num_query=P, query="Text", gated_query=True,
               query_pooling="mean", pred_head="default")
    model = VLSA.from_modules(cfg, pretrained_text_features=params["T"], query_network=pa, logit_scale_init=cases.LOGIT_SCALE)
    with torch.no_grad():
        model.mil_encoder.visual_adapter.weight.copy_(params["W"]); model.mil_encoder.visual_adapter.bias.copy_(params["b"])
    model = model.cuda().eval()
    X = cases.make_bag(700, seed + 7)
    ref = O.vlsa_vlfan_forward(X, pa().detach().cpu(), params["T"], torch.tensor(cases.LOGIT_SCALE), gated_query=True,
                               head_weight=params["W"], head_bias=params["b"])
    with torch.no_grad():
        logits, _, _ = model(X[None].cuda())
        v, A = model.mil_encoder(X[None].cuda(), ret_with_attn=True)
    assert (logits.cpu() - ref["logits"]).abs().max().item() < TOL
    assert (A[0].cpu() - ref["A"]).abs().max().item() < TOL
    logits2, _, _ = model(X[None].cuda())                 # gradients reach both residuals through the HIP backward
    logits2.sum().backward()
    assert pa.residual_features.grad.abs().max().item() > 0 and pa.neg_residual_features.grad.abs().max().item() > 0


def test_query_div_loss_on_device():
    from vlsa_amd.deepmil import VLFAN
    fx = H.load_fixture("query_div")
    for tag, gated in (("plain", False), ("gated", True)):
        enc = VLFAN(dim_in=512, use_feat_proj=False, query="Parameter", num_query=6, gated_query=gated).cuda()
        with torch.no_grad():
            enc.Q.copy_(torch.from_numpy(fx[f"{tag}.Q"]))
        l1, l2 = enc.query_div_loss(last_div=True), enc.query_div_loss(last_div=False)
        assert l1.is_cuda and abs(l1.item() - float(fx[f"{tag}.loss_last_div"])) < 1e-6
        assert abs(l2.item() - float(fx[f"{tag}.loss_all"])) < 1e-6
        l1.backward()
        assert enc.Q.grad is not None and torch.isfinite(enc.Q.grad).all()


@pytest.mark.parametrize("case", cases.DEEPMIL_BF16_CASES, ids=[c[0] for c in cases.DEEPMIL_BF16_CASES])
def test_deepmil_bf16_bag_fused_scores_vs_reference_fixture(case):
    from vlsa_amd.vlsa import VLSA
    (name, N, K, pooling, seed) = case
    fx = H.load_fixture("deepmil_" + name)
    X = cases.make_bag(N, seed, "iid", torch.bfloat16)
    params = cases.make_params(1, K, seed + 1000)
    cfg = dict(name="DeepMIL", dim_in=512, dim_hid=256, num_cls=512, use_feat_proj=False, drop_rate=0.25,
               pooling=pooling, pred_head="Adapter", dim_reduction=4, keep_ratio=0.8)
    model = VLSA.from_modules(cfg, pretrained_text_features=params["T"], logit_scale_init=cases.LOGIT_SCALE)
    enc = model.mil_encoder
    ad = cases.make_adapter_params(seed + 4000)
    with torch.no_grad():
        enc.visual_adapter.fc[0].weight.copy_(ad["down"]); enc.visual_adapter.fc[2].weight.copy_(ad["up"])
    _load_pool(enc.sigma, cases.make_pool_params(pooling, seed + 3000), pooling)
    model = model.cuda().eval()
    for dt in (torch.bfloat16, torch.float32):          # bf16 storage: fused MFMA kernel; same values as fp32: fp32 route
        Xd = X.to(dt)[None].cuda()
        with torch.no_grad():
            logits, img, _ = model(Xd)
            v, attn = enc(Xd, ret_with_attn=True)
        if dt == torch.bfloat16:
            assert hasattr(enc, "_fused_scores")
        assert np.abs(logits.cpu().numpy() - fx["logits"]).max() < TOL, dt
        assert np.abs(attn.cpu().numpy().ravel() - fx["attn"].ravel()).max() < TOL, dt
        assert np.abs(v.cpu().numpy().ravel() - fx["v"].ravel()).max() < TOL, dt
