"""Drop-in modules (VLSA / VLFAN / FeatMIL / DeepMIL) on the GPU vs the golden vectors produced by the reference:
forward through the fused inference path and through the autograd path, and backward gradients."""
import numpy as np
import pytest
import torch
import torch.nn as nn

import cases
import helpers as H

pytestmark = pytest.mark.gpu
TOL = 1e-4
GRAD_RTOL = 1e-4   # "backward grads within 1e-4 relative" (of the largest entry of each gradient tensor); the worst
# error observed over the 621 gradient comparisons of the GPU suite is 2.6e-5 (profiles/r04_grad_errors.txt)
GRAD_ATOL = 1e-5   # gradients that are exactly 0 in the reference (softmax-invariant biases, N = 1 bags): rounding noise <= 3.5e-6


class TextParam(nn.Module):
    def __init__(self, T):
        super().__init__()
        self.T = nn.Parameter(T.clone())


def build_vlsa(case, params, pool, requires_grad=True):
    from vlsa_amd.prompt_adapter import PromptAdapter
    from vlsa_amd.vlsa import VLSA
    (name, N, P, K, pooling, head, gated, kind, seed, grads) = case
    cfg = dict(name="VLFAN", dim_in=512, dim_hid=256, use_feat_proj=False, drop_rate=0.25, num_query=P,
               query="Parameter" if gated else "Text", gated_query=gated, query_pooling=pooling, pred_head=head)
    tp = TextParam(params["T"])
    qnet = None if gated else PromptAdapter(method="TaskRes", num_prompts=P, pretrained_prompt_features=params["prompt"],
                                            res_ratio=0.5)
    model = VLSA.from_modules(cfg, text_provider=lambda: tp.T, prompt_learner=tp, query_network=qnet, logit_scale_init=cases.LOGIT_SCALE)
    enc = model.mil_encoder
    with torch.no_grad():
        if gated:
            enc.Q.copy_(0.5 * params["resid"] + params["prompt"])
        else:
            enc.Q.residual_features.copy_(params["resid"])
        if head != "Identity":
            enc.visual_adapter.weight.copy_(params["W"])
            enc.visual_adapter.bias.copy_(params["b"])
        qp = enc.query_pooling
        if pooling == "weight":
            qp.copy_(pool["weight"])
        elif pooling == "attention":
            qp.attention[0].weight.copy_(pool["w1"]); qp.attention[0].bias.copy_(pool["b1"])
            qp.attention[2].weight.copy_(pool["w2"]); qp.attention[2].bias.copy_(pool["b2"])
        elif pooling == "gated_attention":
            qp.fc1[0].weight.copy_(pool["wa"]); qp.fc1[0].bias.copy_(pool["ba"])
            qp.score[0].weight.copy_(pool["wg"]); qp.score[0].bias.copy_(pool["bg"])
            qp.fc2.weight.copy_(pool["w2"]); qp.fc2.bias.copy_(pool["b2"])
    model = model.cuda().eval()
    return model, tp


def pool_grads(qp):
    if isinstance(qp, nn.Parameter):
        return {"weight": qp.grad}
    if type(qp).__name__ == "Attention_Pooling":
        return {"w1": qp.attention[0].weight.grad, "b1": qp.attention[0].bias.grad,
                "w2": qp.attention[2].weight.grad, "b2": qp.attention[2].bias.grad}
    if type(qp).__name__ == "Gated_Attention_Pooling":
        return {"wa": qp.fc1[0].weight.grad, "ba": qp.fc1[0].bias.grad, "wg": qp.score[0].weight.grad,
                "bg": qp.score[0].bias.grad, "w2": qp.fc2.weight.grad, "b2": qp.fc2.bias.grad}
    return {}


@pytest.mark.parametrize("case", cases.VLFAN_CASES, ids=[c[0] for c in cases.VLFAN_CASES])
def test_vlsa_vlfan_forward_backward(case):
    (name, N, P, K, pooling, head, gated, kind, seed, grads) = case
    fx = H.load_fixture("vlfan_" + name)
    X, params, pool = H.vlfan_case_inputs(case)
    model, tp = build_vlsa(case, params, pool)
    Xd = X[None].cuda()
    # (1) eval / no_grad: fused HIP path where the configuration allows it
    with torch.no_grad():
        logits, img, txt = model(Xd)
        v, attn = model.mil_encoder(Xd, ret_with_attn=True)
    assert np.abs(logits.cpu().numpy() - fx["logits"]).max() < TOL
    assert np.abs(img.cpu().numpy() - fx["image_features"]).max() < 1e-5
    assert np.abs(txt.cpu().numpy() - fx["text_features"]).max() < 1e-6
    A = attn[0] if isinstance(attn, tuple) else attn
    assert np.abs(A[0].cpu().numpy() - fx["A"]).max() < TOL
    assert np.abs(v.cpu().numpy() - fx["v"]).max() < TOL * max(1.0, np.abs(fx["v"]).max())
    if isinstance(attn, tuple):
        assert np.abs(attn[1].cpu().numpy().ravel() - fx["pool_ext"].ravel()).max() < 1e-4
    # (2) autograd path
    if not grads:
        return
    logits2, _, _ = model(Xd)
    assert np.abs(logits2.detach().cpu().numpy() - fx["logits"]).max() < TOL
    (logits2 * H.t(fx["G"]).cuda()).sum().backward()
    enc = model.mil_encoder
    chk = lambda key, g: cases.check_big(fx, key, g, atol=GRAD_ATOL, rtol=GRAD_RTOL)  # noqa: E731
    chk("grad.logit_scale", model.logit_scale.grad)
    chk("grad.T", tp.T.grad)
    if head != "Identity":
        chk("grad.W", enc.visual_adapter.weight.grad)
        chk("grad.b", enc.visual_adapter.bias.grad)
    if gated:
        chk("grad.Q", enc.Q.grad)
    else:
        chk("grad.resid", enc.Q.residual_features.grad)
    for k, g in pool_grads(enc.query_pooling).items():
        chk("grad.pool." + k, g)


@pytest.mark.parametrize("case", cases.ZEROSHOT_CASES, ids=[c[0] for c in cases.ZEROSHOT_CASES])
def test_vlsa_zeroshot_and_featmil(case):
    from vlsa_amd.vlsa import VLSA
    (name, N, K, pooling, seed) = case
    fx = H.load_fixture("zeroshot_" + name)
    X = cases.make_bag(N, seed)
    params = cases.make_params(1, K, seed + 1000)
    model = VLSA.from_modules(dict(name="FeatMIL", pooling=pooling), pretrained_text_features=params["T"],
                 logit_scale_init=cases.LOGIT_SCALE).cuda().eval()
    with torch.no_grad():
        logits, img, txt = model(X[None].cuda())
    assert np.abs(logits.cpu().numpy() - fx["logits"]).max() < TOL
    assert np.abs(txt.cpu().numpy() - fx["text_features"]).max() < 1e-6
    if "image_features" in fx:
        assert np.abs(img.cpu().numpy() - fx["image_features"]).max() < 1e-5
    else:
        assert np.abs(img[:8].cpu().numpy() - fx["image_features_rows"]).max() < 1e-6


@pytest.mark.parametrize("case", cases.DEEPMIL_CASES, ids=[c[0] for c in cases.DEEPMIL_CASES])
def test_vlsa_deepmil_forward_backward(case):
    from vlsa_amd.vlsa import VLSA
    (name, N, K, pooling, seed) = case
    fx = H.load_fixture("deepmil_" + name)
    X = cases.make_bag(N, seed)
    params = cases.make_params(1, K, seed + 1000)
    tp = TextParam(params["T"])
    cfg = dict(name="DeepMIL", dim_in=512, dim_hid=256, num_cls=512, use_feat_proj=False, drop_rate=0.25,
               pooling=pooling, pred_head="Adapter", dim_reduction=4, keep_ratio=0.8)
    model = VLSA.from_modules(cfg, text_provider=lambda: tp.T, prompt_learner=tp, logit_scale_init=cases.LOGIT_SCALE)
    enc = model.mil_encoder
    pp = cases.make_pool_params(pooling, seed + 3000)
    ad = cases.make_adapter_params(seed + 4000)
    with torch.no_grad():
        enc.visual_adapter.fc[0].weight.copy_(ad["down"]); enc.visual_adapter.fc[2].weight.copy_(ad["up"])
        sg = enc.sigma
        if pooling == "attention":
            sg.attention[0].weight.copy_(pp["w1"]); sg.attention[0].bias.copy_(pp["b1"])
            sg.attention[2].weight.copy_(pp["w2"]); sg.attention[2].bias.copy_(pp["b2"])
        elif pooling == "gated_attention":
            sg.fc1[0].weight.copy_(pp["wa"]); sg.fc1[0].bias.copy_(pp["ba"])
            sg.score[0].weight.copy_(pp["wg"]); sg.score[0].bias.copy_(pp["bg"])
            sg.fc2.weight.copy_(pp["w2"]); sg.fc2.bias.copy_(pp["b2"])
    model = model.cuda().eval()
    Xd = X[None].cuda()
    with torch.no_grad():
        logits, img, txt = model(Xd)
        if "attn" in fx:
            v, attn = enc(Xd, ret_with_attn=True)
            assert np.abs(attn.cpu().numpy().ravel() - fx["attn"].ravel()).max() < 1e-4
            assert np.abs(v.cpu().numpy().ravel() - fx["v"].ravel()).max() < 1e-4
    assert np.abs(logits.cpu().numpy() - fx["logits"]).max() < TOL
    assert np.abs(img.cpu().numpy() - fx["image_features"]).max() < 1e-5
    if "attn" in fx:
        # utils/model_inference.py:146-178 through the drop-in DeepMIL(ret_with_attn=True): RAW scores out of the encoder, the
        # caller's softmax over the patches, predicted incidence = softmax of the bag logits
        from vlsa_amd.inference import calc_abmil_text_img_similarity
        attn_w, probs = calc_abmil_text_img_similarity(model, X[None])
        raw = torch.from_numpy(np.asarray(fx["attn"])).reshape(1, -1)
        assert not attn_w.is_cuda and tuple(attn_w.shape) == (1, N) and abs(float(attn_w.sum()) - 1.0) < 1e-5
        assert (attn_w - torch.softmax(raw, dim=-1)).abs().max().item() < 1e-5
        assert (probs - torch.softmax(torch.from_numpy(np.asarray(fx["logits"])), dim=-1)).abs().max().item() < 1e-5
    logits2, _, _ = model(Xd)
    assert np.abs(logits2.detach().cpu().numpy() - fx["logits"]).max() < TOL
    (logits2 * H.t(fx["G"]).cuda()).sum().backward()
    chk = lambda key, g: cases.check_big(fx, key, g, atol=GRAD_ATOL, rtol=GRAD_RTOL)  # noqa: E731
    chk("grad.logit_scale", model.logit_scale.grad)
    chk("grad.T", tp.T.grad)
    chk("grad.adapter.down", enc.visual_adapter.fc[0].weight.grad)
    chk("grad.adapter.up", enc.visual_adapter.fc[2].weight.grad)
    if pooling in ("attention", "gated_attention"):
        for k, g in pool_grads(enc.sigma).items():
            chk("grad.pool." + k, g)


def test_backward_full_size_vs_oracle_autograd():
    """N = 10k fp32 and 20k bf16: dQ from the HIP backward vs torch.autograd through the CPU oracle."""
    from oracle import vlsa_oracle as O
    from vlsa_amd import functional as F
    for N, dt in ((10_000, torch.float32), (20_000, torch.bfloat16)):
        X = cases.make_bag(N, 77).to(dt)
        params = cases.make_params(12, 4, 78)
        Q = (0.5 * params["resid"] + params["prompt"]).requires_grad_(True)
        g = cases.gen(79)
        G = torch.randn(12, 512, generator=g)
        ref = O.vlfan_forward(X.float(), Q)
        (ref["out"] * G).sum().backward()
        Qd = Q.detach().cuda().requires_grad_(True)
        out, _ = F.vlfan_cross_attention(X.cuda(), Qd)
        (out * G.cuda()).sum().backward()
        scale = Q.grad.abs().max().item()
        assert (out.detach().cpu() - ref["out"].detach()).abs().max().item() < 1e-4 * max(1.0, ref["out"].abs().max().item())
        cases.record_grad_error("dQ", (Qd.grad.cpu() - Q.grad).abs().max().item(), scale, GRAD_RTOL * scale)
        assert (Qd.grad.cpu() - Q.grad).abs().max().item() < GRAD_RTOL * scale


def test_interpretation_matches_reference_fixture():
    """calc_text_img_similarity / prototype SHAP vs the reference's outputs (tests/golden/interpretation.npz)."""
    from vlsa_amd.inference import calc_text_img_similarity, evaluate_prototype_shap_imp
    fx = H.load_fixture("interpretation")
    N, P, K, seed = 512, 8, 8, 401
    X = cases.make_bag(N, seed, "clustered")
    params = cases.make_params(P, K, seed + 1000)
    case = ("interp", N, P, K, "mean", "default", False, "clustered", seed, False)
    model, _ = build_vlsa(case, params, {})
    for axis in ("V", "L"):
        _, A, cottn, probs, probs2, dec_imp, shap = calc_text_img_similarity(model, X[None], axis_softmax=axis)
        assert np.abs(A.numpy()[:, ::8] - fx[f"{axis}.A"]).max() < 1e-4
        assert np.abs(cottn.numpy()[:, ::8] - fx[f"{axis}.cottn"]).max() < 1e-4
        assert np.abs(probs.numpy() - fx[f"{axis}.probs"]).max() < 1e-4
        assert np.abs(probs2.numpy() - fx[f"{axis}.probs2"]).max() < 1e-4
        assert np.abs(dec_imp.numpy() - fx[f"{axis}.decoupled_imp"]).max() < 1e-4
        assert np.abs(shap.numpy() - fx[f"{axis}.shap"]).max() < 2e-4
    s = evaluate_prototype_shap_imp(fx["shap_in"], 56.31)
    assert np.abs(s.numpy() - fx["shap_out"]).max() < 1e-5


def test_forward_bags_matches_per_bag_forward():
    (name, N, P, K, pooling, head, gated, kind, seed, grads) = cases.VLFAN_CASES[4]
    X, params, pool = H.vlfan_case_inputs(cases.VLFAN_CASES[4])
    model, _ = build_vlsa(cases.VLFAN_CASES[4], params, pool)
    bags = [cases.make_bag(n, 500 + i).to(torch.bfloat16).cuda() for i, n in enumerate((2798, 100, 5000, 33, 1))]
    with torch.no_grad():
        logits, feats, txt = model.forward_bags(bags)
        for i, xb in enumerate(bags):
            l1, f1, t1 = model(xb[None])
            assert (logits[i] - l1[0]).abs().max().item() < 2e-5
            assert (feats[i] - f1[0]).abs().max().item() < 1e-5
    assert logits.shape == (5, K)


@pytest.mark.parametrize("pooling", ["attention", "gated_attention"])
def test_forward_bags_with_module_pooling_matches_per_bag_forward(pooling):
    """query pooling by a module is not fusable into the head kernel: forward_bags still batches the aggregation."""
    from vlsa_amd.vlsa import VLSA
    dev = torch.device("cuda", 0)
    P, K = 12, 4
    params = cases.make_params(P, K, 9300)
    torch.manual_seed(3)
    cfg = dict(name="VLFAN", dim_in=512, dim_hid=64, use_feat_proj=False, query="Parameter", num_query=P, query_pooling=pooling)
    m = VLSA.from_modules(cfg, pretrained_text_features=params["T"].clone()).to(dev).eval()
    bags = [cases.make_bag(n, 9310 + i).to(torch.bfloat16).to(dev)[None] for i, n in enumerate([700, 64, 2798, 1])]
    with torch.no_grad():
        lb, fb, tb = m.forward_bags(bags)
        ref = [m(x) for x in bags]
    assert (lb - torch.cat([r[0] for r in ref])).abs().max().item() < 1e-4
    assert (fb - torch.cat([r[1] for r in ref])).abs().max().item() < 1e-5
    assert (tb - ref[0][2]).abs().max().item() < 1e-6


@pytest.mark.parametrize("P,K", [(12, 12), (8, 4), (1, 4), (16, 8)])
def test_prototype_shapley_device_matches_host_and_sums_to_the_full_value(P, K):
    """vlsa_prototype_shapley (one thread per coalition) vs the vectorised host restatement of utils/model_inference.py:23-79
    (pinned to the reference by interpretation.npz above); efficiency: the values sum to v(all) - v(empty)."""
    from vlsa_amd.inference import evaluate_prototype_shap_imp
    sim = (torch.rand(P, K, generator=cases.gen(77 + P)) * 2 - 1) * 0.08
    ls = 56.31
    host = evaluate_prototype_shap_imp(sim, ls)
    dev = evaluate_prototype_shap_imp(sim.cuda(), ls)
    assert dev.shape == (P,) and not dev.is_cuda
    assert (dev - host).abs().max().item() < 1e-5
    prob = torch.softmax(ls * sim.mean(dim=0), dim=-1)
    v_full = (prob * torch.arange(K, 0, -1, dtype=torch.float32)).sum().item()
    assert abs(dev.sum().item() - (v_full - 1.0)) < 1e-4
