"""Oracle vs the round-2 golden vectors (produced by running the reference: tests/golden/make_golden_r2.py):
Feat_Projecter in front of VLFAN / DeepMIL (forward + autograd incl. the projecter), DeepMIL's Linear head, every
PromptAdapter method (+ negative prompt), DeepMIL attention poolings on bf16-rounded bags.  CPU only."""
import numpy as np
import pytest
import torch

import cases
import helpers as H
from oracle import vlsa_oracle as O

FWD_TOL = 2e-5
GRAD_RTOL = 2e-4


def _leaf(t):
    return t.clone().requires_grad_(True)


def oracle_featproj_case(case, grad=True):
    (name, enc_name, N, P, K, pooling, seed) = case
    X = cases.make_bag(N, seed)
    params = cases.make_params(max(P, 1), K, seed + 1000)
    fp = {k: (_leaf(v) if grad else v) for k, v in cases.make_featproj_params(seed + 5000).items()}
    fpt = (fp["w"], fp["b"], fp["gamma"], fp["beta"])
    T, ls = (_leaf(params["T"]) if grad else params["T"]), torch.tensor(cases.LOGIT_SCALE, requires_grad=grad)
    leaves = {"fp." + k: v for k, v in fp.items()}
    leaves.update(T=T, logit_scale=ls)
    if enc_name == "VLFAN":
        Q = _leaf(0.5 * params["resid"] + params["prompt"]) if grad else 0.5 * params["resid"] + params["prompt"]
        W, b = (_leaf(params["W"]), _leaf(params["b"])) if grad else (params["W"], params["b"])
        leaves.update(Q=Q, W=W, b=b)
        r = O.vlfan_forward(X, Q, query_pooling_method=pooling, head_weight=W, head_bias=b, feat_proj=fpt)
        v = r["v"]
    else:
        pp = {k: (_leaf(t) if grad else t) for k, t in cases.make_pool_params(pooling, seed + 3000).items()}
        ad = {k: (_leaf(t) if grad else t) for k, t in cases.make_adapter_params(seed + 4000).items()}
        leaves.update({"pool." + k: t for k, t in pp.items()})
        leaves.update({"adapter." + k: t for k, t in ad.items()})
        v = O.deepmil_forward(X, pooling, pp, pred_head="Adapter", adapter=(ad["down"], ad["up"]), feat_proj=fpt)["v"]
    logits, vn, Tn = O.vlsa_logits(v[None, :], T, ls)
    return X, logits, vn, leaves


@pytest.mark.parametrize("case", cases.FEATPROJ_CASES, ids=[c[0] for c in cases.FEATPROJ_CASES])
def test_feat_projecter_cases_match_reference(case):
    fx = H.load_fixture("featproj_" + case[0])
    X, logits, vn, leaves = oracle_featproj_case(case)
    assert np.allclose(np.array(cases.checksum(X)), fx["x_checksum"], rtol=1e-9, atol=1e-9)
    assert np.abs(logits.detach().numpy() - fx["logits"]).max() < FWD_TOL
    assert np.abs(vn.detach().numpy() - fx["image_features"]).max() < 1e-6
    (logits * H.t(fx["G"])).sum().backward()
    checked = 0
    for name, leaf in leaves.items():
        key = "grad." + name
        if key in fx or key + "@rows" in fx:
            cases.check_big(fx, key, leaf.grad, atol=2e-5, rtol=GRAD_RTOL)
            checked += 1
    assert checked >= 6 and all(("grad.fp." + k in fx) or ("grad.fp." + k + "@rows" in fx) for k in ("w", "b", "gamma", "beta"))


@pytest.mark.parametrize("case", cases.DEEPMIL_HEAD_CASES, ids=[c[0] for c in cases.DEEPMIL_HEAD_CASES])
def test_deepmil_linear_head_matches_reference(case):
    (name, N, K, pooling, seed) = case
    fx = H.load_fixture("deepmil_" + name)
    X = cases.make_bag(N, seed)
    params = cases.make_params(1, K, seed + 1000)
    pp = {k: _leaf(t) for k, t in cases.make_pool_params(pooling, seed + 3000).items()}
    gp = {k: _leaf(t) for k, t in cases.make_linear_params(seed + 4000, cases.D, cases.D).items()}
    v = O.deepmil_forward(X, pooling, pp, pred_head="default", g_weight=gp["w"], g_bias=gp["b"])["v"]
    logits, vn, _ = O.vlsa_logits(v[None, :], params["T"], torch.tensor(cases.LOGIT_SCALE))
    assert np.abs(logits.detach().numpy() - fx["logits"]).max() < FWD_TOL
    assert np.abs(vn.detach().numpy() - fx["image_features"]).max() < 1e-6
    (logits * H.t(fx["G"])).sum().backward()
    cases.check_big(fx, "grad.g.w", gp["w"].grad, atol=2e-5, rtol=GRAD_RTOL)
    cases.check_big(fx, "grad.g.b", gp["b"].grad, atol=2e-5, rtol=GRAD_RTOL)
    for k, t in pp.items():
        cases.check_big(fx, "grad.pool." + k, t.grad, atol=2e-5, rtol=GRAD_RTOL)


def prompt_adapter_state(method, P, neg, seed):
    """The seeded tensors of a PROMPT_ADAPTER_CASES entry, in the generator's draw order."""
    g = cases.gen(seed)
    st = dict(feats=torch.randn(P, cases.D, generator=g))
    if neg:
        st["neg"] = torch.randn(1, cases.D, generator=g)
    if method == "TaskRes":
        st["resid"] = torch.randn(P, cases.D, generator=g)
        if neg:
            st["neg_resid"] = torch.randn(1, cases.D, generator=g)
    elif method == "Adapter":
        st["adapter"] = cases.make_adapter_params(seed + 4000)
    elif method == "FC":
        st["fc"] = cases.make_linear_params(seed + 4000, cases.D, cases.D, bias=False)["w"]
    return st


@pytest.mark.parametrize("case", cases.PROMPT_ADAPTER_CASES, ids=[c[0] for c in cases.PROMPT_ADAPTER_CASES])
def test_prompt_adapter_methods_match_reference(case):
    (name, method, P, neg, seed) = case
    fx = H.load_fixture("padapter_" + name)
    st = prompt_adapter_state(method, P, neg, seed)
    kw = {}
    if method == "TaskRes":
        kw = dict(residual=_leaf(st["resid"]), neg_residual=_leaf(st["neg_resid"]) if neg else None)
    elif method == "Adapter":
        kw = dict(adapter=(_leaf(st["adapter"]["down"]), _leaf(st["adapter"]["up"])))
    elif method == "FC":
        kw = dict(fc_weight=_leaf(st["fc"]))
    # the reference's Adapter / default methods ignore the negative prompt in forward(); TaskRes / FC append it
    Q = O.prompt_adapter_forward(method, st["feats"], neg_prompt_features=st.get("neg"), **kw)
    assert Q.shape == fx["Q"].shape
    assert np.abs(Q.detach().numpy() - fx["Q"]).max() < 1e-6
    if "G" in fx:
        (Q * H.t(fx["G"])).sum().backward()
        if method == "TaskRes":
            cases.check_big(fx, "grad.residual_features", kw["residual"].grad, atol=1e-6)
            if neg:
                cases.check_big(fx, "grad.neg_residual_features", kw["neg_residual"].grad, atol=1e-6)
        elif method == "Adapter":
            cases.check_big(fx, "grad.adapter.fc.0.weight", kw["adapter"][0].grad, atol=2e-5, rtol=GRAD_RTOL)
            cases.check_big(fx, "grad.adapter.fc.2.weight", kw["adapter"][1].grad, atol=2e-5, rtol=GRAD_RTOL)
        elif method == "FC":
            cases.check_big(fx, "grad.fc.0.weight", kw["fc_weight"].grad, atol=2e-5, rtol=GRAD_RTOL)


@pytest.mark.parametrize("case", cases.DEEPMIL_BF16_CASES, ids=[c[0] for c in cases.DEEPMIL_BF16_CASES])
def test_deepmil_bf16_bag_matches_reference(case):
    (name, N, K, pooling, seed) = case
    fx = H.load_fixture("deepmil_" + name)
    X = cases.make_bag(N, seed, "iid", torch.bfloat16)
    params = cases.make_params(1, K, seed + 1000)
    pp = cases.make_pool_params(pooling, seed + 3000)
    ad = cases.make_adapter_params(seed + 4000)
    r = O.deepmil_forward(X, pooling, pp, pred_head="Adapter", adapter=(ad["down"], ad["up"]))
    logits, vn, _ = O.vlsa_logits(r["v"][None, :], params["T"], torch.tensor(cases.LOGIT_SCALE))
    assert np.abs(logits.numpy() - fx["logits"]).max() < FWD_TOL
    assert np.abs(r["raw"].numpy().ravel() - fx["attn"].ravel()).max() < 2e-6
    assert np.abs(r["v"].numpy().ravel() - fx["v"].ravel()).max() < 1e-5
