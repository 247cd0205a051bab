"""Oracle vs the committed golden vectors (which were produced by running the reference itself).

This is the pin of the oracle: fp32, reference op order, tolerances a few ulp of the
quantities involved.  CPU only.
"""
import numpy as np
import pytest
import torch

import cases
import helpers as H
from oracle import vlsa_oracle as O

FWD_TOL = 2e-5      # logits are O(10); measured differences are ~1e-6
GRAD_RTOL = 2e-4


@pytest.mark.parametrize("case", cases.VLFAN_CASES, ids=[c[0] for c in cases.VLFAN_CASES])
def test_vlfan_forward_matches_reference(case):
    fx = H.load_fixture("vlfan_" + case[0])
    X, params, _ = H.vlfan_case_inputs(case)
    H.check_inputs(fx, X, params)
    r, _ = H.oracle_vlfan_case(case)
    assert np.abs(r["logits"].numpy() - fx["logits"]).max() < FWD_TOL
    assert np.abs(r["v_hat"].numpy() - fx["image_features"]).max() < 1e-6
    assert np.abs(r["T_hat"].numpy() - fx["text_features"]).max() < 1e-6
    assert np.abs(r["A"].numpy() - fx["A"]).max() < 2e-6
    assert np.abs(r["A"].sum(dim=1).numpy() - 1).max() < 1e-5
    if "pool_ext" in fx:
        assert np.abs(r["pool_ext"].numpy().ravel() - fx["pool_ext"].ravel()).max() < 1e-5


@pytest.mark.parametrize("case", [c for c in cases.VLFAN_CASES if c[-1]], ids=[c[0] for c in cases.VLFAN_CASES if c[-1]])
def test_vlfan_backward_matches_reference(case):
    fx = H.load_fixture("vlfan_" + case[0])
    r, leaves = H.oracle_vlfan_case(case, requires_grad=True)
    (r["logits"] * H.t(fx["G"])).sum().backward()
    gated = case[6]
    for key, leaf in (("grad.logit_scale", "logit_scale"), ("grad.T", "T"), ("grad.b", "b"), ("grad.W", "W")):
        if leaf in leaves and (key in fx or key + "@rows" in fx):
            cases.check_big(fx, key, leaves[leaf].grad, atol=2e-5, rtol=GRAD_RTOL)
    # grad wrt Q: reference stores residual grad (TaskRes: dQ * 0.5) or the Parameter grad itself
    if gated:
        cases.check_big(fx, "grad.Q", 2.0 * leaves["resid"].grad, atol=2e-5, rtol=GRAD_RTOL)
    else:
        cases.check_big(fx, "grad.resid", leaves["resid"].grad, atol=2e-5, rtol=GRAD_RTOL)
    for k, leaf in leaves.items():
        if k.startswith("pool."):
            cases.check_big(fx, "grad." + k, leaf.grad, atol=2e-5, rtol=GRAD_RTOL)


@pytest.mark.parametrize("case", cases.VLFAN_CASES, ids=[c[0] for c in cases.VLFAN_CASES])
def test_single_pass_form_equals_reference_order(case):
    """The shardable online-softmax restatement (what the kernels compute) == the reference order."""
    (name, N, P, K, pooling, head, gated, kind, seed, grads) = case
    X, params, _ = H.vlfan_case_inputs(case)
    Q = 0.5 * params["resid"] + params["prompt"]
    ref = O.vlfan_forward(X.double(), Q.double(), gated_query=gated)
    for nshard in (1, 3, 8):
        bounds = [round(i * N / nshard) for i in range(nshard + 1)]
        sh = O.vlfan_forward_sharded(X.double(), Q.double(), bounds, gated_query=gated)
        assert (sh["out"] - ref["out"]).abs().max() < 1e-9 * max(1.0, ref["out"].abs().max().item())
        assert (sh["A"] - ref["A"]).abs().max() < 1e-10


@pytest.mark.parametrize("case", cases.ZEROSHOT_CASES, ids=[c[0] for c in cases.ZEROSHOT_CASES])
def test_zeroshot_matches_reference(case):
    (name, N, K, pooling, seed) = case
    fx = H.load_fixture("zeroshot_" + name)
    X = cases.make_bag(N, seed)
    params = cases.make_params(1, K, seed + 1000)
    H.check_inputs(fx, X, params={"W": params["W"], "T": params["T"]} if "param_checksum" in fx else params)
    logits, vn, Tn = O.vlsa_zeroshot_forward(X, params["T"], torch.tensor(cases.LOGIT_SCALE), pooling)
    assert np.abs(logits.numpy() - fx["logits"]).max() < FWD_TOL
    if "image_features" in fx:
        assert np.abs(vn.numpy() - fx["image_features"]).max() < 1e-6
    else:
        assert np.abs(vn[:8].numpy() - fx["image_features_rows"]).max() < 1e-6


def _deepmil_oracle(case, requires_grad=False):
    (name, N, K, pooling, seed) = case
    X = cases.make_bag(N, seed)
    params = cases.make_params(1, K, seed + 1000)
    pp = {k: v.clone().requires_grad_(requires_grad) for k, v in cases.make_pool_params(pooling, seed + 3000).items()}
    ad = {k: v.clone().requires_grad_(requires_grad) for k, v in cases.make_adapter_params(seed + 4000).items()}
    T = params["T"].clone().requires_grad_(requires_grad)
    ls = torch.tensor(cases.LOGIT_SCALE, requires_grad=requires_grad)
    r = O.deepmil_forward(X, pooling, pp, pred_head="Adapter", adapter=(ad["down"], ad["up"]), keep_ratio=0.8)
    logits, vn, Tn = O.vlsa_logits(r["v"][None], T, ls)
    return X, r, logits, vn, Tn, dict(pool=pp, adapter=ad, T=T, logit_scale=ls)


@pytest.mark.parametrize("case", cases.DEEPMIL_CASES, ids=[c[0] for c in cases.DEEPMIL_CASES])
def test_deepmil_matches_reference(case):
    fx = H.load_fixture("deepmil_" + case[0])
    X, r, logits, vn, Tn, leaves = _deepmil_oracle(case, requires_grad=True)
    assert np.allclose(np.array(cases.checksum(X)), fx["x_checksum"])
    assert np.abs(logits.detach().numpy() - fx["logits"]).max() < FWD_TOL
    assert np.abs(vn.detach().numpy() - fx["image_features"]).max() < 1e-6
    if "attn" in fx:
        assert np.abs(r["raw"].detach().numpy().ravel() - fx["attn"].ravel()).max() < 1e-5
        assert np.abs(r["v"].detach().numpy().ravel() - fx["v"].ravel()).max() < 1e-5
    (logits * H.t(fx["G"])).sum().backward()
    for k, leaf in leaves["pool"].items():
        cases.check_big(fx, "grad.pool." + k, leaf.grad, atol=2e-5, rtol=GRAD_RTOL)
    cases.check_big(fx, "grad.adapter.down", leaves["adapter"]["down"].grad, atol=2e-5, rtol=GRAD_RTOL)
    cases.check_big(fx, "grad.adapter.up", leaves["adapter"]["up"].grad, atol=2e-5, rtol=GRAD_RTOL)
    cases.check_big(fx, "grad.T", leaves["T"].grad, atol=2e-5, rtol=GRAD_RTOL)
    cases.check_big(fx, "grad.logit_scale", leaves["logit_scale"].grad, atol=2e-5, rtol=GRAD_RTOL)


def test_interpretation_matches_reference():
    fx = H.load_fixture("interpretation")
    N, P, K, seed = 512, 8, 8, 401
    X = cases.make_bag(N, seed, "clustered")
    assert np.allclose(np.array(cases.checksum(X)), fx["x_checksum"])
    params = cases.make_params(P, K, seed + 1000)
    Q = 0.5 * params["resid"] + params["prompt"]
    ls = torch.tensor(cases.LOGIT_SCALE)
    for axis in ("V", "L"):
        A_sm, cottn, probs, probs2, dec_imp, dec = O.decoupled_similarity(
            X, Q, params["T"], ls, params["W"], params["b"], axis_softmax=axis)
        assert np.abs(A_sm.numpy()[:, ::8] - fx[f"{axis}.A"]).max() < 2e-6
        assert np.abs(cottn.numpy()[:, ::8] - fx[f"{axis}.cottn"]).max() < 2e-6
        assert np.abs(probs.numpy() - fx[f"{axis}.probs"]).max() < 1e-5
        assert np.abs(probs2.numpy() - fx[f"{axis}.probs2"]).max() < 1e-5
        assert np.abs(dec_imp.numpy() - fx[f"{axis}.decoupled_imp"]).max() < 1e-5
        shap = O.prototype_shap(dec, float(ls.exp()))
        assert np.abs(shap.numpy() - fx[f"{axis}.shap"]).max() < 1e-4
    shap = O.prototype_shap(H.t(fx["shap_in"]), 56.31)
    assert np.abs(shap.numpy() - fx["shap_out"]).max() < 1e-5


def test_query_div_loss_matches_reference():
    fx = H.load_fixture("query_div")
    for tag in ("plain", "gated"):
        Q = H.t(fx[f"{tag}.Q"])
        assert abs(O.query_div_loss(Q, 6, True).item() - float(fx[f"{tag}.loss_last_div"])) < 1e-6
        assert abs(O.query_div_loss(Q, 6, False).item() - float(fx[f"{tag}.loss_all"])) < 1e-6


def test_concordance_index_matches_reference_fixture():
    """oracle.concordance_index (the metric of the training parity test) vs the reference's evaluator on seeded cases."""
    fx = H.load_fixture("cindex")
    for (n, K, seed) in cases.CINDEX_CASES:
        y, inc = cases.make_cindex_case(n, K, seed)
        assert abs(O.concordance_index(y, inc) - float(fx[f"c{seed}"][0])) < 1e-12, seed
