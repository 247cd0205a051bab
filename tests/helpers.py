"""Shared helpers for the parity tests: fixture loading and oracle evaluation of a golden case."""
import os

import numpy as np
import torch

import cases
from oracle import vlsa_oracle as O

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_fixture(name):
    return dict(np.load(os.path.join(GOLDEN, name + ".npz")))


def t(a):
    return torch.from_numpy(np.asarray(a))


def vlfan_case_inputs(case):
    (name, N, P, K, pooling, head, gated, kind, seed, grads) = case
    X = cases.bag_for_case(N, kind, seed)
    params = cases.make_params(P, K, seed + 1000, gated, aligned_to=X if kind == "adversarial" else None)
    pool = cases.make_pool_params(pooling, seed + 3000)
    if pooling == "weight":
        pool = {"weight": pool["weight"][:, :P].clone()}
    return X, params, pool


def check_inputs(fx, X, params):
    cs = np.array(cases.checksum(X))
    assert np.allclose(cs, fx["x_checksum"], rtol=1e-9, atol=1e-9), "seeded bag differs from the fixture's"
    if "param_checksum" in fx:
        pc = np.array([float(params["W"].double().sum()), float(params["T"].double().sum())])
        assert np.allclose(pc, fx["param_checksum"], rtol=1e-9, atol=1e-9)


def oracle_vlfan_case(case, dtype=torch.float32, requires_grad=False):
    """Evaluate the oracle on a VLFAN golden case. Returns (result dict, leaves dict)."""
    (name, N, P, K, pooling, head, gated, kind, seed, grads) = case
    X, params, pool = vlfan_case_inputs(case)
    X = X.to(dtype)
    leaves = {}
    resid = params["resid"].to(dtype).clone().requires_grad_(requires_grad)
    prompt = params["prompt"].to(dtype)
    Q = 0.5 * resid + prompt
    leaves["resid"] = resid
    Tt = params["T"].to(dtype).clone().requires_grad_(requires_grad)
    ls = torch.tensor(cases.LOGIT_SCALE, dtype=dtype, requires_grad=requires_grad)
    leaves["T"], leaves["logit_scale"] = Tt, ls
    W = b = None
    if head != "Identity":
        W = params["W"].to(dtype).clone().requires_grad_(requires_grad)
        b = params["b"].to(dtype).clone().requires_grad_(requires_grad)
        leaves["W"], leaves["b"] = W, b
    pp = {k: v.to(dtype).clone().requires_grad_(requires_grad) for k, v in pool.items()}
    for k, v in pp.items():
        leaves["pool." + k] = v
    r = O.vlsa_vlfan_forward(X, Q, Tt, ls, gated_query=gated, query_pooling_method=pooling,
                             pooling_params=pp, head_weight=W, head_bias=b)
    return r, leaves
