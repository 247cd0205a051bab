"""The per-bag TRAINING route (``VLSA.forward`` with a gradient needed -> ``VF.slide_train``: one autograd node per bag over
``vlsa_vlfan_forward_bag`` / ``vlsa_vlfan_backward_bag``) -- the loop shape of the reference's handler
(runner/vlsa_handler.py:267-289: one ``net(X)`` per bag, ``torch.cat``, ONE backward).

The values and gradients of that route are pinned against the reference-generated fixtures by
tests/test_gpu_modules.py::test_vlsa_vlfan_forward_backward (every mean-pooling case now takes it) and
tests/test_gpu_handler_loop.py; here: that the route IS taken, that it agrees with the general autograd route and with the
batched one on a handler-shaped step of ragged bags, gradients through the returned unit features, and the case of a backward
that runs after the plan's prepared block has moved on."""
import numpy as np
import pytest
import torch

import cases
import helpers as H
from test_gpu_modules import build_vlsa

pytestmark = pytest.mark.gpu

CASE = ("n2798_shipped", 2798, 12, 12, "mean", "default", False, "iid", 105, True)
GATED = ("n300_gatedq", 300, 8, 8, "mean", "default", True, "iid", 109, True)
IDENT = ("n4096_p13_id", 4096, 13, 4, "mean", "Identity", False, "iid", 106, True)


def _model(case):
    X, params, pool = H.vlfan_case_inputs(case)
    model, tp = build_vlsa(case, params, pool)
    return model.train(), tp


def _trainable(model, tp):
    enc = model.mil_encoder
    out = {"logit_scale": model.logit_scale, "T": tp.T}
    if isinstance(enc.visual_adapter, torch.nn.Linear):
        out["W"], out["b"] = enc.visual_adapter.weight, enc.visual_adapter.bias
    out["Q"] = enc.Q if isinstance(enc.Q, torch.nn.Parameter) else enc.Q.residual_features
    return out


def _grads(model, tp):
    g = {k: p.grad.detach().clone() for k, p in _trainable(model, tp).items()}
    model.zero_grad(set_to_none=True)
    tp.zero_grad(set_to_none=True)
    return g


def _close(a, b, what, rtol=1e-4):
    a, b = a.float().cpu().numpy(), b.float().cpu().numpy()
    cases.record_grad_error(str(what), np.abs(a - b).max(), np.abs(b).max(), rtol * np.abs(b).max() + 2e-6)
    assert np.abs(a - b).max() <= rtol * np.abs(b).max() + 2e-6, (what, np.abs(a - b).max(), np.abs(b).max())


def _bags(sizes, dtype, seed=7):
    g = torch.Generator().manual_seed(seed)
    return [(torch.randn(n, 512, generator=g) + 0.3).to(dtype).cuda() for n in sizes]


def test_the_route_is_taken_and_only_when_a_gradient_is_needed():
    model, tp = _model(CASE)
    X = _bags([700], torch.bfloat16)[0][None]
    out = model(X)
    assert type(out[0].grad_fn).__name__ == "_SlideTrainFnBackward"
    assert out[0].shape == (1, 12) and out[1].shape == (1, 512) and out[2].shape == (12, 512)
    with torch.no_grad():
        ev = model(X)
    assert ev[0].grad_fn is None
    assert np.abs(ev[0].cpu().numpy() - out[0].detach().cpu().numpy()).max() < 2e-5
    # a bag that carries a gradient itself is not this route's business
    Xg = X.float().clone().requires_grad_(True)
    assert type(model(Xg)[0].grad_fn).__name__ != "_SlideTrainFnBackward"


@pytest.mark.parametrize("case", [CASE, GATED, IDENT], ids=["shipped", "gated_query", "identity_head"])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16], ids=["fp32", "bf16"])
def test_handler_shaped_step_equals_the_general_and_the_batched_route(case, dtype, monkeypatch):
    model, tp = _model(case)
    K = case[3]
    bags = _bags([1, 37, 300, 2798, 6001, 64], dtype)
    Gm = torch.randn(len(bags), K, generator=torch.Generator().manual_seed(3)).cuda()

    def step():
        logits = torch.cat([model(x[None])[0] for x in bags], dim=0)
        (logits * Gm).sum().backward()
        return logits.detach().clone(), _grads(model, tp)

    fast_logits, fast = step()
    # the batched route (persistent multi-bag kernels + batched head)
    lb = model.forward_bags(bags)[0]
    (lb * Gm).sum().backward()
    batched = _grads(model, tp)
    # the general per-bag route (autograd over the aggregation function + torch ops)
    monkeypatch.setattr(type(model), "_slide_train", lambda self, X, T: None)
    gen_logits, general = step()
    assert np.abs(fast_logits.cpu().numpy() - gen_logits.cpu().numpy()).max() < 1e-4
    assert np.abs(fast_logits.cpu().numpy() - lb.detach().cpu().numpy()).max() < 1e-4
    for k in general:
        _close(fast[k], general[k], f"{k} vs general")
        _close(fast[k], batched[k], f"{k} vs batched")


def test_gradients_through_the_returned_unit_features(monkeypatch):
    model, tp = _model(CASE)
    X = _bags([900], torch.float32)[0][None]
    gen = torch.Generator().manual_seed(11)
    gv, gt, gl = torch.randn(1, 512, generator=gen).cuda(), torch.randn(12, 512, generator=gen).cuda(), torch.randn(1, 12, generator=gen).cuda()

    def run():
        logits, img, txt = model(X)
        ((logits * gl).sum() + (img * gv).sum() + (txt * gt).sum()).backward()
        return _grads(model, tp)

    fast = run()
    monkeypatch.setattr(type(model), "_slide_train", lambda self, X, T: None)
    general = run()
    for k in general:
        _close(fast[k], general[k], k)


def test_backward_after_the_prepared_block_has_moved_on(monkeypatch):
    """forward(bag A) -> the query parameters change in place -> forward(bag B) re-prepares the plan's block -> the backward of
    A's loss must still use A's queries (its block is rebuilt from the tensors the node saved)."""
    model, tp = _model(CASE)
    A, Bb = (x[None] for x in _bags([500, 800], torch.bfloat16))
    gl = torch.randn(1, 12, generator=torch.Generator().manual_seed(5)).cuda()

    def run():
        la = model(A)[0]
        with torch.no_grad():
            model.mil_encoder.Q.residual_features.add_(0.05)
        lb = model(Bb)[0]
        (la * gl).sum().backward()
        ga = _grads(model, tp)
        (lb * gl).sum().backward()
        gb = _grads(model, tp)
        with torch.no_grad():                      # back to the start for the second run
            model.mil_encoder.Q.residual_features.sub_(0.05)
        return la.detach().clone(), lb.detach().clone(), ga, gb

    la, lb, ga, gb = run()
    plan = next(iter(model._train_plans.values()))
    assert plan.gen >= 2                           # bag B did re-prepare
    monkeypatch.setattr(type(model), "_slide_train", lambda self, X, T: None)
    la2, lb2, ga2, gb2 = run()
    assert np.abs(la.cpu().numpy() - la2.cpu().numpy()).max() < 1e-4
    assert np.abs(lb.cpu().numpy() - lb2.cpu().numpy()).max() < 1e-4
    assert np.abs(la.cpu().numpy() - lb.cpu().numpy()).max() > 1e-3      # the edit matters
    for k in ga2:
        _close(ga[k], ga2[k], f"A:{k}")
        _close(gb[k], gb2[k], f"B:{k}")


@pytest.mark.parametrize("shape", [(1, 4, 4, False), (2, 16, 64, False), (70, 1, 1, False), (513, 15, 3, True)], ids=lambda s: f"N{s[0]}_P{s[1]}_K{s[2]}" + ("_gated" if s[3] else ""))
def test_edge_shapes_against_the_oracle(shape):
    """Smallest / largest query and class counts, one- and two-patch bags, rows with a stride: logits and every gradient of the
    per-bag node vs the CPU oracle's autograd (model/deepmil.py:187-204, model/vlsa.py:188-192 restated in oracle/vlsa_oracle.py)."""
    from oracle import vlsa_oracle as O
    N, P, K, gated = shape
    case = ("edge", N, P, K, "mean", "default", gated, "iid", 700 + N, True)
    model, tp = _model(case)
    wide = _bags([N], torch.float32, seed=N)[0].repeat(1, 2)              # [N, 1024]: the bag is the left half, row stride 1024
    X = wide[:, :512]
    assert X.stride(0) == 1024
    G = torch.randn(1, K, generator=torch.Generator().manual_seed(N + 1)).cuda()
    logits = model(X[None])[0]
    assert type(logits.grad_fn).__name__ == "_SlideTrainFnBackward"
    (logits * G).sum().backward()
    got = _grads(model, tp)
    enc = model.mil_encoder
    leaves = {k: p.detach().cpu().clone().requires_grad_(True) for k, p in _trainable(model, tp).items()}
    if gated:
        Q = leaves["Q"]
    else:
        Q = enc.Q.res_ratio * leaves["Q"] + enc.Q.get_raw_prompt_features().detach().cpu()
    r = O.vlsa_vlfan_forward(X.detach().cpu(), Q, leaves["T"], leaves["logit_scale"], head_weight=leaves["W"], head_bias=leaves["b"],
                             gated_query=gated)
    assert np.abs(logits.detach().cpu().numpy() - r["logits"].detach().numpy()).max() < 1e-4
    (r["logits"] * G.cpu()).sum().backward()
    for k, leaf in leaves.items():
        _close(got[k], leaf.grad, k)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16], ids=["fp32", "bf16"])
def test_trainable_feat_projecter_in_front_of_the_per_bag_node(dtype, monkeypatch):
    """use_feat_proj=True (the constructor default of the reference's encoders, model/deepmil.py:75) with a projecter that trains:
    the projecter is its own HIP node, the per-bag node hands dL/dX of the aggregation back to it (vlsa_vlfan_backward_dx).  Against
    the general route (autograd over the aggregation function + torch ops) on the same bags: logits and every gradient."""
    from vlsa_amd.prompt_adapter import PromptAdapter
    from vlsa_amd.vlsa import VLSA
    from test_gpu_modules import TextParam
    P = K = 12
    params = cases.make_params(P, K, 4711)
    tp = TextParam(params["T"])
    qnet = PromptAdapter(method="TaskRes", num_prompts=P, pretrained_prompt_features=params["prompt"], res_ratio=0.5)
    cfg = dict(name="VLFAN", dim_in=512, dim_hid=256, use_feat_proj=True, drop_rate=0.25, num_query=P, query="Text", query_pooling="mean",
               pred_head="default")
    torch.manual_seed(3)
    model = VLSA.from_modules(cfg, text_provider=lambda: tp.T, prompt_learner=tp, query_network=qnet, logit_scale_init=cases.LOGIT_SCALE)
    with torch.no_grad():
        model.mil_encoder.Q.residual_features.copy_(params["resid"])
    model = model.cuda().train()
    enc = model.mil_encoder
    assert enc.feat_proj is not None and all(p.requires_grad for p in enc.feat_proj.parameters())
    bags = _bags([40, 700, 3001], dtype)
    Gm = torch.randn(len(bags), K, generator=torch.Generator().manual_seed(8)).cuda()
    named = dict(model.named_parameters())
    named["T"] = tp.T

    def step():
        logits = torch.cat([model(x[None])[0] for x in bags], dim=0)
        (logits * Gm).sum().backward()
        g = {k: p.grad.detach().clone() for k, p in named.items() if p.grad is not None}
        model.zero_grad(set_to_none=True); tp.zero_grad(set_to_none=True)
        return logits, g

    fast_logits, fast = step()
    assert type(fast_logits.grad_fn.next_functions[0][0]).__name__ == "_SlideTrainFnBackward"
    monkeypatch.setattr(type(model), "_slide_train", lambda self, X, T: None)
    gen_logits, general = step()
    assert type(gen_logits.grad_fn.next_functions[0][0]).__name__ != "_SlideTrainFnBackward"
    assert np.abs(fast_logits.detach().cpu().numpy() - gen_logits.detach().cpu().numpy()).max() < 1e-4
    assert set(fast) == set(general) and any(k.startswith("mil_encoder.feat_proj") for k in fast)
    for k in general:
        _close(fast[k], general[k], k)


def test_fuzz_against_the_general_route():
    """tools/fuzz_slide_train.py, 40 random configurations (N, P, K, dtype, gated query, adapter, row stride)."""
    import subprocess, sys, os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "fuzz_slide_train.py"), "40"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "40 random cases" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]
