"""The handler-shaped loop on the HIP model (GPU): the model is built from a cfg_vlsa_conch.yaml-shaped dict through
``load_model('VLSA', **arch_cfg)`` ON THE CPU, moved with ``.cuda()`` (runner/base_handler.py:114), a checkpoint is loaded
with ``strict=False``, then three ``_update_network``-shaped steps and a ``test_model``-shaped evaluation run bag by bag --
the reference handler's call pattern (runner/vlsa_handler.py:260-289,315-345).  Everything is compared with
tests/golden/handler_loop.npz, which tests/golden/make_golden_handler.py produced by running the REFERENCE's own handler code
on the same cfg, stand-in loaders, bags and labels.

Also here: the stale-text-feature hazard of the single-slide path: frozen MIL encoder, trained prompts,
``net(X)`` before and after an optimizer step with the caching allocator forced to hand the old block out again.
"""
import gc
import tempfile

import numpy as np
import pytest
import torch

import cases
import handler_cases as HC
import handler_loop as HL
import helpers as H
from oracle import vlsa_oracle as O

pytestmark = pytest.mark.gpu


@pytest.fixture()
def hooks_installed():
    from vlsa_amd import hooks
    prev_t = hooks.set_tokenizer_factory(HC.make_tokenizer)
    prev_m = hooks.set_vl_model_loader(lambda **kw: HC.make_coca_stub())
    yield
    hooks.set_tokenizer_factory(prev_t)
    hooks.set_vl_model_loader(prev_m)


def _build(**overrides):
    from vlsa_amd.model_utils import load_model
    with tempfile.TemporaryDirectory() as tmp:
        p_init, p_proto = HC.write_prompt_files(tmp)
        cfg = HC.make_cfg(p_init, p_proto, **overrides)
        model = HL.build_model(cfg, load_model).cuda()          # built on the CPU, moved afterwards: as the handler does
    missing, unexpected = model.load_state_dict(HC.mil_state(), strict=False)
    assert not unexpected
    return model, cfg


def _bulk(a, ref, what, lr, steps):
    """Adam's first steps move an entry by ~lr * sign(grad): entries whose gradient is within rounding of zero may go the
    other way -- compare the bulk (see tests/test_train_step.py)."""
    a, ref = np.asarray(a, dtype=np.float64).ravel(), np.asarray(ref, dtype=np.float64).ravel()
    d = np.abs(a - ref)
    assert np.mean(d <= 2e-6) >= 0.995, f"{what}: only {np.mean(d <= 2e-6):.4f} of entries within 2e-6"
    assert d.max() <= 2.5 * lr * steps, f"{what}: max diff {d.max():.2e}"


@pytest.mark.parametrize("fused_loss", [False, True])
def test_handler_loop_reproduces_the_reference_run(hooks_installed, fused_loss):
    fx = H.load_fixture("handler_loop")
    model, cfg = _build()
    # frozen prototype features and initial text features: the HIP tower on the device, deferred from construction
    assert np.abs(model.mil_encoder.Q.get_raw_prompt_features().cpu().numpy() - fx["prompt_features"]).max() < 1e-4
    assert np.abs(model.forward_text_only().detach().cpu().numpy() - fx["text_features0"]).max() < 1e-4
    assert np.abs(model.prompt_learner.context_embeds.detach().cpu().numpy() - fx["context0"]).max() == 0
    opt = HL.make_optimizer(model, cfg)
    groups = [len(g["params"]) for g in opt.param_groups] + [int(g["weight_decay"] > 0) for g in opt.param_groups]
    assert groups == [int(v) for v in fx["optimizer_groups"]]
    if fused_loss:
        from vlsa_amd.losses import SurvObjective
        objective = SurvObjective(weight_ifmle=cfg["loss_survifmle_weight"], weight_emd=cfg["loss_survemd_weight"])
    else:
        objective = O.vlsa_objective          # host-side loss on [4, K]: plain torch ops (the oracle's restatement)
    xs, ys = HC.train_batch()
    xs, ys = [x.cuda() for x in xs], [y.cuda() for y in ys]
    model.train()
    lr = cfg["opt_lr"]
    for step in range(HC.STEPS):
        loss, preds = HL.update_network(model, opt, objective, xs, ys)
        ref_loss = float(fx[f"loss{step}"][0])
        assert abs(loss - ref_loss) < 3e-4 * max(1.0, abs(ref_loss)), (step, loss, ref_loss)
        assert np.abs(preds.numpy() - fx[f"preds{step}"]).max() < 2e-4, step
        if step in (0, HC.STEPS - 1):
            _bulk(model.prompt_learner.context_embeds.detach().cpu(), fx[f"context@{step}"], f"context@{step}", lr, HC.STEPS)
            _bulk(model.prompt_learner.rank_embeds.detach().cpu(), fx[f"rank@{step}"], f"rank@{step}", lr, HC.STEPS)
            _bulk(model.mil_encoder.Q.residual_features.detach().cpu(), fx[f"resid@{step}"], f"resid@{step}", lr, HC.STEPS)
            _bulk(model.mil_encoder.visual_adapter.bias.detach().cpu(), fx[f"b@{step}"], f"b@{step}", lr, HC.STEPS)
            _bulk(model.mil_encoder.visual_adapter.weight.detach().cpu()[list(cases.SAMPLE_ROWS)], fx[f"W@{step}@rows"],
                  f"W@{step}", lr, HC.STEPS)
            assert abs(float(model.logit_scale.detach()) - float(fx[f"logit_scale@{step}"])) < 2e-6
    out = HL.test_model(model, HC.eval_loader())
    assert np.abs(out["raw_y_hat"].numpy() - fx["eval.raw_y_hat"]).max() < 3e-4
    assert np.abs(out["y_hat"].numpy() - fx["eval.y_hat"]).max() < 1e-4
    assert np.abs(model.forward_text_only().detach().cpu().numpy() - fx["text_features_end"]).max() < 1e-4


def test_first_step_gradients_match_the_reference(hooks_installed):
    fx = H.load_fixture("handler_loop")
    model, cfg = _build()
    xs, ys = HC.train_batch()
    xs, ys = [x.cuda() for x in xs], [y.cuda() for y in ys]
    model.train()
    preds = torch.cat([model(x)[0] for x in xs], dim=0)
    label = torch.cat(ys, dim=0)
    O.vlsa_objective(preds, label[:, 0], label[:, 1], model.get_logit_scale()).backward()
    for name, p in (("context", model.prompt_learner.context_embeds), ("rank", model.prompt_learner.rank_embeds),
                    ("resid", model.mil_encoder.Q.residual_features), ("logit_scale", model.logit_scale)):
        g, r = p.grad.detach().cpu().numpy(), fx["grad0." + name]
        assert np.abs(g - r).max() < 2e-3 * np.abs(r).max() + 1e-7, name


def test_single_slide_path_follows_retrained_prompts_when_the_mil_encoder_is_frozen(hooks_installed):
    """Frozen ``mil_encoder`` (a supported configuration: runner/vlsa_handler.py:126-149), prompts trained for one step:
    the query part of the single-slide plan's key never changes, and the text features are a fresh tensor per parameter
    version whose freed block the caching allocator hands out again.  ``net(X)`` must follow the new prompts."""
    model, cfg = _build(vlsa_img_encoder_frozen=True)
    opt = HL.make_optimizer(model, cfg)
    X = cases.make_bag(700, 4242, "iid").cuda()[None]
    xs, ys = HC.train_batch()
    xs, ys = [x.cuda() for x in xs], [y.cuda() for y in ys]
    enc = model.mil_encoder

    def oracle_logits():
        with torch.no_grad():
            T = model.forward_text_only().detach().cpu()
            Q = enc.get_query().detach().cpu()
        r = O.vlsa_vlfan_forward(X[0].cpu(), Q, T, model.logit_scale.detach().cpu(), head_weight=enc.visual_adapter.weight.detach().cpu(),
                                 head_bias=enc.visual_adapter.bias.detach().cpu())
        return r["logits"].numpy()

    seen_ptrs = set()
    for rnd in range(4):
        model.eval()
        with torch.no_grad():
            logits = model(X)[0].cpu().numpy()
            seen_ptrs.add(model._text_features().data_ptr())
        assert np.abs(logits - oracle_logits()).max() < 1e-4, f"round {rnd}: stale text features in the single-slide plan"
        model.train()
        # a large step so that stale features would be far outside the tolerance
        for g in opt.param_groups:
            g["lr"] = 0.05
        HL.update_network(model, opt, O.vlsa_objective, xs, ys)
        model._drop_text_cache()           # free the old text features: the next result may land on the very same address
        gc.collect()
        torch.cuda.empty_cache() if rnd % 2 else None
    # the evaluation-time text features did change between rounds (the step is large) -- and the hazard is real only if
    # an address repeats; either way every round matched the oracle above
    assert len(seen_ptrs) >= 1


# ---- look-ahead: the handler's bag-by-bag evaluation loop served from batched launches -------------------
class _PatchItems(torch.utils.data.Dataset):
    """items shaped like WSIPatchSurv's 'patch' mode (dataset/PatchWSI.py:197-215)"""

    def __init__(self, sizes, seed=77):
        self.feats = [cases.make_bag(n, seed + i, "clustered" if i % 2 else "iid") for i, n in enumerate(sizes)]
        self.uid = [f"p{i}" for i in range(len(sizes))]

    def __len__(self):
        return len(self.feats)

    def __getitem__(self, i):
        return torch.Tensor([i]).to(torch.int), (self.feats[i].to(torch.float), torch.Tensor([0])), torch.Tensor([float(i), 1.0])


def _eval_loop(model, loader):
    """runner/vlsa_handler.py:315-345 (tests/handler_loop.py::test_model) + the launches the model issued"""
    return HL.test_model(model, loader)["raw_y_hat"]


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_lookahead_serves_the_handlers_eval_loop_from_batched_launches(hooks_installed, dtype):
    from vlsa_amd.ingest import ResidentBags, ResidentBagView
    model, cfg = _build()
    sizes = [700, 64, 1, 2798, 333, 4100, 65, 900, 17, 1200] * 8          # 80 items: a full 64-bag window + a 16-bag one
    ds = _PatchItems(sizes)
    rb = ResidentBags(ds, dtype=dtype)
    loader = torch.utils.data.DataLoader(rb, batch_size=1, shuffle=False, num_workers=0)
    model.lookahead_bags = 0
    first = _eval_loop(model, loader)                  # epoch 1 uploads; bag-by-bag route (look-ahead off)
    model.lookahead_bags = 64
    calls = []
    orig = model._forward_bags_fused
    model._forward_bags_fused = lambda bags, tf, **kw: (calls.append(len(bags)), orig(bags, tf, **kw))[1]
    ahead = _eval_loop(model, loader)                  # everything resident: windows of 64 + 16
    assert calls == [64, 16], calls
    # the loader really delivered the resident rows themselves (no collate copy), tagged with their item
    _, data_x, _ = next(iter(loader))
    assert isinstance(data_x[0], ResidentBagView) and data_x[0]._vlsa_src[1] == 0 and data_x[0].data_ptr() == rb.resident_view(0).data_ptr()
    # values: the batched kernels vs the per-bag kernels (different partial sums) and vs the oracle
    assert (ahead - first).abs().max().item() < 5e-5          # both routes are within 1e-4 of the oracle (checked below)
    enc = model.mil_encoder
    with torch.no_grad():
        T, Q = model.forward_text_only().cpu(), enc.get_query().cpu()
    for i in (0, 2, 3, 63, 64, 79):
        x = ds.feats[i].to(dtype).float()
        ref = O.vlsa_vlfan_forward(x, Q, T, model.logit_scale.detach().cpu(), head_weight=enc.visual_adapter.weight.detach().cpu(),
                                   head_bias=enc.visual_adapter.bias.detach().cpu())["logits"]
        assert (ahead[i:i + 1] - ref).abs().max().item() < 1e-4, i
    # bit-equal to what forward_bags hands out for the same window
    with torch.no_grad():
        direct = model.forward_bags([rb.resident_view(j) for j in range(64)])[0].cpu()
    assert torch.equal(direct, ahead[:64])
    # a second pass under unchanged parameters: new windows (only the last one is kept), same numbers, bit for bit
    again = _eval_loop(model, loader)
    assert torch.equal(again, ahead)


def test_lookahead_cannot_serve_stale_rows(hooks_installed):
    """An optimizer step, a load_state_dict, an in-place edit of any tensor the result depends on, or train mode between two
    calls of ONE window: the rows computed before are not handed out again."""
    from vlsa_amd.ingest import ResidentBags
    model, cfg = _build()
    opt = HL.make_optimizer(model, cfg)
    for g in opt.param_groups:
        g["lr"] = 0.05          # a large step: stale rows would be far outside the tolerance
    sizes = [500, 300, 700, 64, 900, 129, 2000, 31]
    ds = _PatchItems(sizes, seed=177)
    rb = ResidentBags(ds, dtype=torch.float32)
    for i in range(len(sizes)):
        rb[i]                                           # upload
    xs, ys = HC.train_batch()
    xs, ys = [x.cuda() for x in xs], [y.cuda() for y in ys]
    enc = model.mil_encoder

    def item(i):
        return torch.utils.data.default_collate([rb[i]])[1][0].cuda()

    def oracle(i):
        with torch.no_grad():
            T, Q = model.forward_text_only().cpu(), enc.get_query().cpu()
        return O.vlsa_vlfan_forward(ds.feats[i], Q, T, model.logit_scale.detach().cpu(), head_weight=enc.visual_adapter.weight.detach().cpu(),
                                    head_bias=enc.visual_adapter.bias.detach().cpu())["logits"]

    def check(i, what):
        model.eval()
        with torch.no_grad():
            got = model(item(i))[0].cpu()
        assert (got - oracle(i)).abs().max().item() < 1e-4, what

    check(0, "first window")
    assert model._la is not None and len(model._la["rows"]) == len(sizes)
    check(1, "served from the window")
    model.train()
    HL.update_network(model, opt, O.vlsa_objective, xs, ys)                  # prompts, queries, adapter, logit scale all move
    check(2, "after an optimizer step")
    with torch.no_grad():
        enc.visual_adapter.bias.add_(0.3)                                     # in-place edit, no forward in between
    check(3, "after an in-place parameter edit")
    sd = {k: v.clone() for k, v in model.state_dict().items()}
    sd["mil_encoder.visual_adapter.weight"] = sd["mil_encoder.visual_adapter.weight"] * 1.5
    model.load_state_dict(sd, strict=False)
    check(4, "after load_state_dict")
    with torch.no_grad():
        model.logit_scale.mul_(0.5)
    check(5, "after a logit-scale edit")
    with torch.no_grad():
        model.prompt_learner.rank_embeds.mul_(1.1)                            # text side: a new text-feature tensor
    check(6, "after a prompt edit")
    # train mode under no_grad (dropout-bearing encoders would be stochastic): never served from a window
    model.train()
    with torch.no_grad():
        model._la = None
        model(item(7))
    assert model._la is None


def test_lookahead_shrinks_under_random_access(hooks_installed):
    from vlsa_amd.ingest import ResidentBags
    model, cfg = _build()
    sizes = [300] * 200
    ds = _PatchItems(sizes, seed=277)
    rb = ResidentBags(ds, dtype=torch.bfloat16)
    for i in range(len(sizes)):
        rb[i]
    model.eval()
    calls = []
    orig = model._forward_bags_fused
    model._forward_bags_fused = lambda bags, tf, **kw: (calls.append(len(bags)), orig(bags, tf, **kw))[1]
    order = [0, 120, 3, 77, 160, 30, 141, 15, 99, 6, 180, 50]
    with torch.no_grad():
        outs = [model(torch.utils.data.default_collate([rb[i]])[1][0])[0] for i in order]
    assert calls == [64, 16, 4], calls                 # 64 -> 16 -> 4 -> 1 = the per-bag route from the fourth access on
    with torch.no_grad():
        for i, o in zip(order, outs):
            ref = model.forward_bags([rb.resident_view(i)])[0]
            assert (o - ref).abs().max().item() < 2e-5
        calls.clear()
        for i in range(100, 140):                      # sequential again: after three in a row the windows come back (8, 16, ...)
            model(torch.utils.data.default_collate([rb[i]])[1][0])
    # (round 5: the first hit inside the newest window already launches the one behind it -- one window ahead of the host -- so the
    # 40 sequential accesses see one more window go out than they consume)
    assert calls[:3] == [8, 16, 32] and len(calls) <= 4, calls


def test_load_vlsa_model_from_a_run_directory(hooks_installed):
    """utils/model_inference.py:11-21: config.yaml + train_model-last.pth -> the model on the device, checkpoint loaded strict=False"""
    import os
    import yaml
    from vlsa_amd.inference import load_vlsa_model
    with tempfile.TemporaryDirectory() as tmp:
        p_init, p_proto = HC.write_prompt_files(tmp)
        cfg = HC.make_cfg(p_init, p_proto)
        with open(os.path.join(tmp, "config.yaml"), "w") as f:
            yaml.safe_dump(cfg, f)
        torch.save({"epoch": 10, "model": HC.mil_state()}, os.path.join(tmp, "train_model-last.pth"))
        model, got_cfg = load_vlsa_model(tmp, cuda_id=0, return_cfg=True)
    assert got_cfg["cuda_id"] == 0 and next(model.parameters()).is_cuda
    for k, v in HC.mil_state().items():
        assert torch.equal(dict(model.state_dict())[k].cpu(), v), k
    fx = H.load_fixture("handler_loop")
    assert np.abs(model.forward_text_only().detach().cpu().numpy() - fx["text_features0"]).max() < 1e-4
