"""Host-side checks of the drop-in modules that need no GPU: constructor surface, state-dict key compatibility with
the reference's shipped checkpoint (key names / shapes pinned in tests/golden/ckpt_keys.json), text-feature cache."""
import json
import os

import pytest
import torch
import torch.nn as nn

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

SHIPPED_CFG = dict(name="VLFAN", frozen=False, dim_in=512, dim_hid=256, use_feat_proj=False, drop_rate=0.25,
                   pred_head="default", dim_reduction=4, keep_ratio=0.8, query="Text", num_query=12,
                   query_pooling="mean", gated_query=False, query_text_method="TaskRes", query_text_res_ratio=0.5)


def _model():
    from vlsa_amd.prompt_adapter import PromptAdapter
    from vlsa_amd.vlsa import VLSA
    qnet = PromptAdapter(method="TaskRes", num_prompts=12, pretrained_prompt_features=torch.randn(12, 512))
    return VLSA.from_modules(SHIPPED_CFG, pretrained_text_features=torch.randn(12, 512), query_network=qnet)


def test_state_dict_keys_match_shipped_checkpoint():
    ck = json.load(open(os.path.join(GOLDEN, "ckpt_keys.json")))["model"]
    sd = _model().state_dict()
    ours = {k: list(v.shape) for k, v in sd.items()}
    # the reference filters `prompt_encoder.*` and never saves non-persistent buffers; prompt_learner.* belongs to the
    # (out-of-scope) text side and is carried by whatever prompt learner the caller plugs in
    for k, shape in ck.items():
        if k.startswith("prompt_learner."):
            continue
        assert k in ours, f"missing state-dict key {k}"
        assert ours[k] == shape, (k, ours[k], shape)
    assert set(ours) == {k for k in ck if not k.startswith("prompt_learner.")}
    # a checkpoint with the shipped key set loads with strict=False, as the reference does (vlsa_handler.py:317-318)
    fake = {k: torch.zeros(s) for k, s in ck.items()}
    missing, unexpected = _model().load_state_dict(fake, strict=False)
    assert not missing
    assert all(k.startswith("prompt_learner.") for k in unexpected)


def test_encoder_lookup_and_kwargs_surface():
    from vlsa_amd import deepmil
    from vlsa_amd.vlsa import build_mil_encoder
    for name in ("VLFAN", "FeatMIL", "DeepMIL"):
        assert hasattr(deepmil, name)
    enc = build_mil_encoder(dict(name="DeepMIL", dim_in=512, dim_hid=256, num_cls=512, use_feat_proj=False,
                                 pooling="gated_attention", pred_head="Adapter", frozen=False, unknown_key=1))
    assert {"sigma.fc1.0.weight", "sigma.score.0.weight", "sigma.fc2.weight", "visual_adapter.fc.0.weight",
            "visual_adapter.fc.2.weight"} <= set(enc.state_dict())
    with pytest.raises(ValueError):
        build_mil_encoder(dict(name="TransMIL"))
    enc = build_mil_encoder(dict(name="VLFAN", dim_in=512, query="Parameter", num_query=7, gated_query=True,
                                 query_pooling="weight", pred_head="Identity", use_feat_proj=False))
    assert enc.Q.shape == (8, 512) and enc.query_pooling.shape == (1, 7)
    assert abs(float(enc.get_coattn_logit_scale()) - 100.0) < 1e-3
    with pytest.raises(AssertionError):
        build_mil_encoder(dict(name="VLFAN", query_pooling="median"))


def test_text_feature_cache_tracks_parameter_versions():
    from vlsa_amd.vlsa import VLSA
    calls = []

    class PL(nn.Module):
        def __init__(self):
            super().__init__()
            self.w = nn.Parameter(torch.ones(4, 512))

    pl = PL()

    def provider():
        calls.append(1)
        return pl.w * 2

    m = VLSA.from_modules(dict(name="FeatMIL", pooling="mean"), text_provider=provider, prompt_learner=pl)
    with torch.no_grad():
        a = m.forward_text_only(); b = m.forward_text_only()
        assert len(calls) == 1 and a is b
        pl.w.add_(1.0)                      # what an optimizer step does
        c = m.forward_text_only()
        assert len(calls) == 2 and torch.allclose(c, torch.full((4, 512), 4.0))
    m.forward_text_only()                    # grad mode differs -> recomputed so the graph exists
    assert len(calls) == 3


def test_text_cache_with_module_provider_and_repeated_backward():
    """a PromptAdapter given as the provider (the reference's 'Adapter' prompt learner) must be tracked by the
    cache, registered as `prompt_adapter.*`, re-run after an optimizer step, survive a second backward (per-bag backward /
    gradient accumulation) and follow train()/eval() (dropout of the 'FC' variant)."""
    from vlsa_amd.prompt_adapter import PromptAdapter
    from vlsa_amd.vlsa import VLSA
    torch.manual_seed(0)
    pa = PromptAdapter(method="Adapter", num_prompts=4, pretrained_prompt_features=torch.randn(4, 512))
    m = VLSA.from_modules(dict(name="FeatMIL", pooling="mean"), text_provider=pa)
    assert any(k.startswith("prompt_adapter.adapter.fc.") for k in m.state_dict())
    assert not any(k.startswith("text_provider") for k in m.state_dict())
    a = m.forward_text_only()
    a.sum().backward()                          # first backward frees the provider graph ...
    b = m.forward_text_only()                   # ... so this must be a fresh graph, not the cached tensor
    assert b is not a
    b.sum().backward()                          # would raise "backward through the graph a second time" on the stale cache
    with torch.no_grad():
        c0 = m.forward_text_only().clone()
        assert m.forward_text_only() is m.forward_text_only()
        next(pa.adapter.parameters()).mul_(0.5)   # optimizer step
        assert not torch.equal(m.forward_text_only(), c0)
    fc = PromptAdapter(method="FC", num_prompts=4, pretrained_prompt_features=torch.randn(4, 512))
    m2 = VLSA.from_modules(dict(name="FeatMIL", pooling="mean"), text_provider=fc)
    with torch.no_grad():
        m2.train(); t1 = m2.forward_text_only().clone()
        m2.eval(); t2 = m2.forward_text_only().clone()
        assert not torch.equal(t1, t2)           # dropout on / off: the mode is part of the cache key
    # an opaque callable with no declared modules is never cached
    calls = []
    m3 = VLSA.from_modules(dict(name="FeatMIL", pooling="mean"), text_provider=lambda: (calls.append(1), torch.ones(4, 512))[1])
    with torch.no_grad():
        m3.forward_text_only(); m3.forward_text_only()
    assert len(calls) == 2


def test_query_div_loss_cpu():
    import numpy as np
    from vlsa_amd.deepmil import VLFAN
    fx = dict(np.load(os.path.join(GOLDEN, "query_div.npz")))
    for tag, gated in (("plain", False), ("gated", True)):
        enc = VLFAN(dim_in=512, use_feat_proj=False, query="Parameter", num_query=6, gated_query=gated)
        with torch.no_grad():
            enc.Q.copy_(torch.from_numpy(fx[f"{tag}.Q"]))
        assert abs(enc.query_div_loss(last_div=True).item() - float(fx[f"{tag}.loss_last_div"])) < 1e-6
        assert abs(enc.query_div_loss(last_div=False).item() - float(fx[f"{tag}.loss_all"])) < 1e-6


def test_prototype_shap_matches_reference_fixture_cpu():
    import numpy as np
    from vlsa_amd.inference import evaluate_prototype_shap_imp
    fx = dict(np.load(os.path.join(GOLDEN, "interpretation.npz")))
    s = evaluate_prototype_shap_imp(fx["shap_in"], 56.31)
    assert np.abs(s.numpy() - fx["shap_out"]).max() < 1e-5


def test_step_query_is_not_shared_across_bags_when_the_query_network_has_active_dropout():
    """the 'FC' PromptAdapter carries Dropout(0.25) (model/prompt_learners/prompt_adapter.py:95-104); the reference
    evaluates it once per bag (runner/vlsa_handler.py:267-269), so in training mode every bag sees its own mask.  A
    deterministic query network is still evaluated once per parameter version."""
    from vlsa_amd.deepmil import VLFAN

    class QNet(nn.Module):
        def __init__(self, p):
            super().__init__()
            self.base = nn.Parameter(torch.randn(5, 512))
            self.drop = nn.Dropout(p)
            self.calls = 0

        def forward(self):
            self.calls += 1
            return self.drop(self.base)

    enc = VLFAN(dim_in=512, use_feat_proj=False, query="Text", num_query=5)
    q = QNet(0.25)
    enc.reset_query(q)
    enc.train()
    a, b = enc.step_query(), enc.step_query()
    assert q.calls == 2 and not torch.equal(a, b)          # two masks
    enc.eval()
    a, b = enc.step_query(), enc.step_query()
    assert q.calls == 3 and a is b                         # dropout inactive: shared again
    q0 = QNet(0.0)
    enc.reset_query(q0)
    enc.train()
    a, b = enc.step_query(), enc.step_query()
    assert q0.calls == 1 and a is b
