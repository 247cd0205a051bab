"""The text-feature cache of the drop-in model (vlsa_amd/vlsa.py: forward_text_only / _provider_key), on the CPU with stub
modules: the reference's handler calls the model once per bag, so the key is computed per bag -- it must be cheap AND follow
everything the text features depend on."""
import torch
import torch.nn as nn

from vlsa_amd.vlsa import VLSA


class Learner(nn.Module):
    def __init__(self):
        super().__init__()
        self.context_embeds = nn.Parameter(torch.randn(3, 5, 8))
        self.register_buffer("pseudo_sentence_tokens", torch.ones(3, 5, dtype=torch.long), persistent=False)

    def forward(self):
        return self.context_embeds


class Tower(nn.Module):
    """stands in for the 110-module text tower: a few nested submodules, a dropout (train / eval matters), a call counter"""

    def __init__(self):
        super().__init__()
        self.blocks = nn.ModuleList([nn.Sequential(nn.Linear(8, 8), nn.Dropout(0.5)) for _ in range(3)])
        self.proj = nn.Linear(8, 512)
        self.calls = 0

    def forward(self, prompts_embedding, prompts_pseudo_tokens, **kw):
        self.calls += 1
        x = prompts_embedding
        for b in self.blocks:
            x = x + b(x)
        return self.proj(x.mean(dim=1))


def _model():
    torch.manual_seed(0)
    tower, learner = Tower(), Learner()
    model = VLSA.from_modules(dict(name="FeatMIL", pooling="mean"), prompt_learner=learner, prompt_encoder=tower)
    return model.eval(), tower, learner


def test_text_features_are_computed_once_per_parameter_version():
    model, tower, learner = _model()
    with torch.no_grad():
        a = model.forward_text_only()
        for _ in range(5):
            b = model.forward_text_only()
        assert tower.calls == 1 and b is a
        assert model._tower_lists is not None and model._tower_lists[0] is tower
        n_tensors = len(model._tower_lists[2])
        assert n_tensors == sum(1 for _ in tower.parameters()) + sum(1 for _ in tower.buffers())
        # in-place edit of a tower weight, of a learner parameter: each is a new version
        tower.blocks[1][0].weight.mul_(1.5)
        c = model.forward_text_only()
        assert tower.calls == 2 and not torch.equal(a, c)
        learner.context_embeds.add_(0.25)
        d = model.forward_text_only()
        assert tower.calls == 3 and not torch.equal(c, d)
        assert model.forward_text_only() is d and tower.calls == 3


def test_train_eval_flag_of_any_submodule_and_grad_mode_are_part_of_the_key():
    model, tower, learner = _model()
    with torch.no_grad():
        model.forward_text_only()
        tower.blocks[2][1].train()                 # one dropout deep inside the tower
        model.forward_text_only()
        assert tower.calls == 2
        tower.blocks[2][1].eval()
        model.forward_text_only()
        assert tower.calls == 3
    f = model.forward_text_only()                  # grad mode on: a graph is needed
    assert tower.calls == 4 and f.requires_grad
    g = model.forward_text_only()
    assert g is f and tower.calls == 4             # the bags of a step share the features WITH their graph ...
    f.sum().backward()
    h = model.forward_text_only()                  # ... which a backward pass frees: rebuilt
    assert tower.calls == 5 and h is not f


def test_load_state_dict_and_conversion_drop_the_cache():
    model, tower, learner = _model()
    with torch.no_grad():
        a = model.forward_text_only().clone()
        sd = {k: v.clone() for k, v in model.state_dict().items()}
        sd["prompt_learner.context_embeds"] += 1.0
        model.load_state_dict(sd)
        b = model.forward_text_only()
        assert tower.calls == 2 and not torch.equal(a, b)
        model.double()                             # _apply: same Parameter objects, same versions, different tensors
        c = model.forward_text_only()
        assert tower.calls == 3 and c.dtype == torch.float64


def test_a_parameter_object_swapped_inside_the_tower_is_seen_at_the_next_miss():
    model, tower, learner = _model()
    with torch.no_grad():
        model.forward_text_only()
        old = tower.proj.weight
        tower.proj.weight = nn.Parameter(old.detach() * 2.0)
        learner.context_embeds.add_(0.0)           # any tracked change -> a miss -> the tower is walked again
        b = model.forward_text_only()
        assert tower.calls == 2
        assert any(t is tower.proj.weight for t in model._tower_lists[2]) and not any(t is old for t in model._tower_lists[2])
        tower.proj.weight.mul_(0.5)                # ... and the new object is tracked from then on
        c = model.forward_text_only()
        assert tower.calls == 3 and not torch.equal(b, c)


def test_opaque_provider_is_never_cached():
    calls = []

    def provider():
        calls.append(1)
        return torch.randn(4, 512)
    model = VLSA.from_modules(dict(name="FeatMIL", pooling="mean"), text_provider=provider).eval()
    with torch.no_grad():
        model.forward_text_only()
        model.forward_text_only()
    assert len(calls) == 2


def test_deepcopy_and_pickle_drop_the_native_caches():
    """copy.deepcopy / pickle of the modules go through TransientCaches.__getstate__: plans, ctypes views of the weights and cached
    results are not state (and ctypes structures with pointers cannot be pickled at all)."""
    import copy
    import ctypes
    import io
    from vlsa_amd.prompt_encoder import CONCHPromptEncoder
    model, tower, learner = _model()
    with torch.no_grad():
        a = model.forward_text_only().clone()
    model._plans["x"] = ctypes.pointer(ctypes.c_int(3))            # what a used model holds: native handles
    model._train_plans["y"] = ctypes.pointer(ctypes.c_int(4))
    twin = copy.deepcopy(model)
    assert twin._plans == {} and twin._train_plans == {} and twin._text_cache is None and twin._tower_lists is None
    with torch.no_grad():
        assert torch.equal(twin.forward_text_only(), a)            # same parameters, caches rebuilt on use
    assert twin.prompt_encoder is not tower and twin._tower_lists[0] is twin.prompt_encoder
    buf = io.BytesIO()
    torch.save(model, buf)
    buf.seek(0)
    back = torch.load(buf, weights_only=False)
    with torch.no_grad():
        assert torch.equal(back.forward_text_only(), a)
    enc = CONCHPromptEncoder(width=128, heads=2, layers=1, vocab_size=50, output_dim=64)
    enc._cm, enc._cm_arr = ctypes.pointer(ctypes.c_int(1)), ctypes.pointer(ctypes.c_int(2))
    enc.__dict__["_tt_cache"] = (1, [enc.cls_emb])
    e2 = copy.deepcopy(enc)
    assert e2._cm is None and "_cm_arr" not in e2.__dict__ and "_tt_cache" not in e2.__dict__
    assert set(e2.state_dict()) == set(enc.state_dict())


def test_unpickled_model_in_a_fresh_process_still_sees_a_reassigned_parameter(tmp_path):
    """a VLSA unpickled in a fresh process never runs ``_assemble``; the structure hooks are then installed lazily by the
    first key, so a re-assigned learner parameter changes the key (and the text features are recomputed)."""
    import os
    import subprocess
    import sys
    model, tower, learner = _model()
    path = tmp_path / "model.pt"
    torch.save(model, path)
    here = os.path.dirname(os.path.abspath(__file__))
    child = f"""
import sys
sys.path.insert(0, {os.path.dirname(here)!r}); sys.path.insert(0, {here!r})
import torch, torch.nn as nn
import test_text_cache_cpu as T          # the pickled classes live here
sys.modules.setdefault("tests.test_text_cache_cpu", T)
from vlsa_amd import vlsa as V
assert not V._HOOKS_INSTALLED[0]
m = torch.load({str(path)!r}, weights_only=False)
assert not V._HOOKS_INSTALLED[0]          # unpickling assembled nothing
with torch.no_grad():
    a = m.forward_text_only().clone()
    k0 = m._provider_key()
    assert V._HOOKS_INSTALLED[0]
    calls = m.prompt_encoder.calls
    m.forward_text_only()
    assert m.prompt_encoder.calls == calls
    m.prompt_learner.context_embeds = nn.Parameter(m.prompt_learner.context_embeds.detach() + 1.0)
    k1 = m._provider_key()
    assert k1 != k0
    b = m.forward_text_only()
    assert m.prompt_encoder.calls == calls + 1 and not torch.equal(a, b)
print("ok")
"""
    env = dict(os.environ)
    res = subprocess.run([sys.executable, "-c", child], capture_output=True, text=True, env=env, timeout=300)
    assert res.returncode == 0 and "ok" in res.stdout, res.stdout + res.stderr
