"""The safety net around the stateful shortcuts of the boundary: the environment-level kill switches
(``VLSA_AMD_NO_DEFER`` / ``VLSA_AMD_NO_LOOKAHEAD``: read once at import, patched on the module here), ``VLSA_AMD_PARANOID`` cross-checks of
deferred batches and look-ahead windows against the per-bag route, the identity-FeatMIL encoder under deferral (its second output is
the bag's [N, D] patch features, model/vlsa.py:188-196: such calls must run as they come), a re-assigned encoder parameter between two
hits of one look-ahead window, and the in-place guard of the resident arena."""
import pytest
import torch

import cases
from test_gpu_bagset import _net as _vlfan_net

pytestmark = pytest.mark.gpu


def _bags(n=9, seed=7300, dtype=torch.bfloat16):
    sizes = [700, 64, 1, 2798, 333, 4100, 65, 900, 17][:n]
    return [cases.make_bag(s, seed + i, "clustered").to(dtype).cuda()[None] for i, s in enumerate(sizes)]


class _Items(torch.utils.data.Dataset):
    def __init__(self, sizes, seed=41):
        self.feats = [cases.make_bag(n, seed + i, "clustered") for i, n in enumerate(sizes)]

    def __len__(self):
        return len(self.feats)

    def __getitem__(self, i):
        return torch.tensor([i], dtype=torch.int), (self.feats[i], torch.zeros(1)), torch.ones(2)


def _item(rb, i):
    return torch.utils.data.default_collate([rb[i]])[1][0].cuda()


def _resident(sizes, **kw):
    from vlsa_amd.ingest import ResidentBags
    ds = _Items(sizes, **kw)
    rb = ResidentBags(ds, dtype=torch.float32)
    for i in range(len(sizes)):
        rb[i]                               # upload: look-ahead windows only span items that are resident already
    torch.cuda.synchronize()
    return ds, rb


def test_no_defer_switch_wins_over_the_model_flag(monkeypatch):
    from vlsa_amd import vlsa as V
    from vlsa_amd.deferred import DeferredOutput
    net, _ = _vlfan_net()
    net.train()
    net.defer_training_calls = True
    x = _bags(1)[0]
    assert isinstance(net(x)[0], DeferredOutput)
    net._pending_calls = None
    monkeypatch.setattr(V, "ENV_NO_DEFER", True)
    out = net(x)[0]
    assert type(out) is torch.Tensor and out.grad_fn is not None and net._pending_calls is None


def test_no_lookahead_switch(monkeypatch):
    from vlsa_amd import vlsa as V
    from vlsa_amd.ingest import ResidentBags
    net, _ = _vlfan_net()
    net.eval()
    _, rb = _resident([500, 300, 700, 64])
    with torch.no_grad():
        a = [net(_item(rb, i))[0].clone() for i in range(4)]
        assert net._la is not None and len(net._la["rows"]) == 4
        net._la = None
        monkeypatch.setattr(V, "ENV_NO_LOOKAHEAD", True)
        b = [net(_item(rb, i))[0].clone() for i in range(4)]
        assert net._la is None                                     # every call ran its own bag
    for x, y in zip(a, b):
        assert (x - y).abs().max().item() < 1e-4


def test_paranoid_checks_run_and_catch_a_wrong_batch(monkeypatch):
    from vlsa_amd import vlsa as V
    from vlsa_amd.ingest import ResidentBags
    monkeypatch.setattr(V, "ENV_PARANOID", True)
    # deferred training batch
    net, _ = _vlfan_net()
    net.train()
    net.defer_training_calls = True
    bags = _bags(6)
    outs = [net(x)[0] for x in bags]
    preds = torch.cat(outs, dim=0)
    assert net._paranoid_checks == 1 and preds.grad_fn is not None
    preds.sum().backward()                                         # the check left the batched graph intact
    assert net.mil_encoder.Q.grad is not None
    # look-ahead windows
    net.eval()
    _, rb = _resident([500, 300, 700, 64, 900, 129])
    with torch.no_grad():
        for i in range(6):
            net(_item(rb, i))
    assert net._paranoid_checks >= 2
    # a shortcut that returns wrong numbers is caught
    orig = net._forward_bags_fused

    def wrong(*a, **k):
        out = orig(*a, **k)
        return (out[0] + 1e-3,) + tuple(out[1:])
    monkeypatch.setattr(net, "_forward_bags_fused", wrong)
    net._la = None
    with torch.no_grad(), pytest.raises(V.ParanoidMismatch, match="look-ahead"):
        net(_item(rb, 0))
    monkeypatch.setattr(net, "_forward_bags_fused", orig)
    net.train()
    orig_fb = net.forward_bags

    def wrong_fb(b, **k):
        out = orig_fb(b, **k)
        return (out[0] + 1e-3,) + tuple(out[1:])
    monkeypatch.setattr(net, "forward_bags", wrong_fb)
    o = net(bags[0])[0]
    with pytest.raises(V.ParanoidMismatch, match="deferred"):
        o + 0


class _Prompts(torch.nn.Module):
    """a trainable text side: K raw text features as a parameter (what a CoOp learner + tower hand the model)"""

    def __init__(self, K, seed):
        super().__init__()
        self.t = torch.nn.Parameter(torch.randn(K, 512, generator=cases.gen(seed)))

    def forward(self):
        return self.t * 1.0


@pytest.mark.parametrize("pooling", ["logit_top10", "mean"])
def test_featmil_with_trainable_prompts_under_deferral_equals_the_per_bag_calls(pooling):
    """identity FeatMIL ('logit_*' pooling): image_features is the bag's [N, D] unit patch features -- never a deferred row;
    mean FeatMIL: one row per bag, deferred.  All three outputs and the prompt gradient equal the bag-by-bag run."""
    from vlsa_amd.deferred import DeferredOutput
    from vlsa_amd.vlsa import VLSA
    bags = _bags(5, dtype=torch.float32)
    G = torch.randn(len(bags), 4, generator=cases.gen(5)).cuda()
    res = {}
    for deferred in (False, True):
        net = VLSA.from_modules(dict(name="FeatMIL", dim_in=512, pooling=pooling), text_provider=_Prompts(4, 77)).cuda()
        net.train()
        net.defer_training_calls = deferred
        outs = [net(x) for x in bags]
        if deferred:
            assert isinstance(outs[0][0], DeferredOutput) == (pooling == "mean")
        preds = torch.cat([o[0] for o in outs], dim=0)
        (preds * G).sum().backward()
        res[deferred] = (preds.detach().clone(), [(o[1] + 0).detach().clone() for o in outs], (outs[0][2] + 0).detach().clone(),
                         net.prompt_adapter.t.grad.clone())
    a, b = res[False], res[True]
    assert (a[0] - b[0]).abs().max().item() < 1e-4
    for i, (fa, fb) in enumerate(zip(a[1], b[1])):
        assert fa.shape == fb.shape, (i, fa.shape, fb.shape)          # [N_i, 512] for the identity encoder, [1, 512] for 'mean'
        if pooling != "mean":
            assert fa.shape[0] == bags[i].shape[1]
        assert (fa - fb).abs().max().item() < 1e-5
    assert (a[2] - b[2]).abs().max().item() < 1e-6
    assert (a[3] - b[3]).abs().max().item() < 1e-4 * max(1.0, a[3].abs().max().item())


def test_lookahead_sees_a_reassigned_encoder_parameter_and_a_changed_scalar():
    from oracle import vlsa_oracle as O
    from vlsa_amd.ingest import ResidentBags
    net, _ = _vlfan_net()
    net.eval()
    ds, rb = _resident([500, 300, 700, 64, 900, 129])
    enc = net.mil_encoder

    def check(i, what):
        with torch.no_grad():
            got = net(_item(rb, i))[0].cpu()
            ref = O.vlsa_vlfan_forward(ds.feats[i], enc.get_query().cpu(), net.pretrained_text_features.cpu(), net.logit_scale.detach().cpu(),
                                       head_weight=enc.visual_adapter.weight.detach().cpu(), head_bias=enc.visual_adapter.bias.detach().cpu(),
                                       query_pooling_method=enc.query_pooling)["logits"]
        assert (got - ref).abs().max().item() < 1e-4, what
    check(0, "first window")
    assert len(net._la["rows"]) == 6
    enc.Q = torch.nn.Parameter(enc.Q.detach().flip(0) * 1.3 + 0.1)          # a NEW parameter object, version 0 like the old one
    check(1, "after re-assigning the query parameter")
    enc.query_pooling = "max"                                                 # a plain attribute: no tensor, no flag shows it
    check(2, "after switching the query pooling")


def test_resident_rows_modified_in_place_are_reported():
    from vlsa_amd.ingest import ResidentBags
    rb = ResidentBags(_Items([300, 200]), dtype=torch.float32)
    x = _item(rb, 0)
    y = x * 2                                                                  # out of place: fine
    assert _item(rb, 1).shape[1] == 200 and y.shape == x.shape
    x.mul_(2.0)                                                                # the handler would be corrupting the resident bag
    with pytest.raises(RuntimeError, match="IN PLACE"):
        rb[0]


def test_in_place_write_before_a_later_lazy_upload_is_still_reported():
    """the first epoch uploads lazily -- item 1 goes into the SAME arena segment after item 0's view was written in place; the
    snapshot of the arena's own version must advance by the upload's bumps only, not absorb the user's write."""
    from vlsa_amd.ingest import ResidentBags
    rb = ResidentBags(_Items([300, 200]), dtype=torch.float32)
    x = _item(rb, 0)
    x.add_(1.0)                                                                # in place, BEFORE item 1 was ever read
    try:
        _item(rb, 1)                                                           # its upload must not launder the write ...
    except RuntimeError as e:                                                  # (same segment: reported right here)
        assert "IN PLACE" in str(e)
    with pytest.raises(RuntimeError, match="IN PLACE"):
        rb[0]                                                                  # ... whichever segment item 1 went to


def _oracle_logits(net, X):
    from oracle import vlsa_oracle as O
    enc = net.mil_encoder
    with torch.no_grad():
        ref = O.vlsa_vlfan_forward(X[0].float().cpu(), enc.get_query().detach().float().cpu(), net._text_features().detach().float().cpu(),
                                   net.logit_scale.detach().cpu(), head_weight=enc.visual_adapter.weight.detach().cpu(),
                                   head_bias=enc.visual_adapter.bias.detach().cpu())
    return ref["logits"][0]


def test_hot_call_equals_the_full_route_and_follows_every_state_change(monkeypatch):
    """Round 6: a repeated per-bag inference call under an unchanged model state is ONE pre-built C call (``VLSA._fused_vlfan`` ->
    ``VlfanInferencePlan.hot_call``).  Same bits as the full route (``VLSA_AMD_NO_HOTCALL``), and nothing stale: an in-place parameter
    update, a re-assigned parameter, new text features, the logit scale, train / eval, another bag size or dtype -- each followed by a
    call that is checked against the CPU oracle on the CURRENT values."""
    from vlsa_amd import vlsa as V
    net, params = _vlfan_net()
    net.eval()
    bags = {(n, dt): cases.make_bag(n, 7700 + n, "clustered").to(dt).cuda()[None] for n in (2798, 700) for dt in (torch.bfloat16, torch.float32)}
    X = bags[(2798, torch.bfloat16)]

    def call(x):
        with torch.no_grad():
            return [t.clone() for t in net(x)]
    first = call(X)                                        # full route, leaves a hot entry
    assert len(net._hot) == 1
    hot = call(X)
    monkeypatch.setattr(V, "ENV_NO_HOTCALL", True)
    full = call(X)
    monkeypatch.setattr(V, "ENV_NO_HOTCALL", False)
    for a, b, c in zip(first, hot, full):
        assert torch.equal(a, b) and torch.equal(a, c)
    assert (hot[0][0].cpu() - _oracle_logits(net, X)).abs().max().item() < 1e-4

    def check(what, x=X):
        got = call(x)[0][0].cpu()
        err = (got - _oracle_logits(net, x)).abs().max().item()
        assert err < 1e-4, (what, err)
        again = call(x)[0][0].cpu()                        # ... and the (possibly hot) repeat says the same
        assert torch.equal(got, again), what
        return got
    base = check("unchanged")
    enc = net.mil_encoder
    with torch.no_grad():
        enc.Q.mul_(1.25).add_(0.01)                        # in place: version bump
    assert not torch.equal(check("query updated in place"), base)
    enc.Q = torch.nn.Parameter((enc.Q.detach() * 0.5).clone())          # re-assigned: a new object with version 0
    check("query re-assigned")
    with torch.no_grad():
        net.logit_scale.add_(0.3)
    check("logit scale")
    with torch.no_grad():
        enc.visual_adapter.weight.mul_(0.9)
        enc.visual_adapter.bias.add_(0.05)
    check("adapter")
    with torch.no_grad():
        net.pretrained_text_features.mul_(-1.0)            # the text features the logits are taken against
    check("text features")
    net.train()
    check("train mode")
    net.eval()
    for key, x in bags.items():                            # other shapes / dtypes: their own plans and entries
        check(f"bag {key}", x)
        check(f"bag {key} again", x)
    assert 1 <= len(net._hot) <= 4
    xs = X[:, ::2]                                         # a strided view of the bag: not what the entry was built for
    assert xs.stride(1) != X.stride(1)
    check("strided bag", xs)
    net2 = __import__("copy").deepcopy(net)
    assert net2._hot == {}                                 # native handles are not state
