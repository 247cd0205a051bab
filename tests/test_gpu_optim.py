"""vlsa_amd.optim.FusedAdam (ONE HIP launch over all parameter tensors, vlsa_amd/csrc/adam.hip) against torch.optim.Adam -- the optimizer
the reference's handler builds (optim_factory.py:25-60: two groups, weight decay only on the >= 2-D parameters; cfg_vlsa_conch.yaml:111-113)
-- on the training step's tensor shapes: same parameters and moments after 25 steps, a learning-rate change on the way, the state dict
both ways, and a captured step replayed."""
import copy

import pytest
import torch

pytestmark = pytest.mark.gpu

SHAPES = [(12, 512), (512, 512), (512,), (8, 768), (4, 4, 768), (), (1031,), (3, 70000)]


def _params(seed):
    g = torch.Generator().manual_seed(seed)
    return [torch.nn.Parameter((torch.randn(*s, generator=g) if s else torch.randn((), generator=g)).cuda()) for s in SHAPES]


def _groups(ps):
    return [{"params": [p for p in ps if p.dim() < 2], "weight_decay": 0.0}, {"params": [p for p in ps if p.dim() >= 2], "weight_decay": 1e-5}]


def _set_grads(ps, step):
    g = torch.Generator().manual_seed(1000 + step)
    for i, p in enumerate(ps):
        gr = torch.randn(p.shape, generator=g).cuda() * (0.1 + i)
        p.grad = gr if (step + i) % 7 else gr * 0.0          # a zero gradient now and then (moments decay, no update direction)


def test_matches_torch_adam_over_25_steps_with_a_learning_rate_change():
    from vlsa_amd.optim import FusedAdam
    a, b = _params(5), _params(5)
    oa, ob = FusedAdam(_groups(a), lr=2e-4), torch.optim.Adam(_groups(b), lr=2e-4)
    for step in range(25):
        if step == 10:
            for o in (oa, ob):
                for gr in o.param_groups:
                    gr["lr"] = 5e-4
        _set_grads(a, step); _set_grads(b, step)
        v0 = [p._version for p in a]
        oa.step(); ob.step()
        assert all(p._version > v for p, v in zip(a, v0))       # the raw-pointer update is visible to version-keyed caches
    errs = {tuple(p.shape): ((p - q).abs().max().item(), q.abs().max().item()) for p, q in zip(a, b)}
    assert int(oa.state[a[0]]["step"]) == 25, (int(oa.state[a[0]]["step"]), errs)
    for p, q in zip(a, b):
        assert (p - q).abs().max().item() <= 2e-6 * max(1.0, q.abs().max().item()), (p.shape, errs)
    for p, q in zip(a, b):
        for k in ("exp_avg", "exp_avg_sq"):
            x, y = oa.state[p][k], ob.state[q][k]
            assert (x - y).abs().max().item() <= 1e-6 * max(1e-6, y.abs().max().item()) + 1e-12, (k, p.shape)
    assert int(oa.state[a[0]]["step"]) == 25


def test_state_dict_round_trips_with_torch_adam():
    from vlsa_amd.optim import FusedAdam
    a, b = _params(6), _params(6)
    oa, ob = FusedAdam(_groups(a), lr=1e-3), torch.optim.Adam(_groups(b), lr=1e-3)
    for step in range(5):
        _set_grads(a, step); _set_grads(b, step)
        oa.step(); ob.step()
    # torch -> fused: continue from a torch.optim.Adam checkpoint
    c = _params(6)
    with torch.no_grad():
        for p, q in zip(c, b):
            p.copy_(q)
    oc = FusedAdam(_groups(c), lr=1e-3)
    oc.load_state_dict(copy.deepcopy(ob.state_dict()))      # (state_dict() hands out the live tensors: a copy, as a checkpoint file would be)
    for step in range(5, 9):
        _set_grads(b, step); _set_grads(c, step)
        ob.step(); oc.step()
    for p, q in zip(c, b):
        assert (p - q).abs().max().item() <= 2e-6 * max(1.0, q.abs().max().item())
    # fused -> fused through a state dict
    d = _params(6)
    with torch.no_grad():
        for p, q in zip(d, a):
            p.copy_(q)
    od = FusedAdam(_groups(d), lr=1e-3)
    od.load_state_dict(copy.deepcopy(oa.state_dict()))
    _set_grads(a, 40); _set_grads(d, 40)
    oa.step(); od.step()
    for p, q in zip(d, a):
        assert torch.equal(p, q)


def test_a_captured_step_replays_and_follows_the_learning_rate():
    from vlsa_amd.optim import FusedAdam
    a, b = _params(7), _params(7)
    oa, ob = FusedAdam(_groups(a), lr=1e-3), FusedAdam(_groups(b), lr=1e-3)
    _set_grads(a, 0); _set_grads(b, 0)
    grads = [p.grad for p in a]                               # static gradient buffers, refilled between replays
    oa.step(); ob.step()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=s):
        oa.step()
    for step in range(1, 6):
        if step == 3:
            for o in (oa, ob):
                for gr in o.param_groups:
                    gr["lr"] = 3e-3
            oa.sync_hyper()
        _set_grads(b, step)
        for buf, p in zip(grads, b):
            buf.copy_(p.grad)
        g.replay()
        ob.step()
    torch.cuda.synchronize()
    for p, q in zip(a, b):
        assert torch.equal(p, q), p.shape
    assert int(oa.state[a[0]]["step"]) == 6
