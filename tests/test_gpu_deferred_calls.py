"""``VLSA.defer_training_calls`` (vlsa_amd/deferred.py): the reference handler's bag-by-bag TRAINING loop
(runner/vlsa_handler.py:260-289: ``pred, *_ = self.net(xs[i])`` per bag, ``torch.cat(y_hat)``, loss, ONE backward) served by ONE
``forward_bags`` the moment ``torch.cat`` looks at a prediction.  Same recorded reference run as tests/test_gpu_handler_loop.py, the
mechanics (what triggers, what does not, what can never be served stale), and what is not deferred."""
import numpy as np
import pytest
import torch

import cases
import handler_cases as HC
import handler_loop as HL
import helpers as H
from oracle import vlsa_oracle as O
from test_gpu_handler_loop import _build, _bulk, hooks_installed  # noqa: F401

pytestmark = pytest.mark.gpu


def test_deferred_calls_reproduce_the_reference_run(hooks_installed):  # noqa: F811
    from vlsa_amd.deferred import DeferredOutput
    fx = H.load_fixture("handler_loop")
    model, cfg = _build()
    model.defer_training_calls = True
    opt = HL.make_optimizer(model, cfg)
    xs, ys = HC.train_batch()
    xs, ys = [x.cuda() for x in xs], [y.cuda() for y in ys]
    model.train()
    assert isinstance(model(xs[0])[0], DeferredOutput)                  # (this call's batch is simply never used)
    lr = cfg["opt_lr"]
    for step in range(HC.STEPS):
        loss, preds = HL.update_network(model, opt, O.vlsa_objective, xs, ys)      # the handler's loop, unchanged
        ref_loss = float(fx[f"loss{step}"][0])
        assert abs(loss - ref_loss) < 3e-4 * max(1.0, abs(ref_loss)), (step, loss, ref_loss)
        assert np.abs(preds.numpy() - fx[f"preds{step}"]).max() < 2e-4, step
        if step in (0, HC.STEPS - 1):
            _bulk(model.prompt_learner.context_embeds.detach().cpu(), fx[f"context@{step}"], f"context@{step}", lr, HC.STEPS)
            _bulk(model.prompt_learner.rank_embeds.detach().cpu(), fx[f"rank@{step}"], f"rank@{step}", lr, HC.STEPS)
            _bulk(model.mil_encoder.Q.residual_features.detach().cpu(), fx[f"resid@{step}"], f"resid@{step}", lr, HC.STEPS)
            _bulk(model.mil_encoder.visual_adapter.bias.detach().cpu(), fx[f"b@{step}"], f"b@{step}", lr, HC.STEPS)
            assert abs(float(model.logit_scale.detach()) - float(fx[f"logit_scale@{step}"])) < 2e-6
    out = HL.test_model(model, HC.eval_loader())                                      # eval: nothing is deferred
    assert np.abs(out["raw_y_hat"].numpy() - fx["eval.raw_y_hat"]).max() < 3e-4


def _net(P=12, K=5, seed=811):
    from test_gpu_bagset import _net as build
    net, params = build(P=P, K=K, seed=seed)
    net.train()
    return net, params


def _bags(n=9, seed=7000):
    sizes = [700, 64, 1, 2798, 333, 4100, 65, 900, 17][:n]
    return [cases.make_bag(s, seed + i, "clustered").to(torch.bfloat16).cuda()[None] for i, s in enumerate(sizes)]


def test_nothing_runs_until_an_output_is_used_and_then_everything_does():
    from vlsa_amd.deferred import DeferredOutput
    bags = _bags()
    G = torch.randn(len(bags), 5, generator=cases.gen(7099)).cuda()
    net, _ = _net()
    net.defer_training_calls = True
    outs = [net(x) for x in bags]
    batch = net._pending_calls
    assert batch is not None and len(batch.bags) == len(bags) and batch.real is None
    for i, (pred, feats, text) in enumerate(outs):
        assert all(isinstance(t, DeferredOutput) for t in (pred, feats, text))
        assert tuple(pred.shape) == (1, 5) and tuple(feats.shape) == (1, 512) and tuple(text.shape) == (5, 512)
        assert pred.dtype == torch.float32 and pred.device == bags[0].device and pred.is_cuda and pred.dim() == 2 and len(pred) == 1
        assert pred.requires_grad and pred.size(1) == 5 and pred.numel() == 5
    assert batch.real is None                                              # metadata only: nothing has run
    preds = torch.cat([o[0] for o in outs], dim=0)                          # the handler's first look at a prediction
    assert batch.real is not None and net._pending_calls is None and type(preds) is torch.Tensor and preds.grad_fn is not None
    (preds * G).sum().backward()
    got = [preds.detach().clone()] + [p.grad.clone() for p in net.parameters() if p.grad is not None]
    assert type(outs[3][1] * 1.0) is torch.Tensor                          # later uses of the same batch: the kept rows
    assert torch.equal((outs[2][0] + 0).detach(), preds[2:3].detach())
    # the same numbers as an explicit forward_bags (it IS that call) ...
    ref_net, _ = _net()
    logits = ref_net.forward_bags(bags)[0]
    (logits * G).sum().backward()
    want = [logits.detach().clone()] + [p.grad.clone() for p in ref_net.parameters() if p.grad is not None]
    assert len(got) == len(want) >= 4
    for a, b in zip(got, want):
        assert torch.equal(a, b)
    # ... and, within rounding, as the bag-by-bag route it stands in for
    slow, _ = _net()
    one = torch.cat([slow(x)[0] for x in bags], dim=0)
    assert type(one) is torch.Tensor and (one - preds).abs().max().item() < 5e-5
    (one * G).sum().backward()
    for a, p in zip(got[1:], [p for p in slow.parameters() if p.grad is not None]):
        assert (a - p.grad).abs().max().item() <= 1e-4 * p.grad.abs().max().item() + 1e-6


def test_an_output_is_never_served_under_other_parameters():
    bags = _bags(4)
    net, _ = _net()
    net.defer_training_calls = True
    opt = torch.optim.SGD([p for p in net.parameters() if p.requires_grad], lr=0.1)
    a = net(bags[0])[0]
    first = net._pending_calls
    (net(bags[1])[0].sum() * 1.0).backward()            # uses the batch (bags 0 and 1): a's value exists now
    assert first.real is not None
    opt.step()
    kept = (a + 0).detach().clone()                       # still the value under the OLD parameters: it was computed before the step
    b = net(bags[0])[0]
    assert net._pending_calls is not first
    assert (kept - (b + 0).detach()).abs().max().item() > 1e-4          # the step moved the model: same bag, new value
    # a call whose batch was never used before the step cannot be answered any more: loud, not stale
    c = net(bags[2])[0]
    pending = net._pending_calls
    opt.zero_grad()
    (net(bags[3])[0] * 0).sum()                             # (materialises c's batch as well) ...
    assert pending.real is not None
    d = net(bags[2])[0]
    with torch.no_grad():
        net.logit_scale.add_(0.25)                          # an in-place change with d still pending
    e = net(bags[3])[0]
    assert net._pending_calls is not d._vlsa_batch        # the later call went into a NEW batch
    with pytest.raises(RuntimeError, match="deferred"):
        d + 0
    assert type(e + 0) is torch.Tensor
    # switching to eval with a pending batch: same
    f = net(bags[0])[0]
    net.eval()
    with pytest.raises(RuntimeError, match="deferred"):
        f.sum()
    net.train()


def test_what_is_not_deferred():
    from vlsa_amd.deferred import DeferredOutput
    bags = _bags(2)
    net, _ = _net()
    assert not isinstance(net(bags[0])[0], DeferredOutput)               # off by default
    net.defer_training_calls = True
    with torch.no_grad():
        assert not isinstance(net(bags[0])[0], DeferredOutput)
    net.eval()
    assert not isinstance(net(bags[0])[0], DeferredOutput)
    net.train()
    assert isinstance(net(bags[0])[0], DeferredOutput)
    net._pending_calls = None
    net.mil_encoder.visual_adapter = torch.nn.Sequential(torch.nn.Dropout(0.1), net.mil_encoder.visual_adapter)   # an active dropout layer
    assert not isinstance(net(bags[0])[0], DeferredOutput)


def test_an_epoch_over_resident_bags_with_the_handlers_collector_loop(hooks_installed):  # noqa: F811
    """``_train_each_epoch`` (runner/vlsa_handler.py:192-236): a shuffled loader over ``ResidentBags`` items, ``data_x[0].cuda()`` collected
    into mini-batches of ``bp_every_batch`` bags, ``_update_network`` per mini-batch.  With deferred calls every mini-batch is ONE batched
    forward (the tagged resident views go in as they are); the trajectory equals the bag-by-bag one within rounding."""
    from test_gpu_handler_loop import _PatchItems
    from vlsa_amd.ingest import ResidentBags
    sizes = [700, 64, 1, 2798, 333, 4100, 65, 900, 17, 1200, 300, 2047]
    ds = _PatchItems(sizes)
    K = None
    runs = {}
    for deferred in (False, True):
        model, cfg = _build()
        model.defer_training_calls = deferred
        model.train()
        opt = HL.make_optimizer(model, cfg)
        rb = ResidentBags(ds, dtype=torch.float32)
        g = torch.Generator().manual_seed(5)
        loader = torch.utils.data.DataLoader(rb, batch_size=1, shuffle=True, generator=g, num_workers=0)
        calls = []
        orig = model.forward_bags
        model.forward_bags = lambda bags, **kw: (calls.append(len(bags)), orig(bags, **kw))[1]
        losses, xs, ys = [], [], []
        for epoch in range(2):
            for i_batch, (_idx, data_x, data_y) in enumerate(loader, 1):
                xs.append(data_x[0].cuda())
                K = K or int(model.forward_text_only().shape[0])
                ys.append(torch.stack([data_y[:, 0] % K, data_y[:, 1]], dim=1).cuda())
                if i_batch % 4 == 0 or i_batch == len(loader):
                    loss, _ = HL.update_network(model, opt, O.vlsa_objective, xs, ys)
                    losses.append(loss)
                    xs, ys = [], []
        runs[deferred] = (losses, calls, model.prompt_learner.context_embeds.detach().cpu().clone())
    assert runs[False][1] == [] and runs[True][1] == [4] * 6                  # six mini-batches of four bags: six batched calls
    for a, b in zip(runs[False][0], runs[True][0]):
        assert abs(a - b) < 5e-5 * max(1.0, abs(a)), (a, b)
    _bulk(runs[True][2], runs[False][2], "context after two epochs", cfg["opt_lr"], 6)
