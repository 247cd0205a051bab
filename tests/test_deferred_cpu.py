"""The tensor-subclass mechanics of vlsa_amd/deferred.py without a GPU: a stub model whose ``forward_bags`` is a small differentiable
function on the CPU.  What triggers the batched call and what does not, ``torch.cat`` of consecutive rows returning the batched
tensor itself, gradients, stale outputs raising.  (The real model: tests/test_gpu_deferred_calls.py.)"""
import pytest
import torch

from vlsa_amd.deferred import DeferredOutput, TrainingCalls


class _Stub:
    def __init__(self, K=3, D=4):
        self.w = torch.nn.Parameter(torch.arange(1.0, K * D + 1).reshape(K, D) / 10)
        self._pending_calls, self._materialising, self.calls, self.flag = None, False, 0, 0

    def _defer_key(self):
        return (self.w._version, self.flag)

    def forward_bags(self, bags):
        self.calls += 1
        feats = torch.stack([x.reshape(-1, x.shape[-1]).mean(dim=0) for x in bags])            # [B, D]
        return feats @ self.w.t(), feats, self.w

    def __call__(self, X):
        pc = self._pending_calls
        if pc is None or not pc.same_state():
            pc = self._pending_calls = TrainingCalls(self, self._defer_key(), [self.w], self.w.shape[0], self.w.shape[1], X.device)
        return pc.add(X)


def _bags(n=5):
    g = torch.Generator().manual_seed(3)
    return [torch.randn(1, 6 + i, 4, generator=g) for i in range(n)]


def test_cat_of_the_rows_is_the_batched_tensor_and_gradients_flow():
    m, bags = _Stub(), _bags()
    outs = [m(x) for x in bags]
    assert m.calls == 0 and all(isinstance(o[0], DeferredOutput) for o in outs)
    p = outs[2][0]
    assert tuple(p.shape) == (1, 3) and p.dtype == torch.float32 and p.device.type == "cpu" and len(p) == 1 and p.dim() == 2
    assert p.requires_grad and p.size(-1) == 3 and tuple(outs[0][1].shape) == (1, 4) and tuple(outs[0][2].shape) == (3, 4)
    assert m.calls == 0                                                     # metadata only
    preds = torch.cat([o[0] for o in outs], dim=0)                           # runs ONE forward_bags over all five bags
    assert m.calls == 1 and m._pending_calls is None and type(preds) is torch.Tensor and tuple(preds.shape) == (5, 3)
    batch = p._vlsa_batch
    # ONE node over the batched tensor (not five slices glued together), but fresh storage as torch.cat's always is: an in-place
    # op on the handler's `bag_preds` must not reach what the batch's other outputs read later
    assert torch.equal(preds, batch.real[0]) and preds.data_ptr() != batch.real[0].data_ptr()
    assert preds.grad_fn is not None and preds.grad_fn.next_functions[0][0] is batch.real[0].grad_fn
    part = torch.cat([outs[1][0], outs[2][0], outs[4][0]], dim=0)            # rows 1-2 as one slice + row 4
    assert torch.equal(part, preds[[1, 2, 4]])
    mixed = torch.cat([outs[0][0], torch.ones(2, 3), outs[1][0]], dim=0)
    assert tuple(mixed.shape) == (4, 3) and torch.equal(mixed[3], preds[1])
    preds.sum().backward()
    ref = _Stub()
    ref.forward_bags(bags)[0].sum().backward()
    assert torch.equal(m.w.grad, ref.w.grad)
    # other uses of the same batch: plain tensors, no second evaluation
    assert type(outs[3][0] * 2) is torch.Tensor and type(outs[3][1].detach()) is torch.Tensor and m.calls == 1
    assert float(outs[0][0][0, 1]) == float(preds[0, 1]) and "tensor" in repr(outs[0][0])
    assert torch.equal(outs[0][2] + 0, m.w)


def test_cat_result_does_not_alias_the_batch_and_out_kwarg_and_is_leaf():
    m, bags = _Stub(), _bags(3)
    outs = [m(x) for x in bags]
    preds = torch.cat([o[0] for o in outs], dim=0)
    keep = (outs[1][0] + 0).detach().clone()
    with torch.no_grad():
        preds.clamp_(min=100.0)                                              # in place on the handler's tensor
    assert torch.equal((outs[1][0] + 0).detach(), keep)                      # ... the batch's rows are untouched
    m2 = _Stub()
    o2 = [m2(x) for x in bags]
    buf = torch.empty(3, 3)
    with torch.no_grad():
        r = torch.cat([o[0] for o in o2], dim=0, out=buf)                    # an `out=` tensor with several elements: the generic route
    assert r is buf and torch.equal(buf, m2.forward_bags(bags)[0].detach())
    m3 = _Stub()
    a = m3(bags[0])[0]
    assert m3.calls == 0
    assert a.is_leaf is False and m3.calls == 1                              # the real row is a slice of the batched result: asking materialises


def test_any_operation_triggers_and_a_changed_model_raises():
    m, bags = _Stub(), _bags(3)
    a = m(bags[0])[0]
    assert m.calls == 0
    assert float(a.sum()) == float(m.forward_bags([bags[0]])[0].sum())       # a reduction is a use (calls: 1 + this line's 1)
    assert m.calls == 2
    b = m(bags[1])[0]                                                        # a new batch (the old one is materialised)
    assert b._vlsa_batch is not a._vlsa_batch
    with torch.no_grad():
        m.w.add_(1.0)                                                         # an "optimizer step" with b never looked at
    c = m(bags[2])[0]
    assert c._vlsa_batch is not b._vlsa_batch                                 # the later call went into a batch of its own
    with pytest.raises(RuntimeError, match="deferred"):
        b + 0
    with pytest.raises(RuntimeError, match="deferred"):                       # ... and stays unusable
        torch.cat([b], dim=0)
    assert tuple((c + 0).shape) == (1, 3)
    d = m(bags[0])[0]
    m.flag = 1                                                                # a state change no trainable tensor shows (train / eval flag)
    with pytest.raises(RuntimeError, match="deferred"):
        d.exp()


def test_materialisation_runs_with_gradients_enabled():
    m, bags = _Stub(), _bags(2)
    outs = [m(x)[0] for x in bags]
    with torch.no_grad():
        v = torch.cat(outs, dim=0)          # looked at under no_grad (a logging call): the batch is still built differentiably
    assert not v.requires_grad or v.grad_fn is not None
    (outs[0] * 1.0).sum().backward()
    assert m.w.grad is not None and m.w.grad.abs().sum() > 0


import _ref_import  # noqa: E402


@pytest.mark.skipif(not _ref_import.reference_available(), reason="needs the reference checkout (build container only)")
def test_the_reference_handlers_own_update_network_runs_on_deferred_outputs():
    """``VLSAHandler._update_network`` itself (runner/vlsa_handler.py:260-289, unbound, on a stand-in handler object): its
    ``pred, *_ = self.net(xs[i])`` loop records, its ``torch.cat(y_hat, dim=0)`` runs the ONE batched call, its backward and optimizer
    step see the batched graph."""
    import os
    import types
    cwd = os.getcwd()
    _ref_import.import_reference()
    try:
        os.chdir(_ref_import.REF_ROOT)
        from runner.vlsa_handler import VLSAHandler
    finally:
        os.chdir(cwd)
    m, bags = _Stub(), _bags(4)
    ys = [torch.tensor([[float(i), 1.0]]) for i in range(4)]
    opt = torch.optim.SGD([m.w], lr=0.1)
    seen = {}

    def calc_objective_loss(raw_pred, label):
        seen["pred"], seen["label"] = raw_pred, label
        return (raw_pred * raw_pred).mean()
    handler = types.SimpleNamespace(net=m, optimizer=opt, calc_objective_loss=calc_objective_loss)
    w0 = m.w.detach().clone()
    val_loss, val_preds = VLSAHandler._update_network(handler, bags, ys)
    assert m.calls == 1                                                     # four net(x) calls, ONE evaluation
    assert type(seen["pred"]) is torch.Tensor and tuple(seen["pred"].shape) == (4, 3) and tuple(seen["label"].shape) == (4, 2)
    ref = _Stub()
    logits = ref.forward_bags(bags)[0]
    loss = (logits * logits).mean()
    loss.backward()
    assert abs(val_loss - float(loss)) < 1e-7 and torch.equal(val_preds, logits.detach())
    assert torch.allclose(m.w.detach(), w0 - 0.1 * ref.w.grad)               # the step the batched graph gives
