"""Loss tail of the training step on the device: ``SurvIFMLE`` (loss/loss_surv.py:127-169) and ``SurvEMD``
(loss/loss_surv_ext.py:58-109) with the reference's constructor arguments and call signatures, plus ``SurvObjective`` =
the handler's ``calc_objective_loss`` for ``loss_type: SurvIFMLE-SurvEMD`` (runner/vlsa_handler.py:241-258) taking the RAW
bag logits (softmax converter fused in).  Forward value and gradient come from ONE kernel launch (``vlsa_surv_loss``);
autograd only scales the saved gradient by the incoming one.
"""
from __future__ import annotations


import torch
import torch.nn as nn

from . import _native as nat
from .functional import _need_gpu, _p, _stream


def _launch(x, t, e, from_logits, ls, alpha, eps, p, raw, w_ifmle, w_emd, want_grad):
    _need_gpu(x)
    lib = nat.load()
    x = x.detach().float().contiguous()
    B, K = x.shape
    t = t.reshape(-1).to(device=x.device, dtype=torch.int64).contiguous()
    e = e.reshape(-1).to(device=x.device, dtype=torch.float32).contiguous()
    if t.numel() != B or e.numel() != B:
        raise ValueError("t and e must hold one entry per sample")
    if ls is not None and not isinstance(ls, torch.Tensor):
        ls = torch.tensor(float(ls), device=x.device)
    if ls is not None:
        ls = ls.detach().to(device=x.device, dtype=torch.float32).reshape(1).contiguous()
    o1 = torch.empty(B, dtype=torch.float32, device=x.device) if w_ifmle != 0 else None
    o2 = torch.empty(B, dtype=torch.float32, device=x.device) if w_emd != 0 else None
    g = torch.empty(B, K, dtype=torch.float32, device=x.device) if want_grad else None
    nat.check(lib.vlsa_surv_loss(_p(x), _p(t), _p(e), B, K, int(from_logits), _p(ls), float(alpha), float(eps), int(p),
                                 int(raw), float(w_ifmle), float(w_emd), _p(o1), _p(o2), _p(g), _stream()), "vlsa_surv_loss")
    return o1, o2, g


class _SurvLossFn(torch.autograd.Function):
    """per-sample loss [B] = w_ifmle * ifmle_i + w_emd * emd_i, differentiable w.r.t. the predictions only."""

    @staticmethod
    def forward(ctx, x, t, e, ls, from_logits, alpha, eps, p, raw, w_ifmle, w_emd):
        o1, o2, g = _launch(x, t, e, from_logits, ls, alpha, eps, p, raw, w_ifmle, w_emd, want_grad=True)
        ctx.save_for_backward(g)
        if o1 is None or o2 is None:
            return w_emd * o2 if o1 is None else w_ifmle * o1
        return w_ifmle * o1 + w_emd * o2

    @staticmethod
    def backward(ctx, dl):
        (g,) = ctx.saved_tensors
        return (dl.reshape(-1, 1) * g,) + (None,) * 10


class _SurvObjectiveFn(torch.autograd.Function):
    """mean_i (w_ifmle ifmle_i + w_emd emd_i) from the raw logits: ONE launch forward (``vlsa_surv_objective``: value, gradient and the
    mean over the batch; optionally the exp of the raw logit scale), one multiplication backward."""

    @staticmethod
    def forward(ctx, x, t, e, ls, ls_is_log, alpha, eps, p, raw, w_ifmle, w_emd):
        _need_gpu(x)
        lib = nat.load()
        x = x.detach().float().contiguous()
        B, K = x.shape
        t = t.reshape(-1).to(device=x.device, dtype=torch.int64).contiguous()
        e = e.reshape(-1).to(device=x.device, dtype=torch.float32).contiguous()
        if t.numel() != B or e.numel() != B:
            raise ValueError("t and e must hold one entry per sample")
        ls = ls.detach().to(device=x.device, dtype=torch.float32).reshape(1).contiguous()
        out = torch.empty(1 + B * K, dtype=torch.float32, device=x.device)      # objective | gradient: one allocation
        g = out[1:].view(B, K)
        nat.check(lib.vlsa_surv_objective(_p(x), _p(t), _p(e), B, K, 1, _p(ls), int(ls_is_log), float(alpha), float(eps), int(p), int(raw),
                                          float(w_ifmle), float(w_emd), _p(out), _p(g), _stream()), "vlsa_surv_objective")
        ctx.save_for_backward(g)
        return out[0]

    @staticmethod
    def backward(ctx, dl):
        (g,) = ctx.saved_tensors
        return (dl * g,) + (None,) * 10


def _reduce(loss, reduction):
    if reduction == "mean":
        return loss.mean()
    if reduction == "sum":
        return loss.sum()
    return loss.reshape(-1, 1)  # the reference's 'none' keeps the [B, 1] column


class SurvIFMLE(nn.Module):
    def __init__(self, alpha=0.0, eps=1e-7, reduction="mean", **kws):
        super().__init__()
        assert reduction in ["sum", "mean", "none"]
        self.alpha, self.eps, self.reduction = alpha, eps, reduction

    def forward(self, incidence_hat, t, e, cur_alpha=None):
        alpha = self.alpha if cur_alpha is None else cur_alpha
        loss = _SurvLossFn.apply(incidence_hat, t, e, None, False, alpha, self.eps, 2, True, 1.0, 0.0)
        return _reduce(loss, self.reduction)


class SurvEMD(nn.Module):
    def __init__(self, p=2, raw_distance=True, reduction="mean", **kws):
        super().__init__()
        assert reduction in ["mean", "sum", "none"]
        if p not in (1, 2):
            raise NotImplementedError("the fused kernel covers p = 1 and p = 2 (cfg_vlsa_conch.yaml uses 2)")
        self.p, self.raw_distance, self.reduction = p, raw_distance, reduction

    def forward(self, y_hat, t, e, cur_logit_scale=10.0):
        loss = _SurvLossFn.apply(y_hat, t, e, cur_logit_scale, False, 0.0, 1e-7, self.p, self.raw_distance, 0.0, 1.0)
        if self.reduction == "mean":
            return loss.mean()
        if self.reduction == "sum":
            return loss.sum()
        return loss


class SurvObjective(nn.Module):
    """weight_ifmle * SurvIFMLE(softmax(raw)) + weight_emd * SurvEMD(softmax(raw)), both 'mean'-reduced, from the raw
    [B, K] logits in one launch (``label``: [B, 2] = (t, e) as in the handler, or t and e separately)."""

    def __init__(self, weight_ifmle=1.0, weight_emd=1.0, alpha=0.0, eps=1e-7, p=2, raw_distance=True):
        super().__init__()
        if p not in (1, 2):
            raise NotImplementedError("the fused kernel covers p = 1 and p = 2")
        self.w1, self.w2, self.alpha, self.eps, self.p, self.raw = weight_ifmle, weight_emd, alpha, eps, p, raw_distance

    def forward(self, raw_pred, t, e=None, cur_logit_scale=10.0, log_logit_scale=None):
        """cur_logit_scale: exp(logit_scale) as the handler passes it (``net.get_logit_scale()``); log_logit_scale: the raw parameter
        instead (exponentiated inside the launch: one kernel less per step; no gradient flows into it either way -- the reference detaches it)."""
        if e is None:
            t, e = t[:, 0], t[:, 1]
        ls, is_log = (log_logit_scale, True) if log_logit_scale is not None else (cur_logit_scale, False)
        if isinstance(ls, torch.Tensor) and raw_pred.is_cuda and raw_pred.dim() == 2 and raw_pred.shape[0] <= 4096:
            return _SurvObjectiveFn.apply(raw_pred, t, e, ls, is_log, self.alpha, self.eps, self.p, self.raw, self.w1, self.w2)
        if is_log:
            ls = ls.detach().exp() if isinstance(ls, torch.Tensor) else float(torch.tensor(ls).exp())
        loss = _SurvLossFn.apply(raw_pred, t, e, ls, True, self.alpha, self.eps, self.p, self.raw, self.w1, self.w2)
        return loss.mean()

    def value_and_grad(self, raw_pred, t, e=None, log_logit_scale=None):
        """(objective, d objective / d raw_pred) from the one launch, outside autograd -- for a caller that owns the step and starts the
        backward pass itself with ``raw_pred.backward(grad)`` (``vlsa_amd.train_step.TrainStep``): ``loss.backward()`` costs two more
        launches (the ones_like seed and its product with the saved gradient), ~10 us of a graph-replayed step.  [B <= 4096, K] logits
        on the GPU, the raw (log) logit scale as a tensor."""
        if e is None:
            t, e = t[:, 0], t[:, 1]
        lib = nat.load()
        x = raw_pred.detach().float().contiguous()
        _need_gpu(x)
        B, K = x.shape
        t = t.reshape(-1).to(device=x.device, dtype=torch.int64).contiguous()
        e = e.reshape(-1).to(device=x.device, dtype=torch.float32).contiguous()
        if t.numel() != B or e.numel() != B:
            raise ValueError("t and e must hold one entry per sample")
        ls = log_logit_scale.detach().to(device=x.device, dtype=torch.float32).reshape(1).contiguous()
        out = torch.empty(1 + B * K, dtype=torch.float32, device=x.device)
        g = out[1:].view(B, K)
        nat.check(lib.vlsa_surv_objective(_p(x), _p(t), _p(e), B, K, 1, _p(ls), 1, float(self.alpha), float(self.eps), int(self.p),
                                          int(self.raw), float(self.w1), float(self.w2), _p(out), _p(g), _stream()), "vlsa_surv_objective")
        return out[0], g
