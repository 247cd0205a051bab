"""``FusedAdam``: torch.optim.Adam's update for the training step's parameters as ONE HIP launch (``vlsa_adam_step``,
vlsa_amd/csrc/adam.hip).

The reference builds ``optim.Adam(params, lr=opt_lr, weight_decay=opt_weight_decay)`` with two parameter groups -- no weight decay on
1-D parameters and biases (optim_factory.py:25-60; cfg_vlsa_conch.yaml:111-113: lr 2e-4, wd 1e-5) -- and calls ``optimizer.step()`` once
per 32 bags (runner/vlsa_handler.py:283-289).  Same constructor surface and the same arithmetic here (amsgrad / maximize / foreach are
not offered: the reference does not use them).  What it is for: inside the hipGraph-replayed step (``vlsa_amd.train_step.TrainStep``)
every dependent launch costs ~5 us, and torch's fused Adam needs four launches of ~15 us for six small tensors.  Step counter and
per-group (lr, weight decay) live in device memory, so a captured ``step()`` replays correctly and a learning-rate schedule only has to
call ``sync_hyper()`` (``step()`` does it itself) -- no re-capture.

State layout follows torch's: ``state[p] = {"step", "exp_avg", "exp_avg_sq"}`` (``step`` is ONE shared device counter exposed under every
parameter), so ``state_dict()`` / ``load_state_dict()`` round-trip with ``torch.optim.Adam`` checkpoints of the same parameters.
"""
from __future__ import annotations

import ctypes

import torch

from . import _native as nat
from .functional import _stream


class FusedAdam(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0):
        if lr < 0 or eps < 0 or weight_decay < 0 or not (0 <= betas[0] < 1 and 0 <= betas[1] < 1):
            raise ValueError("invalid Adam hyper-parameters")
        super().__init__(params, dict(lr=lr, betas=tuple(betas), eps=eps, weight_decay=weight_decay, capturable=True))
        b0 = self.param_groups[0]["betas"], self.param_groups[0]["eps"]
        for g in self.param_groups:
            if (g["betas"], g["eps"]) != b0:
                raise ValueError("FusedAdam takes one (betas, eps) for all groups (lr and weight_decay may differ per group)")
        self._dev = None
        self._counter = None           # device int32[2]: steps taken | launch ticket
        self._hyper = None             # device float32[2 * groups]: (lr, weight_decay) per group
        self._hyper_host = None
        self._table = None             # (key, ctypes array): rebuilt when a gradient tensor moved

    # -- device-side state ---------------------------------------------------------------------------------------------------------
    def _init_device(self, dev):
        self._dev = dev
        steps = 0
        for g in self.param_groups:                      # a loaded torch.optim.Adam checkpoint: its per-parameter step counts
            for p in g["params"]:
                st = self.state.get(p)
                if st and "step" in st:
                    steps = max(steps, int(float(st["step"])))
        self._counter = torch.tensor([steps, 0], dtype=torch.int32, device=dev)
        self._hyper = torch.zeros(2 * len(self.param_groups), dtype=torch.float32, device=dev)
        self._hyper_host = None
        self.sync_hyper()

    def sync_hyper(self):
        """copy the groups' (lr, weight_decay) to the device table if they changed (a scheduler writes ``param_groups[i]['lr']``);
        called by ``step()``; ``TrainStep`` calls it before every graph replay"""
        if self._hyper is None:
            return
        cur = [float(v) for g in self.param_groups for v in (g["lr"], g["weight_decay"])]
        if cur != self._hyper_host:
            if torch.cuda.is_current_stream_capturing():
                raise RuntimeError("FusedAdam: learning rate / weight decay changed inside a graph capture; call sync_hyper() before capturing")
            self._hyper.copy_(torch.tensor(cur, dtype=torch.float32), non_blocking=False)
            self._hyper_host = cur

    def _state_of(self, p):
        st = self.state[p]
        if "exp_avg" not in st:
            st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
            st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
        st["step"] = self._counter[0]                    # the shared device counter (a 0-dim view: torch's capturable layout)
        return st

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        live = [(gi, p) for gi, g in enumerate(self.param_groups) for p in g["params"] if p.grad is not None]
        if not live:
            return loss
        dev = live[0][1].device
        if self._dev is None:
            self._init_device(dev)
        self.sync_hyper()
        key = tuple((p.data_ptr(), p.grad.data_ptr(), gi) for gi, p in live)
        if self._table is None or self._table[0] != key:
            arr = (nat.AdamTensor * len(live))()
            for i, (gi, p) in enumerate(live):
                if p.dtype != torch.float32 or p.grad.dtype != torch.float32 or not p.is_contiguous() or not p.grad.is_contiguous() or p.device != dev:
                    raise nat.VlsaNativeError("FusedAdam takes contiguous fp32 parameters and gradients on one device")
                st = self._state_of(p)
                arr[i] = nat.AdamTensor(p.data_ptr(), p.grad.data_ptr(), st["exp_avg"].data_ptr(), st["exp_avg_sq"].data_ptr(), p.numel(), gi)
            self._table = (key, arr)
        g0 = self.param_groups[0]
        nat.check(nat.load().vlsa_adam_step(ctypes.cast(self._table[1], ctypes.c_void_p), len(live), ctypes.c_void_p(self._hyper.data_ptr()),
                                            ctypes.c_void_p(self._counter.data_ptr()), float(g0["betas"][0]), float(g0["betas"][1]),
                                            float(g0["eps"]), _stream()), "vlsa_adam_step")
        for _, p in live:                                # the launch wrote the parameters: what an in-place torch op would record
            torch._C._increment_version([p]) if _ITERABLE_BUMP else torch._C._increment_version(p)
        return loss

    def load_state_dict(self, state_dict):
        super().load_state_dict(state_dict)
        self._dev = self._counter = self._hyper = self._table = None      # re-read the counters / moments on the next step


def _probe_bump():
    t = torch.zeros(1)
    try:
        torch._C._increment_version([t])
        return True
    except (RuntimeError, TypeError):
        return False


_ITERABLE_BUMP = _probe_bump()
