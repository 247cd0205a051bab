"""Pooling / adapter layers with the reference's attribute and state-dict names (model/layers.py:50-153).

These operate on small inputs in the shipped configuration (the P <= 16 aggregated rows inside VLFAN, the
[1, D] bag vector in the Adapter head): plain torch modules on the device.  When DeepMIL applies the attention
poolings to all N patches the N-sized part runs in HIP (vlsa_amd.deepmil.DeepMIL)."""
from __future__ import annotations

import torch
import torch.nn as nn
import torch.nn.functional as F

from ._native import TransientCaches as _TransientCaches


class Adapter(nn.Module):
    """Bias-free bottleneck MLP with ReLU after both layers (model/layers.py:50-62). Keys: fc.0.weight, fc.2.weight."""

    def __init__(self, c_in: int, reduction: int = 4):
        super().__init__()
        hid = c_in // reduction
        self.fc = nn.Sequential(nn.Linear(c_in, hid, bias=False), nn.ReLU(inplace=True),
                                nn.Linear(hid, c_in, bias=False), nn.ReLU(inplace=True))

    def forward(self, x):
        return self.fc(x)


class Feat_Projecter(_TransientCaches, nn.Module):
    """Linear + LayerNorm on every row (model/layers.py:65-82). Keys: projecter.0.*, projecter.1.*."""
    _transient = {"_fused": None}

    def __init__(self, in_dim: int = 1024, out_dim: int = 1024):
        super().__init__()
        self.projecter = nn.Sequential(nn.Linear(in_dim, out_dim), nn.LayerNorm(out_dim))

    def _needs_autograd(self, x) -> bool:
        return torch.is_grad_enabled() and (x.requires_grad or any(p.requires_grad for p in self.parameters()))

    def forward(self, x):
        lin, norm = self.projecter[0], self.projecter[1]
        bag_grad = torch.is_grad_enabled() and x.requires_grad
        if x.is_cuda and x.dim() in (2, 3) and not bag_grad:
            # on the device: ONE fused HIP kernel per bag (vlsa_feat_project), fp32 out for bf16 or fp32 bags; when the
            # projecter trains, the same kernel also stores the LayerNorm statistics and the backward is HIP as well
            # (vlsa_feat_project_backward: dW, db, dgamma, dbeta -- the bags themselves carry no gradient)
            from . import functional as VF
            x2 = x.reshape(-1, x.shape[-1])
            if VF.FusedFeatProjecter.supported(x2, lin, norm) and norm.elementwise_affine:
                if not hasattr(self, "_fused"):
                    self._fused = VF.FusedFeatProjecter()
                if self._needs_autograd(x):
                    y = self._fused.autograd(x2, lin.weight, lin.bias, norm.weight, norm.bias, norm.eps)
                else:
                    y = self._fused(x2, lin.weight, lin.bias, norm.weight, norm.bias, norm.eps)
                return y.reshape(*x.shape[:-1], 512)
        if x.dtype != lin.weight.dtype:
            x = x.to(lin.weight.dtype)          # bf16 bags of the resident arena: the modules compute in the parameters' dtype
        if x.dim() == 3:
            b, n, d = x.shape
            return self.projecter(x.reshape(-1, d)).reshape(b, n, -1)
        return self.projecter(x)


def _note(module, x):
    """the pooling modules' own ``forward`` is plain torch: meant for the P query rows inside VLFAN; ``DeepMIL`` drives the fused
    kernels with the modules' parameters instead of calling it on a bag"""
    if x.is_cuda:
        from ._native import note_torch_route
        note_torch_route(type(module).__name__ + ".forward", x.shape[-2], "called directly on the rows of a bag; DeepMIL.forward / "
                         "VF.attn_pool_autograd run the same parameters through the fused score + pooling kernels")


class Gated_Attention_Pooling(nn.Module):
    """ABMIL gated attention (model/layers.py:85-122). Keys: fc1.0.*, score.0.*, fc2.*.
    Returns (pooled[B, d], attn) where attn is the softmax weights, or the raw scores if ret_raw_attn."""

    def __init__(self, in_dim: int, hid_dim: int, dropout: float = 0.5):
        super().__init__()
        self.fc1 = nn.Sequential(nn.Linear(in_dim, hid_dim), nn.Tanh(), nn.Dropout(dropout))
        self.score = nn.Sequential(nn.Linear(in_dim, hid_dim), nn.Sigmoid(), nn.Dropout(dropout))
        self.fc2 = nn.Linear(hid_dim, 1)

    def forward(self, x, ret_raw_attn: bool = False):
        if x.dim() == 2:
            x = x.unsqueeze(0)
        _note(self, x)
        raw = self.fc2(self.fc1(x) * self.score(x)).transpose(2, 1)  # [B, 1, n]
        attn = F.softmax(raw, dim=2)
        out = torch.matmul(attn, x).squeeze(1)
        return (out, raw.squeeze(1)) if ret_raw_attn else (out, attn.squeeze(1))


class Attention_Pooling(nn.Module):
    """ABMIL attention (model/layers.py:125-153). Keys: attention.0.*, attention.2.*.
    The reference returns the RAW scores by default; its ret_raw_attn=False branch raises (UnboundLocalError,
    SURVEY.md 7.4-9) -- here that branch returns the softmax weights, which is what the code evidently meant."""

    def __init__(self, in_dim: int = 1024, hid_dim: int = 512):
        super().__init__()
        self.attention = nn.Sequential(nn.Linear(in_dim, hid_dim), nn.Tanh(), nn.Linear(hid_dim, 1))

    def forward(self, x, ret_raw_attn: bool = True):
        if x.dim() == 2:
            x = x.unsqueeze(0)
        _note(self, x)
        raw = self.attention(x).transpose(2, 1)  # [B, 1, n]
        attn = F.softmax(raw, dim=2)
        out = torch.matmul(attn, x).squeeze(1)
        return (out, raw.squeeze(1)) if ret_raw_attn else (out, attn.squeeze(1))
