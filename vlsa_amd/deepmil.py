"""MIL encoders usable as ``VLSA.mil_encoder`` -- drop-in counterparts of the reference's ``model/deepmil.py``
classes (same constructor kwargs, attribute names and state-dict keys), with the N-sized arithmetic in HIP.

    VLFAN         model/deepmil.py:74-215     language-guided cross-attention aggregation (the shipped encoder)
    FeatMIL       model/deepmil.py:40-67      mean / max / identity (zero-shot)
    DeepMIL       model/deepmil.py:222-292    ABMIL-style (gated-)attention pooling over the N patches
    logit_pooling model/deepmil.py:16-37      top-k / mean pooling of per-patch class logits

The reference picks the encoder with ``getattr(model.deepmil, cfg['name'])(**cfg)`` (model/utils_vl.py:129-138);
``vlsa_amd.vlsa.build_mil_encoder`` does the same lookup in this module.
"""
from __future__ import annotations

import math

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import functional as VF
from .layers import Adapter, Attention_Pooling, Feat_Projecter, Gated_Attention_Pooling

__all__ = ["logit_pooling", "FeatMIL", "VLFAN", "DeepMIL"]


_LOGIT_POOLINGS = {"logit_mean": None, "logit_max": 1}


def _parse_logit_pooling(method: str):
    """'logit_mean' -> None (all patches), 'logit_max' -> 1, 'logit_top<k>' -> k; anything else is not a logit pooling."""
    if method in _LOGIT_POOLINGS:
        return _LOGIT_POOLINGS[method]
    head, _, digits = method.partition("logit_top")
    if head == "" and digits.isdigit():
        return int(digits)
    raise NotImplementedError(f"The pooling ({method}) is not implemented.")


def logit_pooling(logits: torch.Tensor, method: str):
    """logits: [N, C] per-patch class logits -> (preds[1], pooled[1, C]).  (model/deepmil.py:16-37)
    The per-class top-k mean runs in HIP; when the logits carry a gradient (zero-shot prompt tuning) the selection is
    done with torch.topk so that autograd reaches them, as in the reference."""
    topk = _parse_logit_pooling(method)
    N = logits.size(0)
    k = N if topk is None else min(topk, N)
    if torch.is_grad_enabled() and logits.requires_grad:
        pooled = logits.topk(k, dim=0).values.mean(dim=0, keepdim=True) if k < N else logits.mean(dim=0, keepdim=True)
    else:
        pooled = VF.topk_mean(logits.t().contiguous(), k)[None, :]
    return pooled.argmax(dim=1), pooled


class FeatMIL(nn.Module):
    """Feature aggregation only: 'mean' | 'max' over the patches, anything else = identity (zero-shot)."""

    def __init__(self, pooling="mean", **kwargs):
        super().__init__()
        self.network = nn.Identity()
        self.pooling = pooling

    def forward(self, X):
        assert X.shape[0] == 1
        if self.pooling == "mean":
            return VF.scored_pool(X, None)[None, :]
        if self.pooling == "max":
            return VF.colmax(X)[None, :]
        return X.squeeze(0)


class VLFAN(VF.nat.TransientCaches, nn.Module):
    """Language-guided visual feature aggregation network (model/deepmil.py:74-215).

    P query vectors (text prototypes through a query network, or an ``nn.Parameter``) cross-attend over the N
    patches with cosine scores x 100, softmax over the patches; the P aggregated rows are pooled over the queries
    and passed through the visual adapter.  The cross attention runs in the HIP streaming kernels.
    """

    _transient = {"_step_query": None, "_enc_plans": None, "_fused_scores": None, "_coattn_scale": None}

    def __init__(self, dim_in=1024, dim_hid=256, use_feat_proj=True, drop_rate=0.25, query="Parameter", num_query=10,
                 gated_query=False, query_pooling="mean", pred_head="default", dim_reduction=4, keep_ratio=0.8, **kwargs):
        super().__init__()
        self._pos_gated_query = -1
        self.feat_proj = Feat_Projecter(dim_in, dim_in) if use_feat_proj else None
        self.num_query = num_query
        self.query_type = query
        self.gated_query = gated_query
        assert self.query_type in ["Parameter", "Text"]
        if self.query_type != "Parameter":
            self.Q = None  # call reset_query() with the query network
        else:
            self.Q = nn.Parameter(torch.randn(num_query + 1 if gated_query else num_query, dim_in))
        assert query_pooling in ["mean", "max", "weight", "attention", "gated_attention"]
        if query_pooling == "attention":
            self.query_pooling = Attention_Pooling(dim_in, dim_hid)
        elif query_pooling == "gated_attention":
            self.query_pooling = Gated_Attention_Pooling(dim_in, dim_hid, dropout=drop_rate)
        elif query_pooling == "weight":
            self.query_pooling = nn.Parameter(torch.randn(1, num_query))
        else:
            self.query_pooling = query_pooling
        self.pred_head = pred_head
        self.visual_adapter = nn.Identity() if pred_head == "Identity" else nn.Linear(dim_in, dim_in)
        self.use_custom_coattn = True
        self.coattn_logit_scale = torch.ones([]) * math.log(100)  # plain tensor attribute, as in the reference

    # -- reference API ---------------------------------------------------------------------------------
    def get_coattn_logit_scale(self):
        return self.coattn_logit_scale.exp()

    def coattn_scale(self) -> float:
        """exp(coattn_logit_scale) as a Python float, cached per (tensor object, in-place version): the per-bag routes ask for it
        once per bag."""
        t = self.coattn_logit_scale
        c = self.__dict__.get("_coattn_scale")
        if c is None or c[0] is not t or c[1] != t._version:
            c = self.__dict__["_coattn_scale"] = (t, t._version, float(t.detach().exp()))
        return c[2]

    def reset_query(self, query_network):
        assert self.query_type != "Parameter", f"Cannot override Q (query) for query_type ({self.query_type})."
        self.Q = query_network

    def get_query(self):
        assert self.Q is not None, f"You have to call `reset_query` to reset query for query_type ({self.query_type})."
        return self.Q() if callable(self.Q) else self.Q

    def step_query(self):
        """``get_query()`` for the per-bag training path: when the queries come from a module (the text PromptAdapter), its
        output -- WITH its autograd graph -- is shared by all bags that see the same parameter versions, train / eval flag and
        grad mode (the reference's step evaluates the adapter once per bag: runner/vlsa_handler.py:267-269; same values, 32 x
        the launches and autograd nodes).  A backward pass through the cached tensor frees that graph: the next call rebuilds.
        NOT shared when the query network is stochastic: an active ``nn.Dropout`` (the 'FC' PromptAdapter has Dropout(0.25),
        model/prompt_learners/prompt_adapter.py:95-104) draws a fresh mask per bag in the reference, so every call evaluates
        the network again -- a cached output would give all bags of a step (a frozen adapter: of the whole run) one mask."""
        Qs = self.Q
        if not isinstance(Qs, nn.Module):
            return self.get_query()
        key = [torch.is_grad_enabled()]
        stack = [Qs]
        while stack:
            m = stack.pop()
            if m.training and isinstance(m, nn.modules.dropout._DropoutNd) and m.p > 0:
                self._step_query = None
                return self.get_query()
            key.append((id(m), m.training))
            for t in m._parameters.values():
                if t is not None:
                    key.append((id(t), t._version, t.requires_grad))
            for t in m._buffers.values():
                if t is not None:
                    key.append((id(t), t._version))
            stack.extend(c for c in m._modules.values() if c is not None)
        key = tuple(key)
        cached = getattr(self, "_step_query", None)
        if cached is None or cached[0] != key:
            q = self.get_query()
            if q.requires_grad and q.grad_fn is not None:
                q.register_hook(self._drop_step_query)
            cached = self._step_query = (key, q)
        return cached[1]

    def _drop_step_query(self, *_):
        self._step_query = None

    def query_div_loss(self, last_div=True, **kws):
        """Diversity penalty on the queries = mean |cosine| (model/deepmil.py:157-168): between the gate query (last row)
        and every other query when there is one and ``last_div``; otherwise over all ordered pairs of distinct queries."""
        unit = F.normalize(self.get_query(), dim=-1)
        n = unit.shape[0]
        if last_div and n == self.num_query + 1:
            cos = unit[:-1] @ unit[-1]                       # [P]: gate vs the P prototypes
            return cos.abs().mean()
        gram = (unit @ unit.t()).abs()
        off_diag = gram.sum() - gram.diagonal().sum()        # the diagonal (self-similarity) does not count
        return off_diag / (n * (n - 1))

    def forward_query_pooling(self, X):
        """[B, P, C] aggregated rows -> ([B, C] pooled, scores of the pooling module or None)  (model/deepmil.py:133-150)"""
        pool = self.query_pooling
        if isinstance(pool, nn.Module):                       # Attention_Pooling / Gated_Attention_Pooling over the P rows
            needs_graph = torch.is_grad_enabled() and (X.requires_grad or any(p.requires_grad for p in pool.parameters()))
            dropout_on = pool.training and isinstance(pool, Gated_Attention_Pooling) and pool.fc1[2].p > 0
            if X.is_cuda and not needs_graph and not dropout_on and X.shape[1] <= 16 and X.shape[2] <= 1024:
                return VF.query_pool_attention(X, pool)       # inference: two HIP launches instead of ~10 torch ops on [P, 512]
            return pool(X)
        if isinstance(pool, nn.Parameter):                    # 'weight': learnable convex combination of the P rows
            mix = torch.softmax(pool, dim=-1)                 # [1, P]
            return torch.einsum("op,bpc->bc", mix, X), None
        if pool == "mean":
            return X.mean(dim=1), None
        return X.amax(dim=1), None

    # -- fused inference support -------------------------------------------------------------------------
    def fused_head_spec(self):
        """(pool mode, pool weight, W, b) if pooling + adapter can run in the fused HIP head, else None.  A Feat_Projecter
        does not stand in the way: the inference routes project the bag first (``project``: one fused HIP launch per bag)."""
        if isinstance(self.query_pooling, str):
            mode, pw = self.query_pooling, None
        elif isinstance(self.query_pooling, nn.Parameter):
            mode, pw = "weight", self.query_pooling
        elif not (self.query_pooling.training and isinstance(self.query_pooling, Gated_Attention_Pooling) and self.query_pooling.fc1[2].p > 0):
            mode, pw = "module", self.query_pooling       # (gated-)attention over the P rows: vlsa_query_pool_attention
        else:
            return None
        if isinstance(self.visual_adapter, nn.Linear):
            return mode, pw, self.visual_adapter.weight, self.visual_adapter.bias
        return mode, pw, None, None

    def project(self, X):
        """The bag as the cross attention sees it: Feat_Projecter(X) when use_feat_proj=True (model/deepmil.py:176-179)."""
        return X if self.feat_proj is None else self.feat_proj(X)

    def _projecter_trains(self) -> bool:
        return (self.feat_proj is not None and torch.is_grad_enabled()
                and any(p.requires_grad for p in self.feat_proj.parameters()))

    def _fused_encode(self, X, ret_with_attn):
        """Inference, nothing to differentiate: the whole encoder (query preparation, streaming aggregation, merge, attention
        weights, query pooling, adapter) as ONE host call into the C ABI (``VlfanInferencePlan``; the text part of the fused
        head is fed a dummy class).  None when the configuration has no fused form."""
        spec = self.fused_head_spec()
        if (spec is None or spec[0] == "module" or torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters())
                or not X.is_cuda or X.shape[-1] != 512 or X.shape[1] == 0 or X.dtype not in (torch.float32, torch.bfloat16)):
            return None
        mode, pw, W, b = spec
        X2 = VF._bag2d(X)
        Q = self.get_query()
        if torch.is_grad_enabled() and Q.requires_grad:
            return None      # a query that needs a gradient but is no registered parameter (closure / plain callable): autograd route
        Q = Q.detach().float().contiguous()
        P = Q.shape[0] - (1 if self.gated_query else 0)
        if not (1 <= P <= 16):
            return None
        N, dev = X2.shape[0], X2.device
        plans = self.__dict__.setdefault("_enc_plans", {})
        key = (N, dev, bool(ret_with_attn), P, mode, W is None)
        plan = plans.get(key)
        if plan is None:
            if len(plans) > 32:
                plans.clear()
            plan = plans[key] = VF.VlfanInferencePlan(N, 512, P, 1, dev, gated=self.gated_query, pool=mode, identity_head=W is None,
                                                      want_attn=ret_with_attn, coattn_scale=self.coattn_scale())
            plan._dummy = (torch.ones(1, 512, device=dev), torch.zeros((), device=dev))
        outs = {"v": torch.empty(512, dtype=torch.float32, device=dev)}
        if ret_with_attn:
            outs["A"] = torch.empty(P, N, dtype=torch.float32, device=dev)
        plan.run(X2, Q, plan._dummy[0], plan._dummy[1], None if W is None else W.detach().float().contiguous(),
                 None if b is None else b.detach().float().contiguous(),
                 None if pw is None else pw.detach().float().reshape(-1).contiguous(), outs=outs)
        v = outs["v"][None]
        return (v, outs["A"][None]) if ret_with_attn else v

    def forward(self, X, ret_with_attn=False):
        assert X.shape[0] == 1
        if self.feat_proj is not None:
            X = self.feat_proj(X)
        if not (torch.is_grad_enabled() and X.requires_grad):
            fused = self._fused_encode(X, ret_with_attn)
            if fused is not None:
                return fused
        Q = self.get_query()
        if self.gated_query:
            assert self._pos_gated_query == -1, "The gated query is placed at the end by default."
            assert Q.shape[0] == self.num_query + 1, f"Query number is expected to be {self.num_query + 1}."
        scale = self.coattn_scale()
        # HIP forward and backward: dQ always, dX as well when the bag carries a gradient (a trainable Feat_Projecter in front:
        # its output is an fp32 [N, 512] bag; vlsa_vlfan_backward_dx)
        out, A = VF.vlfan_cross_attention(X, Q, gated=self.gated_query, coattn_scale=scale, want_attn=ret_with_attn)
        pooled_out, pooled_ext = self.forward_query_pooling(out.unsqueeze(0))
        visual_features = self.visual_adapter(pooled_out)
        if ret_with_attn:
            A = A.unsqueeze(0)
            attn = (A, pooled_ext.detach()) if pooled_ext is not None else A
            return visual_features, attn
        return visual_features

    def forward_bags(self, bags, ret_with_attn=False, projected=False):
        """Differentiable forward over a list of bags (each [1, N_i, C] or [N_i, C]) sharing this encoder: the cross
        attention of all bags runs in the persistent multi-bag kernels (forward and backward), the P x C tail as batched
        torch ops.  Equals ``torch.cat([self(x) for x in bags])``; one training step of the reference
        (runner/vlsa_handler.py:260-289) is 32 such bags.  Returns visual features [B, C]; with ``ret_with_attn`` also the
        per-bag attention exactly as ``forward(x, ret_with_attn=True)`` hands it out: a list of ``A`` [1, P, N_i], or of
        ``(A, pool_scores)`` when the query pooling is a module (model/deepmil.py:206-215).  ``projected``: the bags already
        went through ``project``."""
        if self.feat_proj is not None and not projected:
            # one fused HIP launch per bag, fp32 out; a projecter that trains records its LayerNorm statistics and receives its
            # gradient from the HIP backward kernels (the aggregation hands dX back for the projected bags)
            bags = [self.feat_proj(x) for x in bags]
        outs_cat, attn = self._aggregate_bags(bags, ret_with_attn)
        pooled_out, pooled_ext = self.forward_query_pooling(outs_cat)
        feats = self.visual_adapter(pooled_out)
        if not ret_with_attn:
            return feats
        if pooled_ext is not None:
            attn = [(a, pooled_ext[i:i + 1].detach()) for i, a in enumerate(attn)]
        return feats, attn

    def aggregate_bags(self, bags):
        """The P aggregated rows of every bag, [B, P, C], differentiable w.r.t. the queries (persistent multi-bag kernels forward
        and backward) -- ``forward_bags`` without the query pooling and the adapter.  A trainable Feat_Projecter is part of
        the graph (HIP forward + backward; the aggregation hands dX back for the projected fp32 bags)."""
        if self.feat_proj is not None:
            bags = [self.feat_proj(x) for x in bags]
        return self._aggregate_bags(bags, False)[0]

    def _aggregate_bags(self, bags, ret_with_attn):
        Q = self.get_query()
        scale = self.coattn_scale()
        outs, attn = [], []
        is_set = isinstance(bags, VF.BagSet)        # checked once, descriptor rows kept: sub-sets stay BagSets
        for i in range(0, len(bags), 64):
            r = VF.vlfan_cross_attention_bags(bags.chunk(i, 64) if is_set else bags[i:i + 64], Q, gated=self.gated_query, coattn_scale=scale,
                                              want_attn=ret_with_attn)
            if ret_with_attn:
                outs.append(r[0])
                attn.extend(a.unsqueeze(0) for a in r[1])
            else:
                outs.append(r)
        return (outs[0] if len(outs) == 1 else torch.cat(outs)), attn


class DeepMIL(VF.nat.TransientCaches, nn.Module):
    """ABMIL-style encoder (model/deepmil.py:222-292): optional Feat_Projecter, mean / max / (gated-)attention pooling
    over the N patches, Adapter head mixed with ``keep_ratio`` or a Linear head."""

    _transient = {"_fused_scores": None}

    def __init__(self, dim_in=1024, dim_hid=256, num_cls=2, use_feat_proj=True, drop_rate=0.25, pooling="attention",
                 pred_head="default", dim_reduction=4, keep_ratio=0.8, **kwargs):
        super().__init__()
        assert pooling in ["mean", "max", "attention", "gated_attention"]
        assert pred_head in ["default", "Adapter"]
        self.feat_proj = Feat_Projecter(dim_in, dim_in) if use_feat_proj else None
        if pooling == "gated_attention":
            self.sigma = Gated_Attention_Pooling(dim_in, dim_hid, dropout=drop_rate)
        elif pooling == "attention":
            self.sigma = Attention_Pooling(dim_in, dim_hid)
        else:
            self.sigma = pooling
        self.pred_head = pred_head
        if pred_head == "Adapter":
            assert 0 <= keep_ratio <= 1.0
            self.keep_ratio = keep_ratio
            self.visual_adapter = Adapter(dim_in, dim_reduction)
        else:
            self.g = nn.Linear(dim_in, num_cls)

    def pool_bags(self, flat):
        """Pooled bag vectors [B, C] of a list of validated [N_i, 512] device bags (one dtype) without autograd: the
        (gated-)attention poolings run as ONE score launch and ONE pooling launch per <= 64 bags when the fused kernel applies
        (512 -> 256 hidden, no active dropout), otherwise bag by bag.  model/deepmil.py:270-283 per bag."""
        sg = self.sigma
        if isinstance(sg, str):
            if sg == "mean" and len(flat) > 0 and all(x.is_cuda and x.shape[1] == 512 and x.shape[0] > 0
                                                      and x.dtype == flat[0].dtype for x in flat):
                return torch.cat([VF.mean_pool_bags(flat[i:i + 64]) for i in range(0, len(flat), 64)])
            return torch.stack([VF.scored_pool(x, None) if sg == "mean" else VF.colmax(x) for x in flat])
        gated = isinstance(sg, Gated_Attention_Pooling)
        lin_a = sg.fc1[0] if gated else sg.attention[0]
        if (not (gated and sg.training and sg.fc1[2].p > 0) and len(flat) > 0
                and all(VF.FusedAttnScores.supported(x, lin_a.in_features, lin_a.out_features) for x in flat)):
            if not hasattr(self, "_fused_scores"):
                self._fused_scores = VF.FusedAttnScores()
            w = ((lin_a.weight, lin_a.bias, sg.score[0].weight, sg.score[0].bias, sg.fc2.weight, sg.fc2.bias) if gated else
                 (lin_a.weight, lin_a.bias, None, None, sg.attention[2].weight, sg.attention[2].bias))
            return torch.cat([self._fused_scores.pool_bags(flat[i:i + 64], *w)[0] for i in range(0, len(flat), 64)])
        return torch.stack([VF.scored_pool(x, self._attention_scores(x)) for x in flat])

    def _attention_scores(self, X2):
        """raw scores a[N] of the pooling module on all patches.  512 -> 256 hidden (the reference's sizes): ONE fused MFMA kernel,
        hidden activations in registers -- inference, and under autograd as well (HIP backward that recomputes them tile by tile,
        vlsa_attn_scores_backward), the gated module's training-mode dropout included (counter-based masks inside both kernels).
        Other widths, or a bag that itself carries a gradient: library GEMMs for the hidden projections + torch / HIP elementwise."""
        sg = self.sigma
        gated = isinstance(sg, Gated_Attention_Pooling)
        lin_a = sg.fc1[0] if gated else sg.attention[0]
        grad_mode = torch.is_grad_enabled()
        bag_grad = grad_mode and X2.requires_grad
        need_grad = bag_grad or (grad_mode and any(p.requires_grad for p in sg.parameters()))
        drop_p = float(sg.fc1[2].p) if (gated and sg.training and sg.fc1[2].p > 0) else 0.0
        if (not bag_grad and (not gated or sg.fc1[2].p == sg.score[2].p)
                and VF.FusedAttnScores.supported(X2, lin_a.in_features, lin_a.out_features)):
            if not hasattr(self, "_fused_scores"):
                self._fused_scores = VF.FusedAttnScores()
            w = ((lin_a.weight, lin_a.bias, sg.score[0].weight, sg.score[0].bias, sg.fc2.weight, sg.fc2.bias) if gated else
                 (lin_a.weight, lin_a.bias, None, None, sg.attention[2].weight, sg.attention[2].bias))
            if need_grad or drop_p > 0:
                return VF.attn_scores_autograd(X2, self._fused_scores, *w, drop_p=drop_p)
            return self._fused_scores(X2, *w)
        VF.nat.note_torch_route("DeepMIL attention scores", X2.shape[0],
                                f"widths {lin_a.in_features} -> {lin_a.out_features}, bag gradient {bag_grad}: the fused score kernel covers 512 -> 256 "
                                "hidden units, bags without a gradient of their own and equal dropout rates in both branches")
        Xf = X2 if X2.dtype == torch.float32 else X2.float()
        if isinstance(sg, Attention_Pooling):
            lin1, lin2 = sg.attention[0], sg.attention[2]
            H = Xf @ lin1.weight.t()
            if need_grad:
                return (torch.tanh(H + lin1.bias) @ lin2.weight.t() + lin2.bias).squeeze(-1)
            return VF.attn_scores(H, None, lin1.bias, None, lin2.weight, lin2.bias)
        la, lg, l2 = sg.fc1[0], sg.score[0], sg.fc2
        H, Hg = Xf @ la.weight.t(), Xf @ lg.weight.t()
        if need_grad or drop_p > 0:
            e = sg.fc1[2](torch.tanh(H + la.bias)) * sg.score[2](torch.sigmoid(Hg + lg.bias))
            return (e @ l2.weight.t() + l2.bias).squeeze(-1)
        return VF.attn_scores(H, Hg, la.bias, lg.bias, l2.weight, l2.bias)

    def forward(self, X, ret_with_attn=False):
        assert X.shape[0] == 1
        if self.feat_proj is not None:
            X = self.feat_proj(X)
        raw_attn = fused_logit = None
        x_grad = torch.is_grad_enabled() and X.requires_grad   # a trainable Feat_Projecter in front: the pooling must hand dX back
        if self.sigma == "mean":
            out_feat = X.float().mean(dim=1) if x_grad else VF.scored_pool(X, None)[None, :]
        elif self.sigma == "max":
            out_feat = X.float().amax(dim=1) if x_grad else VF.colmax(X)[None, :]
        else:
            X2 = VF._bag2d(X)
            sg = self.sigma
            gated = isinstance(sg, Gated_Attention_Pooling)
            sm = sg._modules          # (plain dict look-ups: seven nn.Sequential.__getitem__ calls are 10 us of a 40 us module call)
            if gated:
                f1, sc = sm["fc1"]._modules, sm["score"]._modules
                lin_a, lin_g, lin_o, drop_a, drop_g = f1["0"], sc["0"], sm["fc2"], f1["2"], sc["2"]
            else:
                at = sm["attention"]._modules
                lin_a, lin_g, lin_o, drop_a, drop_g = at["0"], None, at["2"], None, None
            if (x_grad and (not gated or drop_a.p == drop_g.p)
                    and VF.FusedAttnScores.supported(X2, lin_a.in_features, lin_a.out_features)):
                # scores + pooling as ONE autograd node whose backward is HIP end to end: parameter gradients AND
                # dX = dHa Wa + dHg Wg + A dpooled (vlsa_attn_scores_backward_dx); 512 -> 256 hidden, the reference's sizes
                if not hasattr(self, "_fused_scores"):
                    self._fused_scores = VF.FusedAttnScores()
                w = (lin_a.weight, lin_a.bias, lin_g.weight if gated else None, lin_g.bias if gated else None, lin_o.weight, lin_o.bias)
                drop_p = float(drop_a.p) if (gated and sg.training and drop_a.p > 0) else 0.0
                pooled, a = VF.attn_pool_autograd(X2, self._fused_scores, *w, drop_p=drop_p)
                out_feat = pooled[None, :]
            else:
                fused = None
                if (not torch.is_grad_enabled() and not (gated and sg.training and drop_a.p > 0)
                        and VF.FusedAttnScores.supported(X2, lin_a.in_features, lin_a.out_features)):
                    # inference on a large bf16 bag: scores and pooling in ONE launch (vlsa_gated_scores_pool / _adapter)
                    if not hasattr(self, "_fused_scores"):
                        self._fused_scores = VF.FusedAttnScores()
                    w = (lin_a.weight, lin_a.bias, lin_g.weight if gated else None, lin_g.bias if gated else None, lin_o.weight, lin_o.bias)
                    adapter = None
                    if self.pred_head == "Adapter":
                        fc = self.visual_adapter._modules["fc"]._modules
                        W1, W2 = fc["0"].weight, fc["2"].weight
                        if W1.shape[1] == 512 and W2.shape[0] == 512 and W1.shape[0] % 4 == 0:
                            adapter = (W1, W2, self.keep_ratio)
                    fused = self._fused_scores.scores_and_pool(X2, *w, adapter=adapter)
                if fused is not None:
                    out_feat, a = fused[0], fused[1]
                    if len(fused) == 3:
                        fused_logit = fused[2]
                else:
                    a = self._attention_scores(X2)
                    if x_grad:         # other layer widths than 512 -> 256: library GEMMs (see _attention_scores) + torch pooling
                        out_feat = torch.softmax(a, dim=0)[None, :] @ X2.float()
                    else:
                        out_feat = VF.scored_pool(X2, a)[None, :]
            if ret_with_attn:  # what the reference hands back: raw scores (attention) / softmax weights (gated attention)
                raw_attn = a[None, :] if isinstance(self.sigma, Attention_Pooling) else F.softmax(a, dim=0)[None, :]
        if fused_logit is not None:
            logit = fused_logit                      # (computed behind the pooling in the same host call)
        elif self.pred_head == "Adapter":
            fc = self.visual_adapter.fc
            if not (torch.is_grad_enabled() and any(p.requires_grad for p in fc.parameters())) and out_feat.is_cuda:
                logit = VF.adapter_head(out_feat, fc[0].weight, fc[2].weight, self.keep_ratio)[None, :]   # two small HIP launches
            else:
                logit = self.keep_ratio * out_feat + (1 - self.keep_ratio) * self.visual_adapter(out_feat)
        else:
            logit = self.g(out_feat)
        if ret_with_attn:
            return logit, raw_attn.detach()
        return logit
