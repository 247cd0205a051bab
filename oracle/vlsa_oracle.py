"""CPU restatement of the reference's per-bag forward (TEST INFRASTRUCTURE -- see oracle/__init__.py).

Every function follows the op ORDER of the reference so that fp32 results agree to rounding; each
docstring cites the reference lines (paths relative to /root/reference).  Plain torch ops on CPU
tensors, no autograd tricks: gradients for the backward parity tests come from torch.autograd on
these very functions (exactly how the reference obtains them).

The second half (``vlfan_partial`` / ``merge_partials``) restates the SAME math in the
single-pass, shardable form the HIP kernels use; tests assert both halves agree.
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Sequence, Tuple

import torch
import torch.nn.functional as F

COATTN_SCALE = 100.0  # exp(log(100)), model/deepmil.py:120-126


# ----------------------------------------------------------------------------------------------
# Reference-order restatement
# ----------------------------------------------------------------------------------------------
def l2_normalize(x: torch.Tensor, eps: float = 1e-12) -> torch.Tensor:
    """F.normalize(x, dim=-1): x / max(||x||_2, eps)  (model/deepmil.py:187,189; model/vlsa.py:186,189)."""
    return x / x.norm(dim=-1, keepdim=True).clamp_min(eps)


def vlfan_attention_logits(X: torch.Tensor, Q: torch.Tensor, gated_query: bool = False,
                           scale: float = COATTN_SCALE) -> torch.Tensor:
    """Scaled cosine scores [P, N]  (model/deepmil.py:187-197).

    X: [N, D]; Q: [P, D] or [P+1, D] when ``gated_query`` (last row is the subtracted query).
    """
    Qn = l2_normalize(Q)
    Xn = l2_normalize(X)
    A_ = Qn @ Xn.t()
    if gated_query:
        A_ = A_[:-1, :] - A_[-1:, :]
    return scale * A_


def attention_pooling(x: torch.Tensor, w1: torch.Tensor, b1: torch.Tensor, w2: torch.Tensor,
                      b2: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
    """Attention_Pooling (model/layers.py:137-153). x: [n, d]. Returns (out[d], raw[n], attn[n]).

    The reference returns the RAW scores by default (ret_raw_attn=True); its other branch is broken.
    """
    raw = (torch.tanh(x @ w1.t() + b1) @ w2.t() + b2).squeeze(-1)
    attn = torch.softmax(raw, dim=0)
    return attn @ x, raw, attn


def gated_attention_pooling(x: torch.Tensor, wa: torch.Tensor, ba: torch.Tensor, wg: torch.Tensor,
                            bg: torch.Tensor, w2: torch.Tensor, b2: torch.Tensor
                            ) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
    """Gated_Attention_Pooling in eval mode / drop_rate 0 (model/layers.py:103-122).

    Returns (out[d], raw[n], attn[n]); the reference returns the NORMALISED attn by default.
    """
    emb = torch.tanh(x @ wa.t() + ba)
    scr = torch.sigmoid(x @ wg.t() + bg)
    raw = ((emb * scr) @ w2.t() + b2).squeeze(-1)
    attn = torch.softmax(raw, dim=0)
    return attn @ x, raw, attn


def query_pooling(out: torch.Tensor, method: str, params: Optional[Dict[str, torch.Tensor]] = None
                  ) -> Tuple[torch.Tensor, Optional[torch.Tensor]]:
    """VLFAN.forward_query_pooling (model/deepmil.py:133-150). out: [P, D] -> ([D], ext or None)."""
    if method == "mean":
        return out.mean(dim=0), None
    if method == "max":
        return out.max(dim=0).values, None
    if method == "weight":
        w = torch.softmax(params["weight"].reshape(-1), dim=-1)
        return w @ out, None
    if method == "attention":
        o, raw, _ = attention_pooling(out, params["w1"], params["b1"], params["w2"], params["b2"])
        return o, raw  # raw scores are what the reference hands back
    if method == "gated_attention":
        o, _, attn = gated_attention_pooling(out, params["wa"], params["ba"], params["wg"], params["bg"],
                                             params["w2"], params["b2"])
        return o, attn
    raise NotImplementedError(method)


def vlfan_forward(X: torch.Tensor, Q: torch.Tensor, *, gated_query: bool = False,
                  query_pooling_method: str = "mean", pooling_params=None,
                  head_weight: Optional[torch.Tensor] = None, head_bias: Optional[torch.Tensor] = None,
                  scale: float = COATTN_SCALE, feat_proj: Optional[Tuple[torch.Tensor, ...]] = None):
    """VLFAN.forward (model/deepmil.py:170-215); feat_proj = (w, b, ln_w, ln_b) when use_feat_proj=True.

    Returns dict(v[D], A[P,N], out[P,D], pooled[D], pool_ext).
    head_weight None <=> pred_head='Identity'.
    """
    if feat_proj is not None:          # use_feat_proj=True: Linear + LayerNorm on every patch row (model/deepmil.py:176-179)
        X = feat_projecter_forward(X, *feat_proj)
    A_ = vlfan_attention_logits(X, Q, gated_query, scale)
    A = torch.softmax(A_, dim=-1)
    out = A @ X  # un-normalised X, model/deepmil.py:200
    pooled, ext = query_pooling(out, query_pooling_method, pooling_params)
    v = pooled if head_weight is None else F.linear(pooled, head_weight, head_bias)
    return dict(v=v, A=A, out=out, pooled=pooled, pool_ext=ext, raw=A_)


def vlsa_logits(v: torch.Tensor, T: torch.Tensor, logit_scale: torch.Tensor):
    """Tail of VLSA.forward (model/vlsa.py:185-192). v: [M, D] (M=1 for a bag), T: [K, D].

    Returns (logits[M,K], v_hat[M,D], T_hat[K,D]).
    """
    Tn = l2_normalize(T)
    vn = l2_normalize(v)
    logits = logit_scale.exp() * vn @ Tn.t()
    return logits, vn, Tn


def logit_pooling(logits: torch.Tensor, method: str):
    """model/deepmil.py:16-37. logits: [N, C] -> (preds[1], pooled[1,C])."""
    if method[:9] in ("logit_max", "logit_top"):
        topk = 1 if method == "logit_max" else int(method.split("top")[-1])
        maxk = min(topk, logits.size(0))
        values, _ = logits.topk(maxk, 0, True, True)
        pooled = values.mean(dim=0, keepdim=True)
    elif method == "logit_mean":
        pooled = logits.mean(dim=0, keepdim=True)
    else:
        raise NotImplementedError(method)
    return pooled.argmax(dim=1), pooled


def featmil_forward(X: torch.Tensor, pooling: str) -> torch.Tensor:
    """FeatMIL.forward (model/deepmil.py:51-67). X: [N, D] -> [1, D] (mean/max) or [N, D] (identity)."""
    if pooling == "mean":
        return X.mean(dim=0, keepdim=True)
    if pooling == "max":
        return X.max(dim=0, keepdim=True).values
    return X


def vlsa_zeroshot_forward(X: torch.Tensor, T: torch.Tensor, logit_scale: torch.Tensor, pooling: str):
    """VLSA.forward with a FeatMIL encoder (model/vlsa.py:181-198 + model/deepmil.py:51-67,16-37)."""
    feats = featmil_forward(X, pooling)
    logits, vn, Tn = vlsa_logits(feats, T, logit_scale)
    if logits.shape[0] > 1:
        _, logits = logit_pooling(logits, pooling)
    return logits, vn, Tn


def adapter_forward(x: torch.Tensor, w_down: torch.Tensor, w_up: torch.Tensor) -> torch.Tensor:
    """Adapter (model/layers.py:50-62): ReLU(ReLU(x Wd^T) Wu^T), bias-free."""
    return torch.relu(torch.relu(x @ w_down.t()) @ w_up.t())


def feat_projecter_forward(x: torch.Tensor, w: torch.Tensor, b: torch.Tensor, ln_w: torch.Tensor,
                           ln_b: torch.Tensor, eps: float = 1e-5) -> torch.Tensor:
    """Feat_Projecter (model/layers.py:65-82): LayerNorm(Linear(x))."""
    return F.layer_norm(F.linear(x, w, b), (w.shape[0],), ln_w, ln_b, eps)


def deepmil_forward(X: torch.Tensor, pooling: str, pool_params=None, *, pred_head: str = "Adapter",
                    adapter: Optional[Tuple[torch.Tensor, torch.Tensor]] = None, keep_ratio: float = 0.8,
                    g_weight: Optional[torch.Tensor] = None, g_bias: Optional[torch.Tensor] = None,
                    feat_proj: Optional[Tuple[torch.Tensor, ...]] = None):
    """DeepMIL.forward (model/deepmil.py:261-292) in eval mode. X: [N, D].

    Returns dict(v[D], feat[D], raw[N] or None): ``raw`` is what ``ret_with_attn`` returns -- the raw
    scores for 'attention', the softmax weights for 'gated_attention'.
    """
    if feat_proj is not None:
        X = feat_projecter_forward(X, *feat_proj)
    ret_attn = None
    if pooling == "mean":
        feat = X.mean(dim=0)
    elif pooling == "max":
        feat = X.max(dim=0).values
    elif pooling == "attention":
        feat, raw, _ = attention_pooling(X, pool_params["w1"], pool_params["b1"], pool_params["w2"],
                                         pool_params["b2"])
        ret_attn = raw
    elif pooling == "gated_attention":
        feat, _, attn = gated_attention_pooling(X, pool_params["wa"], pool_params["ba"], pool_params["wg"],
                                                pool_params["bg"], pool_params["w2"], pool_params["b2"])
        ret_attn = attn
    else:
        raise NotImplementedError(pooling)
    if pred_head == "Adapter":
        v = keep_ratio * feat + (1 - keep_ratio) * adapter_forward(feat, adapter[0], adapter[1])
    else:
        v = F.linear(feat, g_weight, g_bias)
    return dict(v=v, feat=feat, raw=ret_attn)


def query_div_loss(Q: torch.Tensor, num_query: int, last_div: bool = True) -> torch.Tensor:
    """VLFAN.query_div_loss (model/deepmil.py:157-168)."""
    nQ = l2_normalize(Q)
    if len(Q) == num_query + 1 and last_div:
        sim = nQ[-1:] @ nQ[:-1].t()
    else:
        sim = nQ @ nQ.t()
        sim = sim[~torch.eye(len(Q), dtype=torch.bool)]
    return sim.abs().mean()


def taskres_query(prompt_features: torch.Tensor, residual: torch.Tensor, res_ratio: float = 0.5,
                  neg_prompt_features: Optional[torch.Tensor] = None,
                  neg_residual: Optional[torch.Tensor] = None) -> torch.Tensor:
    """PromptAdapter.forward, method='TaskRes' (model/prompt_learners/prompt_adapter.py:125-135)."""
    q = res_ratio * residual + prompt_features
    if neg_prompt_features is not None:
        n = neg_prompt_features if neg_residual is None else res_ratio * neg_residual + neg_prompt_features
        q = torch.cat([q, n], dim=0)
    return q


def prompt_adapter_forward(method: str, prompt_features: torch.Tensor, *, residual=None, res_ratio: float = 0.5,
                           neg_prompt_features=None, neg_residual=None, adapter=None, keep_ratio: float = 0.8,
                           fc_weight=None) -> torch.Tensor:
    """PromptAdapter.forward in eval mode for every method (model/prompt_learners/prompt_adapter.py:118-149):
    'default' -> the frozen features; 'Adapter' -> (1-keep) Adapter(f) + keep f (the negative prompt is NOT appended);
    'TaskRes' -> see taskres_query; 'FC' -> [f ; neg] W^T (bias-free Linear, dropout off)."""
    if method == "Adapter":
        return (1 - keep_ratio) * adapter_forward(prompt_features, adapter[0], adapter[1]) + keep_ratio * prompt_features
    if method == "TaskRes":
        return taskres_query(prompt_features, residual, res_ratio, neg_prompt_features, neg_residual)
    if method == "FC":
        src = prompt_features if neg_prompt_features is None else torch.cat([prompt_features, neg_prompt_features], dim=0)
        return src @ fc_weight.t()
    return prompt_features


def vlsa_vlfan_forward(X, Q, T, logit_scale, **vlfan_kw):
    """Full shipped path: VLSA.forward with a VLFAN encoder and cached text features
    (model/vlsa.py:181-198, branch 160-161).  X: [N, D].  Returns dict with logits[1,K], v_hat[1,D],
    T_hat[K,D], A[P,N], incidence[1,K] (utils/func.py:43-44)."""
    r = vlfan_forward(X, Q, **vlfan_kw)
    logits, vn, Tn = vlsa_logits(r["v"][None, :], T, logit_scale)
    r.update(logits=logits, v_hat=vn, T_hat=Tn, incidence=torch.softmax(logits, dim=-1))
    return r


def decoupled_similarity(X, Q, T, logit_scale, head_weight, head_bias, axis_softmax: str = "V"):
    """utils/model_inference.py:81-144 (calc_text_img_similarity), mean query pooling, Linear head.

    Returns (A_softmax[P,N], cottn[P,N], probs[1,K], probs2[1,K], decoupled_imp[P,K], decoupled[P,K]).
    """
    Tn = l2_normalize(T)
    ls = float(logit_scale.exp())
    _A = COATTN_SCALE * (l2_normalize(Q) @ l2_normalize(X).t())
    A_sm = torch.softmax(_A, dim=0 if axis_softmax == "L" else 1)
    r = vlfan_forward(X, Q, head_weight=head_weight, head_bias=head_bias)
    v = r["v"][None, :]
    L = v.norm(dim=-1)
    probs = torch.softmax(ls * (v / L) @ Tn.t(), dim=-1)
    enc_X = F.linear(X, head_weight, head_bias) / L
    dec = r["A"] @ (enc_X @ Tn.t())
    dec_imp = torch.softmax(ls * dec, dim=0)
    probs2 = torch.softmax(ls * dec.mean(dim=0, keepdim=True), dim=-1)
    return A_sm, r["A"], probs, probs2, dec_imp, dec


def prototype_shap(decoupled: torch.Tensor, logit_scale: float) -> torch.Tensor:
    """evaluate_prototype_shap_imp (utils/model_inference.py:23-79): exact Shapley values over the 2^P
    prototype subsets of survival risk = sum_k (K-k) softmax(ls * mean_p sim)[k]; V(empty)=1."""
    P, K = decoupled.shape
    wts = (K - torch.arange(0, K)).to(decoupled.dtype)
    V = torch.zeros(2 ** P, dtype=torch.float32)
    V[0] = 1.0
    for i in range(1, 2 ** P):
        idx = [b for b in range(P) if (i >> b) & 1]
        prob = torch.softmax(logit_scale * decoupled[idx].mean(dim=0), dim=0)
        V[i] = float((wts * prob).sum())
    fac = [math.factorial(i) for i in range(P + 1)]
    W = [fac[i] * fac[P - i - 1] / fac[P] for i in range(P)]
    shap = torch.zeros(P)
    for i in range(P):
        s = 0.0
        for j in range(2 ** P):
            if (j >> i) & 1:
                continue
            s += W[bin(j).count("1")] * (V[j + 2 ** i] - V[j])
        shap[i] = s
    return shap


# ----------------------------------------------------------------------------------------------
# Single-pass / shardable restatement (what the HIP kernels compute)
# ----------------------------------------------------------------------------------------------
def vlfan_partial(X: torch.Tensor, Qhat: torch.Tensor, gated_query: bool = False,
                  scale: float = COATTN_SCALE):
    """One shard's online-softmax partial.  X: [n, D] rows of the shard, Qhat: UNIT-norm queries.

    Returns (m[P], l[P], acc[P, D], s[P, n]) with s the scaled scores,
    m = max_n s, l = sum_n exp(s - m), acc = sum_n exp(s - m) x_n.
    An empty shard returns m=-inf, l=0, acc=0.
    """
    P = Qhat.shape[0] - (1 if gated_query else 0)
    D = X.shape[1]
    if X.shape[0] == 0:
        z = X.new_zeros
        return X.new_full((P,), -math.inf), z((P,)), z((P, D)), z((P, 0))
    r = X.norm(dim=-1).clamp_min(1e-12)
    s = (Qhat @ X.t()) / r
    if gated_query:
        s = s[:-1] - s[-1:]
    s = scale * s
    m = s.max(dim=1).values
    e = torch.exp(s - m[:, None])
    return m, e.sum(dim=1), e @ X, s


def merge_partials(ms: Sequence[torch.Tensor], ls: Sequence[torch.Tensor], accs: Sequence[torch.Tensor]):
    """Log-sum-exp merge of shard partials -> (m[P], l[P], out[P, D] = softmax-weighted rows)."""
    M = torch.stack(list(ms))
    L = torch.stack(list(ls))
    ACC = torch.stack(list(accs))
    m = M.max(dim=0).values
    w = torch.exp(M - m[None])
    w = torch.where(torch.isfinite(M), w, torch.zeros_like(w))
    l = (L * w).sum(dim=0)
    acc = (ACC * w[..., None]).sum(dim=0)
    return m, l, acc / l[:, None]


def vlfan_forward_sharded(X: torch.Tensor, Q: torch.Tensor, bounds: List[int], *, gated_query=False,
                          scale: float = COATTN_SCALE):
    """Same result as vlfan_forward(...)['out'] / ['A'] computed shard by shard.  bounds: row offsets."""
    Qh = l2_normalize(Q)
    parts = [vlfan_partial(X[a:b], Qh, gated_query, scale) for a, b in zip(bounds[:-1], bounds[1:])]
    m, l, out = merge_partials([p[0] for p in parts], [p[1] for p in parts], [p[2] for p in parts])
    s = torch.cat([p[3] for p in parts], dim=1)
    A = torch.exp(s - m[:, None]) / l[:, None]
    return dict(out=out, A=A, m=m, l=l)


# ----------------------------------------------------------------------------------------------
# Host-side losses of the training step (used ONLY by the training-step parity tests; the drop-in keeps
# the reference's own loss code, which is out of scope -- SURVEY.md section 2)
# ----------------------------------------------------------------------------------------------
def surv_ifmle(incidence: torch.Tensor, t: torch.Tensor, e: torch.Tensor, alpha: float = 0.0, eps: float = 1e-7):
    """SurvIFMLE, reduction='mean' (loss/loss_surv.py:144-169). incidence [B,K] softmax-ed, t [B] bins, e [B] events."""
    B = len(t)
    t = t.view(B, 1).long()
    c = 1 - e.view(B, 1).float()
    cif = torch.cumsum(incidence, dim=1)
    unc = -(1 - c) * torch.log(torch.gather(incidence, 1, t).clamp(min=eps))
    cen = -c * torch.log((1 - torch.gather(cif, 1, t)).clamp(min=eps))
    return ((1.0 - alpha) * (cen + unc) + alpha * unc).mean()


def surv_emd(y_hat: torch.Tensor, t: torch.Tensor, e: torch.Tensor, logit_scale, p: int = 2):
    """SurvEMD, raw_distance=True, reduction='mean' (loss/loss_surv_ext.py:43-109)."""
    B, K = y_hat.shape
    ls = logit_scale.detach() if isinstance(logit_scale, torch.Tensor) else logit_scale
    t = t.view(-1, 1).long()
    e = e.view(-1, 1).long()
    target = torch.zeros(B, K, dtype=t.dtype, device=t.device).scatter_(1, t, 1)
    for i in range(B):  # censored: every later bin is also a valid target (loss_surv_ext.py:51-54)
        loc = int(t[i, 0]) + 1
        if loc < K:
            target[i, loc:] = target[i, loc:] + (1 - e[i, 0])
    target_dist = torch.softmax((2 * target - 1) * ls, dim=-1)
    pred = (1 - e) * ((1 - target) * y_hat + target * ls) + e * y_hat
    pred_dist = torch.softmax(pred, dim=-1)
    d = torch.cumsum(pred_dist, dim=-1) - torch.cumsum(target_dist, dim=-1)
    dist = (d ** 2).sum(dim=-1) if p == 2 else d.abs().pow(p).sum(dim=-1)
    return dist.mean()


def vlsa_objective(logits: torch.Tensor, t, e, logit_scale_exp):
    """calc_objective_loss with loss_type SurvIFMLE-SurvEMD, weights 1/1 (runner/vlsa_handler.py:241-258)."""
    inc = torch.softmax(logits, dim=-1)
    return surv_ifmle(inc, t, e) + surv_emd(inc, t, e, logit_scale_exp)


# ----------------------------------------------------------------------------------------------
# Evaluation metric of the training parity test (BASELINE.json: "c-index parity")
# ----------------------------------------------------------------------------------------------
def discrete_risk(incidence: torch.Tensor) -> torch.Tensor:
    """risk score the reference ranks patients by: sum_k S_k with S = 1 - cumsum(incidence) (eval/cindex.py:36-43);
    LOW values = high risk (the reference passes -risk as the estimate)."""
    return (1.0 - torch.cumsum(incidence.double(), dim=1)).sum(dim=1)


def concordance_index(y_true: torch.Tensor, incidence: torch.Tensor, tied_tol: float = 1e-8) -> float:
    """Harrell's C as the reference evaluates it (eval/cindex.py:6-43 -> its vendored concordance_index_censored,
    eval/cindex.py:46 ff., scikit-survival's definition): a pair (i, j) is comparable when i had an event and j's time is
    later, or equal with j censored; it is concordant when i's estimate (-risk) is larger, half-counted when the two
    estimates differ by <= tied_tol.  y_true: [n, 2] = (time or time bin, event)."""
    t, e = y_true[:, 0].double(), y_true[:, 1] > 0.5
    est = -discrete_risk(incidence)
    n = t.shape[0]
    num = den = 0.0
    for i in range(n):
        if not bool(e[i]):
            continue
        comp = (t > t[i]) | ((t == t[i]) & ~e)
        comp[i] = False
        if not bool(comp.any()):
            continue
        d = est[i] - est[comp]
        den += float(comp.sum())
        num += float((d > tied_tol).sum()) + 0.5 * float((d.abs() <= tied_tol).sum())
    if den == 0:
        raise ValueError("no comparable pairs")
    return num / den
