"""Functional API over the C ABI: allocation, stream plumbing and autograd glue.  All arithmetic on the
N-sized data runs in libvlsa_hip.so; torch is used for device memory, streams and autograd bookkeeping.

Reference ops replaced (paths relative to the upstream repo): model/deepmil.py:187-200 (cross attention),
133-150 + 204 (query pooling, visual adapter), model/vlsa.py:185-192 (cosine logits).
"""
from __future__ import annotations

import ctypes
import os
from dataclasses import dataclass
from typing import Optional, Tuple

import torch

from . import _native as nat
from ._native import VlsaNativeError

COATTN_SCALE = 100.0  # exp(coattn_logit_scale), model/deepmil.py:120-126


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)
_cur_device = getattr(torch._C, "_cuda_getDevice", None)


def _stream():
    """The current HIP stream of the current device as a C pointer.  The raw accessors (what torch.cuda.current_stream() wraps) save
    ~7 us of Python object construction per call -- this runs several times per bag in the bag-by-bag loops."""
    if _raw_stream is not None and _cur_device is not None:
        return ctypes.c_void_p(_raw_stream(_cur_device()))
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _p(t: Optional[torch.Tensor]):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def _need_gpu(*tensors):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise VlsaNativeError(
                "vlsa_amd runs on MI355X only: got a CPU tensor (there is no CPU fallback; the CPU oracle under "
                "oracle/ is test infrastructure)")


def _f32c(t: torch.Tensor) -> torch.Tensor:
    t = t.detach()
    if t.dtype != torch.float32:
        t = t.float()
    return t.contiguous()


_BAG_DTYPES = (torch.bfloat16, torch.float32)


def _bag2d(X: torch.Tensor) -> torch.Tensor:
    """[1,N,D] or [N,D] -> [N,D] view with unit inner stride and 16-byte aligned rows (copy only if needed)."""
    # the common case first (a list of 64 slide-sized bags pays this per bag, and the launch itself is ~0.6 us per bag): a contiguous
    # [N, D] bf16 / fp32 tensor whose rows are a multiple of 16 bytes, at a 16-byte aligned address
    if (X.dim() == 2 and X.dtype in _BAG_DTYPES and X.is_contiguous() and (X.shape[1] * X.element_size()) % 16 == 0
            and X.data_ptr() % 16 == 0):
        return X
    if X.dim() == 3:
        if X.shape[0] != 1:
            raise AssertionError("X.shape[0] must be 1 (one bag per call; model/deepmil.py:175)")
        X = X[0]
    if X.dim() != 2:
        raise ValueError(f"expected a [N, D] or [1, N, D] bag, got {tuple(X.shape)}")
    if X.dtype not in (torch.float32, torch.bfloat16):
        X = X.float()
    esz = X.element_size()
    if X.shape[0] > 0 and (X.stride(1) != 1 or (X.stride(0) * esz) % 16 != 0 or X.data_ptr() % 16 != 0
                           or X.stride(0) < X.shape[1]):
        X = X.contiguous()
    return X


def _no_bag_grad(*bags):
    """The HIP aggregation kernels produce gradients for the queries / scores only (the reference's bags carry none:
    SURVEY.md a14).  A bag that requires grad (e.g. the output of a trainable Feat_Projecter) must not be silently
    detached: the modules route that case to device torch ops; calling the functional API with it is an error."""
    if torch.is_grad_enabled():
        for x in bags:
            if x is not None and x.requires_grad:
                raise VlsaNativeError("the bag requires grad, but the HIP aggregation has no dX: use the module API "
                                      "(VLFAN / DeepMIL route a differentiable bag through torch ops) or detach the bag")


@dataclass
class PreparedQueries:
    """Device block produced by vlsa_prepare_queries (unit queries, effective queries, bf16 split)."""
    buf: torch.Tensor
    nq: int
    P: int
    D: int
    gated: bool

    def _view(self, off: int, rows: int) -> torch.Tensor:
        return self.buf[off:off + rows * self.D * 4].view(torch.float32).view(rows, self.D)

    @property
    def qeff(self) -> torch.Tensor:   # [P, D] effective queries (q^_p - q^_gate)
        return self._view(0, 16)[: self.P]

    @property
    def qhat(self) -> torch.Tensor:   # [nq, D] unit queries
        off = 16 * self.D * 4 + 3 * 16 * self.D * 2
        return self._view(off, 17)[: self.nq]

    @property
    def qnorm(self) -> torch.Tensor:  # [nq] max(||q||, 1e-12)
        off = 16 * self.D * 4 + 3 * 16 * self.D * 2 + 17 * self.D * 4
        return self.buf[off:off + 128].view(torch.float32)[: self.nq]


def prepare_queries(Q: torch.Tensor, gated: bool = False, coattn_scale: float = COATTN_SCALE) -> PreparedQueries:
    _need_gpu(Q)
    lib = nat.load()
    Q = _f32c(Q)
    nq, D = Q.shape
    P = nq - 1 if gated else nq
    if not (1 <= P <= nat.MAX_P):
        raise ValueError(f"number of queries P={P} outside [1, {nat.MAX_P}]")
    buf = torch.empty(lib.vlsa_qprep_bytes(D), dtype=torch.uint8, device=Q.device)
    nat.check(lib.vlsa_prepare_queries(_p(Q), nq, D, int(gated), float(coattn_scale), _p(buf), _stream()),
              "vlsa_prepare_queries")
    return PreparedQueries(buf, nq, P, D, gated)


def num_partials(N: int) -> int:
    return int(nat.load().vlsa_num_partials(N))


def vlfan_partial(X: torch.Tensor, qp: PreparedQueries, kernel: int = nat.KERNEL_AUTO, want_scores: bool = False):
    """One streaming pass over a shard's rows -> per-workgroup partials (pm[G,16], pl[G,16], pacc[G,P,D])
    and, if asked, the log2-domain scores [P, N]."""
    _need_gpu(X)
    lib = nat.load()
    X = _bag2d(X)
    N, D = X.shape
    if D != qp.D:
        raise ValueError(f"feature dim mismatch: bag {D}, queries {qp.D}")
    G = num_partials(N)
    dev = X.device
    pm = torch.empty(G, nat.P_STRIDE, dtype=torch.float32, device=dev)
    pl = torch.empty(G, nat.P_STRIDE, dtype=torch.float32, device=dev)
    pacc = torch.empty(G, qp.P, D, dtype=torch.float32, device=dev)
    scores = torch.empty(qp.P, N, dtype=torch.float32, device=dev) if want_scores else None
    if N == 0:
        pm.fill_(float("-inf"))
        pl.zero_()
        pacc.zero_()
        return pm, pl, pacc, scores
    dt = nat.DT_F32 if X.dtype == torch.float32 else nat.DT_BF16
    nat.check(lib.vlsa_vlfan_partial(_p(X), dt, N, X.stride(0), D, _p(qp.buf), qp.P, kernel,
                                     _p(pm), _p(pl), _p(pacc), _p(scores), _stream()), "vlsa_vlfan_partial")
    return pm, pl, pacc, scores


def vlfan_merge(pm: torch.Tensor, pl: torch.Tensor, pacc: torch.Tensor, normalise: bool = True):
    """Log-sum-exp merge of G partials -> (m2[16], l[16], out[P, D])."""
    _need_gpu(pm, pl, pacc)
    lib = nat.load()
    G, P, D = pacc.shape
    dev = pacc.device
    m2 = torch.empty(nat.P_STRIDE, dtype=torch.float32, device=dev)
    l = torch.empty(nat.P_STRIDE, dtype=torch.float32, device=dev)
    out = torch.empty(P, D, dtype=torch.float32, device=dev)
    nat.check(lib.vlsa_vlfan_merge(_p(pm), _p(pl), _p(pacc), G, P, D, int(normalise), _p(m2), _p(l), _p(out),
                                   _stream()), "vlsa_vlfan_merge")
    return m2, l, out


def attn_normalise(scores: torch.Tensor, m2: torch.Tensor, l: torch.Tensor) -> torch.Tensor:
    _need_gpu(scores)
    lib = nat.load()
    P, N = scores.shape
    A = torch.empty_like(scores)
    nat.check(lib.vlsa_attn_normalise(_p(scores), P, N, _p(m2), _p(l), _p(A), _stream()), "vlsa_attn_normalise")
    return A


def normalize_rows(T: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    """F.normalize(T, dim=-1) for a small [rows, D] matrix; also returns the clamped norms."""
    _need_gpu(T)
    lib = nat.load()
    T = _f32c(T)
    rows, D = T.shape
    out = torch.empty_like(T)
    norms = torch.empty(rows, dtype=torch.float32, device=T.device)
    nat.check(lib.vlsa_normalize_rows(_p(T), rows, D, _p(out), _p(norms), _stream()), "vlsa_normalize_rows")
    return out, norms


_POOL_CODES = {"mean": nat.POOL_MEAN, "max": nat.POOL_MAX, "weight": nat.POOL_WEIGHT, "given": nat.POOL_GIVEN}


def head_forward(rows: torch.Tensor, pool: str, pool_w: Optional[torch.Tensor], W: Optional[torch.Tensor],
                 b: Optional[torch.Tensor], That: torch.Tensor, logit_scale: torch.Tensor,
                 want_incidence: bool = False):
    """Query pooling + visual adapter + cosine logits (model/deepmil.py:203-204, model/vlsa.py:188-192).

    rows: [P, D] aggregated rows (or [1, D] with pool='given'); That: [K, D] unit-norm text features;
    logit_scale: the 0-dim parameter (pre-exp), read on device.
    """
    _need_gpu(rows, That, logit_scale)
    lib = nat.load()
    rows = _f32c(rows)
    P, D = rows.shape
    K = That.shape[0]
    dev = rows.device
    That = _f32c(That)
    W_ = _f32c(W) if W is not None else None
    b_ = _f32c(b) if b is not None else None
    pw = _f32c(pool_w).reshape(-1) if pool_w is not None else None
    ls = _f32c(logit_scale).reshape(1)
    ws = torch.zeros(lib.vlsa_head_workspace_bytes(D), dtype=torch.uint8, device=dev)
    f = lambda n: torch.empty(n, dtype=torch.float32, device=dev)  # noqa: E731
    pooled, v, vhat, vnorm, logits = f(D), f(D), f(D), f(1), f(K)
    inc = f(K) if want_incidence else None
    nat.check(lib.vlsa_head_forward(_p(rows), P, D, _POOL_CODES[pool], _p(pw), _p(W_), _p(b_), _p(That), K, _p(ls),
                                    _p(ws), _p(pooled), _p(v), _p(vhat), _p(vnorm), _p(logits), _p(inc), _stream()),
              "vlsa_head_forward")
    return dict(pooled=pooled, v=v, vhat=vhat, vnorm=vnorm, logits=logits, incidence=inc)


class HeadTickets:
    """Ticket counters of the batched training head (one int32 per bag, handed back zeroed by the kernel), owned by whoever
    runs the head -- ``VLSA`` keeps one per model.  One buffer per (device, B, stream): two streams must not share a counter,
    and two heads of different models never do (no process-global state: SURVEY.md 8(b))."""

    def __init__(self):
        self._bufs = {}

    def get(self, device, B: int, stream_id: int) -> torch.Tensor:
        key = (device, B, stream_id)
        t = self._bufs.get(key)
        if t is None:
            if len(self._bufs) > 64:
                self._bufs.clear()
            t = self._bufs[key] = torch.zeros(B, dtype=torch.int32, device=device)
        return t


class _HeadTrainFn(torch.autograd.Function):
    """(logits [B, K], v^ [B, D], T^ [K, D]) from the aggregated rows [B, P, D]: mean query pooling, Linear / identity adapter,
    normalisation and cosine logits (model/deepmil.py:203-204, model/vlsa.py:188-192) -- ONE host call and three launches forward
    (vlsa_head_forward_batch_text; one bag: vlsa_normalize_rows + vlsa_head_forward_batch), two backward (vlsa_head_backward_batch), in place of ~40 autograd kernels:
    the optimizer step is bound by its number of dependent launches."""

    @staticmethod
    def forward(ctx, rows, W, b, T, logit_scale, tickets):
        ctx.set_materialize_grads(False)      # the handler uses the logits only: no zero-filled g_vhat / g_That per backward
        lib, s = nat.load(), _stream()
        rows = _f32c(rows)
        B, P, D = rows.shape
        dev = rows.device
        Tc = _f32c(T)
        K = Tc.shape[0]
        Wc = None if W is None else _f32c(W)
        bc = None if b is None else _f32c(b)
        ls = _f32c(logit_scale).reshape(1)
        f = lambda *sh: torch.empty(*sh, dtype=torch.float32, device=dev)  # noqa: E731
        pooled, v, vhat, vnorm, logits = f(B, D), f(B, D), f(B, D), f(B), f(B, K)
        if B > 1:
            # (the text normalisation rides in the pooling launch: three launches for the head, no tickets)
            That, tnorm = f(K, D), f(K)
            nat.check(lib.vlsa_head_forward_batch_text(_p(rows), B, P, D, nat.POOL_MEAN, None, _p(Wc), _p(bc), _p(Tc), K, _p(ls), _p(That), _p(tnorm),
                                                       _p(pooled), _p(v), _p(vhat), _p(vnorm), _p(logits), None, s), "vlsa_head_forward_batch_text")
        else:
            That, tnorm = normalize_rows(Tc)
            # the kernel hands the tickets back zeroed; without an owner: a fresh zeroed buffer (one memset launch more)
            tk = torch.zeros(B, dtype=torch.int32, device=dev) if tickets is None else tickets.get(dev, B, s.value or 0)
            nat.check(lib.vlsa_head_forward_batch(_p(rows), B, P, D, nat.POOL_MEAN, None, _p(Wc), _p(bc), _p(That), K, _p(ls), _p(tk),
                                                  _p(pooled), _p(v), _p(vhat), _p(vnorm), _p(logits), None, s), "vlsa_head_forward_batch")
        ctx.save_for_backward(pooled, vhat, vnorm, That, tnorm, logits, ls, *([Wc] if Wc is not None else []))
        ctx.meta = (B, P, D, K, Wc is not None, b is not None, tuple(logit_scale.shape))
        return logits, vhat, That

    @staticmethod
    def backward(ctx, dlogits, g_vhat, g_That):
        lib, s = nat.load(), _stream()
        B, P, D, K, has_w, has_b, ls_shape = ctx.meta
        pooled, vhat, vnorm, That, tnorm, logits, ls = ctx.saved_tensors[:7]
        Wc = ctx.saved_tensors[7] if has_w else None
        dev = pooled.device
        f = lambda *sh: torch.empty(*sh, dtype=torch.float32, device=dev)  # noqa: E731
        dl = _f32c(dlogits) if dlogits is not None else torch.zeros(B, K, dtype=torch.float32, device=dev)
        gv = None if g_vhat is None else _f32c(g_vhat)
        gt = None if g_That is None else _f32c(g_That)
        ws, drows, dT, dls = f(B * D + B), f(B, P, D), f(K, D), f(1)
        dW, db = (f(D, D), f(D)) if has_w else (None, None)
        nat.check(lib.vlsa_head_backward_batch(_p(dl), _p(gv), _p(gt), _p(pooled), _p(vhat), _p(vnorm), _p(That), _p(tnorm), _p(logits),
                                               _p(Wc), _p(ls), B, P, D, K, _p(ws), _p(drows), _p(dW), _p(db), _p(dT), _p(dls), s),
                  "vlsa_head_backward_batch")
        return drows, dW, (db if has_b else None), dT, dls.reshape(ls_shape), None


def head_train(rows: torch.Tensor, W, b, T: torch.Tensor, logit_scale: torch.Tensor, tickets: Optional[HeadTickets] = None):
    """Differentiable (logits, unit image features, unit text features) of a batch of aggregated rows [B, P, 512] -- see
    ``_HeadTrainFn``; W / b: the Linear adapter (None: identity); tickets: the caller's ``HeadTickets`` (None: a zeroed
    buffer is allocated per call)."""
    _need_gpu(rows, T, logit_scale)
    return _HeadTrainFn.apply(rows, W, b, T, logit_scale, tickets)


def vlfan_aggregate(X: torch.Tensor, Q: torch.Tensor, gated: bool = False, coattn_scale: float = COATTN_SCALE,
                    kernel: int = nat.KERNEL_AUTO, want_attn: bool = False):
    """Inference-only cross-attention aggregation: out[P, D] = softmax_N(100 cos(Q, X)) @ X, plus A[P, N]."""
    qp = prepare_queries(Q, gated, coattn_scale)
    pm, pl, pacc, scores = vlfan_partial(X, qp, kernel, want_scores=want_attn)
    m2, l, out = vlfan_merge(pm, pl, pacc, normalise=True)
    A = attn_normalise(scores, m2, l) if want_attn else None
    return out, A, (m2, l, qp)


def debug_probe(which: int, device="cuda") -> torch.Tensor:
    lib = nat.load()
    out = torch.zeros(64, 4, dtype=torch.float32, device=device)
    nat.check(lib.vlsa_debug_probe(which, _p(out), out.numel() * 4, _stream()), "vlsa_debug_probe")
    return out


class SlideTrainPlan:
    """Scratch and the prepared (queries, text features) block of the per-bag TRAINING path: ``slide_train`` is what
    ``VLSA.forward`` runs for one bag when a gradient is needed and the encoder is a VLFAN with mean query pooling -- the loop
    shape of the reference's handler (runner/vlsa_handler.py:267-289: one ``net(X)`` per bag, predictions concatenated, ONE
    backward).  That loop is bound by the host side of its autograd nodes, so a bag costs ONE autograd node, one Python -> C
    crossing and one allocation each way (``vlsa_vlfan_forward_bag`` / ``vlsa_vlfan_backward_bag``).  The prepared block
    (normalised / split queries, unit text features) is rebuilt only when the query or text-feature TENSOR changes: all bags of
    a step share both.  Everything is stream-ordered on the caller's stream; any bag size (scratch is sized for 256 partials)."""

    def __init__(self, D: int, P: int, K: int, device, gated: bool, identity_head: bool, coattn_scale: float):
        lib = nat.load()
        self.lib, self.D, self.P, self.K = lib, int(D), int(P), int(K)
        self.nq, self.gated, self.identity_head, self.scale = int(P) + (1 if gated else 0), bool(gated), bool(identity_head), float(coattn_scale)
        G = int(lib.vlsa_num_partials(1 << 40))
        f = lambda *sh: torch.empty(*sh, dtype=torch.float32, device=device)  # noqa: E731
        self.qprep = torch.empty(lib.vlsa_qprep_bytes(D), dtype=torch.uint8, device=device)
        self.That, self.tnorm = f(K, D), f(K)
        self.pm, self.pl, self.pacc = f(G + 1, nat.P_STRIDE), f(G + 1, nat.P_STRIDE), f(G, P, D)
        self.head_ws = torch.zeros(lib.vlsa_head_workspace_bytes(D), dtype=torch.uint8, device=device)
        self.bwd_prep = torch.empty(lib.vlsa_bwd_prep_bytes(D), dtype=torch.uint8, device=device)
        self.hws = f(D + 4)
        self._c = {k: getattr(self, k).data_ptr() for k in ("qprep", "That", "tnorm", "pm", "pl", "pacc", "head_ws", "bwd_prep", "hws")}
        self.gen, self._src, self._That_out = 0, None, None
        # float offsets of the per-call record the backward reads: out | m2 | l | pooled | v | vnorm (logits and v^ are the node's
        # outputs: tensors of their own, so that in-place edits by the caller stay legal)
        o, self.off = 0, {}
        for name, n in (("out", P * D), ("m2", 16), ("l", 16), ("pooled", D), ("v", D), ("vnorm", 4)):
            self.off[name] = o
            o += n
        self.rec_floats = o
        # ... and of the per-call gradient block: dQ | dW | db | dT | dls | drows | dE.  Its head (up to and including dls[0]) has
        # the layout of the step's FLAT parameter tensor Q | W | b | T | logit_scale (``step_params``): a bag hands ONE gradient
        # tensor to autograd instead of five, and the five parameter gradients are split off once per step.
        o, self.goff = 0, {}
        for name, n in (("dQ", self.nq * D), ("dW", 0 if identity_head else D * D), ("db", 0 if identity_head else D), ("dT", K * D), ("dls", 4),
                        ("drows", P * D), ("dE", P * D)):
            self.goff[name] = o
            o += n
        self.grad_floats = o
        self.flat_floats = self.goff["dls"] + 1
        self._flat, self._flat_src, self._zero_b = None, None, None
        self._conv = {}

    def f32c(self, slot: str, t: torch.Tensor) -> torch.Tensor:
        """fp32-contiguous view of a step parameter, converted ONCE per (tensor object, in-place version, grad mode): a fresh
        ``.float().contiguous()`` per bag would defeat the identity check of ``step_params`` (one cat + one preparation per bag
        instead of per step).  The conversion is differentiable; a backward pass through it frees its graph, so the memo goes
        with it (as for the flat tensor)."""
        if t.dtype == torch.float32 and t.is_contiguous():
            return t
        c = self._conv.get(slot)
        if c is None or c[0] is not t or c[1] != t._version or c[3] != torch.is_grad_enabled():
            conv = t.float().contiguous()
            if conv.requires_grad and conv.grad_fn is not None:
                conv.register_hook(lambda *_: self._conv.pop(slot, None))
            c = self._conv[slot] = (t, t._version, conv, torch.is_grad_enabled())
        return c[2]

    def step_params(self, Q, W, b, T, logit_scale) -> torch.Tensor:
        """Q | W | b | T | logit_scale as ONE fp32 tensor (a ``torch.cat``: differentiable), shared -- with its graph -- by every bag
        that sees the same tensor objects at the same in-place versions; a backward pass through it frees the graph, the next
        call rebuilds.  (W None: identity adapter, no W / b part; b None: a zero part.)"""
        vers = (torch.is_grad_enabled(), Q._version, -1 if W is None else W._version, -1 if b is None else b._version, T._version,
                logit_scale._version)
        src = self._flat_src
        if (self._flat is not None and src[0] is Q and src[1] is W and src[2] is b and src[3] is T and src[4] is logit_scale
                and src[5] == vers):
            return self._flat
        parts = [Q.reshape(-1)]
        if W is not None:
            if b is None:
                if self._zero_b is None:
                    self._zero_b = torch.zeros(self.D, dtype=torch.float32, device=W.device)
                b_part = self._zero_b
            else:
                b_part = b
            parts += [W.reshape(-1), b_part]
        parts += [T.reshape(-1), logit_scale.reshape(1)]
        flat = torch.cat(parts)
        if flat.requires_grad and flat.grad_fn is not None:
            flat.register_hook(self._drop_flat)
        self._flat, self._flat_src = flat, (Q, W, b, T, logit_scale, vers)
        return flat

    def _drop_flat(self, *_):
        self._flat = None

    def prepared_for(self, flat: torch.Tensor) -> bool:
        src = self._src
        return src is not None and src[0] is flat and src[1] == flat._version

    def mark_prepared(self, flat):
        self._src = (flat, flat._version)               # the tensor is kept alive: its identity is the key
        self.gen += 1
        self._That_out = None

    def unit_text(self):
        if self._That_out is None:
            self._That_out = self.That.clone()          # one copy per preparation, shared (read-only) by every bag's output
        return self._That_out.detach()


class _SlideTrainFn(torch.autograd.Function):
    """One bag: X2 [N, 512] and the step's flat parameter tensor (``SlideTrainPlan.step_params``) -> logits, unit features."""

    @staticmethod
    def _params(plan, flat):
        """C pointers of Q, W, b, T, logit_scale inside the flat tensor"""
        g, fb = plan.goff, flat.data_ptr()
        ident = plan.identity_head
        return (fb, None if ident else fb + 4 * g["dW"], None if ident else fb + 4 * g["db"], fb + 4 * g["dT"], fb + 4 * g["dls"])

    @staticmethod
    def forward(ctx, X2, flat, plan):
        ctx.set_materialize_grads(False)      # unused outputs (v^, T^) reach backward as None, not as zero-filled tensors
        lib, s, c, off = plan.lib, _stream(), plan._c, plan.off
        N = X2.shape[0]
        dev = X2.device
        rec = torch.empty(plan.rec_floats, dtype=torch.float32, device=dev)
        logits = torch.empty(1, plan.K, dtype=torch.float32, device=dev)
        vhat = torch.empty(1, plan.D, dtype=torch.float32, device=dev)
        base = rec.data_ptr()
        fresh = not plan.prepared_for(flat)
        G = int(lib.vlsa_num_partials(N))
        pQ, pW, pb, pT, pls = _SlideTrainFn._params(plan, flat)
        at = lambda name: base + 4 * off[name]  # noqa: E731
        nat.check(lib.vlsa_vlfan_forward_bag(_p(X2), _dt(X2), N, X2.stride(0), plan.D, pQ if fresh else None, plan.nq, int(plan.gated),
                                             plan.scale, pT, plan.K, pls, nat.POOL_MEAN, None, pW, pb, nat.KERNEL_AUTO,
                                             c["qprep"], c["That"], c["tnorm"], c["pm"], c["pl"], c["pacc"], G, at("m2"), at("l"),
                                             at("out"), None, None, c["head_ws"], at("pooled"), at("v"), vhat.data_ptr(), at("vnorm"),
                                             logits.data_ptr(), None, s), "vlsa_vlfan_forward_bag")
        if fresh:
            plan.mark_prepared(flat)
        ctx.plan, ctx.gen = plan, plan.gen
        ctx.save_for_backward(X2, flat, rec, logits, vhat)
        return logits, vhat, plan.unit_text()

    @staticmethod
    def backward(ctx, dlogits, g_vhat, g_That):
        plan = ctx.plan
        lib, s, c, off, goff = plan.lib, _stream(), plan._c, plan.off, plan.goff
        X2, flat, rec, logits, vhat = ctx.saved_tensors
        N, dev, K, D, P, nq = X2.shape[0], X2.device, plan.K, plan.D, plan.P, plan.nq
        pQ, pW, pb, pT, pls = _SlideTrainFn._params(plan, flat)
        qprep, That, tnorm = c["qprep"], c["That"], c["tnorm"]
        keep = None
        if plan.gen != ctx.gen:
            # the plan's prepared block has moved on (another forward with new queries / text features ran before this backward):
            # rebuild this bag's block from the saved flat tensor
            keep = (torch.empty(lib.vlsa_qprep_bytes(D), dtype=torch.uint8, device=dev), torch.empty(K, D, dtype=torch.float32, device=dev),
                    torch.empty(K, dtype=torch.float32, device=dev))
            nat.check(lib.vlsa_prepare_queries_and_text(pQ, nq, D, int(plan.gated), plan.scale, _p(keep[0]), pT, K, _p(keep[1]),
                                                        _p(keep[2]), s), "vlsa_prepare_queries_and_text")
            qprep, That, tnorm = (t.data_ptr() for t in keep)
        gb = torch.empty(plan.grad_floats, dtype=torch.float32, device=dev)
        gbase, base = gb.data_ptr(), rec.data_ptr()
        at = lambda name: base + 4 * off[name]  # noqa: E731
        gat = lambda name: gbase + 4 * goff[name]  # noqa: E731
        dl = _f32c(dlogits) if dlogits is not None else torch.zeros(1, K, dtype=torch.float32, device=dev)
        gv = None if g_vhat is None else _f32c(g_vhat)
        gt = None if g_That is None else _f32c(g_That)
        G = int(lib.vlsa_num_partials(N))
        has_w = pW is not None
        nat.check(lib.vlsa_vlfan_backward_bag(_p(X2), _dt(X2), N, X2.stride(0), D, qprep, nq, int(plan.gated), plan.scale, _p(dl), _p(gv), _p(gt),
                                              at("pooled"), vhat.data_ptr(), at("vnorm"), That, tnorm, logits.data_ptr(), pW, pls, at("out"),
                                              at("m2"), at("l"), K, c["hws"], gat("drows"), gat("dW") if has_w else None,
                                              gat("db") if has_w else None, gat("dT"), gat("dls"), c["bwd_prep"], c["pm"], c["pl"], c["pacc"], G,
                                              gat("dE"), gat("dQ"), s), "vlsa_vlfan_backward_bag")
        dX = None
        if ctx.needs_input_grad[0]:      # the bag is the output of a trainable Feat_Projecter (fp32 [N, 512]): dL/dX of the aggregation
            dX = torch.empty(N, D, dtype=torch.float32, device=dev)
            tkeep, p_desc, p_dx, _, p_ts, n_tiles, _ = _row_tables([X2], 64, extra=[dX])
            delta = torch.empty(1, nat.P_STRIDE, dtype=torch.float32, device=dev)
            nat.check(lib.vlsa_vlfan_backward_dx(p_desc, p_dx, 1, D, qprep, P, plan.scale, p_ts, n_tiles, gat("drows"), at("out"), at("m2"),
                                                 at("l"), _p(delta), s), "vlsa_vlfan_backward_dx")
        return dX, gb[:plan.flat_floats], None


def slide_train(X2: torch.Tensor, Q: torch.Tensor, W, b, T: torch.Tensor, logit_scale: torch.Tensor, plan: SlideTrainPlan):
    """Differentiable (logits [1, K], unit image features [1, D], unit text features [K, D]) of ONE bag [N, 512] (bf16 / fp32):
    cross attention with the queries Q, mean query pooling, Linear / identity adapter, cosine logits (model/deepmil.py:187-204,
    model/vlsa.py:188-192).  Gradients: Q, W, b, T, logit_scale -- and the bag itself when it is an fp32 tensor that requires
    grad (the output of a trainable Feat_Projecter: vlsa_vlfan_backward_dx).  All tensors fp32 contiguous on
    the bag's device (Q [nq, 512], T [K, 512], logit_scale 0-dim); see ``SlideTrainPlan``."""
    _need_gpu(X2, Q, T, logit_scale)
    return _SlideTrainFn.apply(X2, plan.step_params(Q, W, b, T, logit_scale), plan)


_PARTIALS = {}


def _partials_of(N: int) -> int:
    """vlsa_num_partials(N), kept per size (one ctypes call per new size instead of one per bag)"""
    g = _PARTIALS.get(N)
    if g is None:
        if len(_PARTIALS) > 1 << 16:
            _PARTIALS.clear()
        g = _PARTIALS[N] = int(nat.load().vlsa_num_partials(N))
    return g


class VlfanInferencePlan:
    """Pre-allocated buffers + raw C-ABI calls for the fused inference forward of one bag shape.

    One ``run(X, Q, T, ...)`` = what ``VLSA.forward`` does per bag in eval mode with cached text features
    (model/vlsa.py:181-198 via the branch at 160-161): query normalisation, text normalisation, the
    streaming aggregation, the partial merge and the incidence head.  No torch allocation or sync inside,
    so a sequence of runs can be captured in a hipGraph.
    """

    def __init__(self, N: int, D: int, P: int, K: int, device, gated: bool = False, pool: str = "mean",
                 identity_head: bool = False, kernel: int = nat.KERNEL_AUTO, want_attn: bool = False,
                 coattn_scale: float = COATTN_SCALE):
        lib = nat.load()
        self.lib, self.N, self.D, self.P, self.K = lib, N, D, P, K
        self.gated, self.pool, self.kernel, self.scale = gated, _POOL_CODES[pool], kernel, float(coattn_scale)
        self.identity_head = identity_head
        # N = None (round 6): a plan for bags of ANY size -- scratch for the largest partial count (256 records of P x 512: 6.3 MB at
        # P = 12), N and G taken from the bag at every run.  The reference's slides all differ in size (2k - 12k patches at TCGA): a plan
        # per size made `VLSA`'s plan / hot-call caches thrash in exactly the loop they exist for.  (Not with `want_attn`: [P, N] buffers.)
        if N is None and want_attn:
            raise ValueError("an any-size inference plan cannot hold attention weights: give N")
        self.G = num_partials(N if N is not None else (1 << 40))
        f = lambda *s: torch.empty(*s, dtype=torch.float32, device=device)  # noqa: E731
        self.qprep = torch.empty(lib.vlsa_qprep_bytes(D), dtype=torch.uint8, device=device)
        self.pm, self.pl, self.pacc = f(self.G, nat.P_STRIDE), f(self.G, nat.P_STRIDE), f(self.G, P, D)
        self.m2, self.l, self.out = f(nat.P_STRIDE), f(nat.P_STRIDE), f(P, D)
        self.That, self.tnorm = f(K, D), f(K)
        self.ws = torch.zeros(lib.vlsa_head_workspace_bytes(D), dtype=torch.uint8, device=device)
        self.pooled, self.v, self.vhat, self.vnorm = f(D), f(D), f(D), f(1)
        self.logits, self.incidence = f(K), f(K)
        self.scores = f(P, N) if want_attn else None
        self.A = f(P, N) if want_attn else None
        self._c = {k: _p(getattr(self, k)) for k in ("qprep", "pm", "pl", "pacc", "m2", "l", "out", "That", "tnorm", "ws",
                                                     "pooled", "v", "vhat", "vnorm", "logits", "incidence", "scores", "A")}

    def run(self, X: torch.Tensor, Q: torch.Tensor, T: torch.Tensor, logit_scale: torch.Tensor,
            W: Optional[torch.Tensor] = None, b: Optional[torch.Tensor] = None,
            pool_w: Optional[torch.Tensor] = None, outs: Optional[dict] = None, params_key=None, query_pool_module=None):
        """params_key: a hashable that changes whenever Q or T change (parameter versions); when it equals the previous
        call's, the query / text preparation launch is skipped and the plan's prepared block is reused (outs['That'], if
        given, is then filled by a copy of the cached unit text features).
        X [N, D] (fp32/bf16, unit inner stride), Q [nq, D] fp32, T [K, D] fp32 raw text features,
        logit_scale 0-dim fp32 -- all contiguous device tensors (not checked here: hot path).
        outs: optional {'logits': [K], 'vhat': [D], 'That': [K, D]} fp32 tensors the kernels write INSTEAD of the plan's own
        buffers (a caller that must hand out fresh tensors per bag saves three copy kernels)."""
        lib, s, k = self.lib, _stream(), self._c
        reuse = params_key is not None and params_key == getattr(self, "_params_key", None)
        self._params_key = None          # set again once the call that prepares this key has returned OK
        N_, G_ = (self.N, self.G) if self.N is not None else (X.shape[0], _partials_of(X.shape[0]))
        if outs:
            k = dict(k)
            for name, t in outs.items():
                if name == "That":      # the prepared text features live in the plan so that later calls can reuse them
                    continue
                k[name] = _p(t)
        nq = self.P + 1 if self.gated else self.P
        dt = nat.DT_F32 if X.dtype == torch.float32 else nat.DT_BF16
        # one Python -> C crossing for the five launches (this path is host-bound: the handler calls it bag by bag)
        nat.check(lib.vlsa_vlfan_forward_bag(_p(X), dt, N_, X.stride(0), self.D, None if reuse else _p(Q), nq, int(self.gated), self.scale, _p(T),
                                             self.K, _p(logit_scale), -1 if query_pool_module is not None else self.pool, _p(pool_w),
                                             None if self.identity_head else _p(W), None if self.identity_head else _p(b),
                                             self.kernel, k["qprep"], k["That"], k["tnorm"], k["pm"], k["pl"], k["pacc"], G_,
                                             k["m2"], k["l"], k["out"], k["scores"], k["A"], k["ws"], k["pooled"], k["v"], k["vhat"],
                                             k["vnorm"], k["logits"], k["incidence"], s), "vlsa_vlfan_forward_bag")
        self._params_key = params_key
        if query_pool_module is not None:   # (gated-)attention pooling over the P rows, then the head on the pooled row
            pooled, self.pool_scores = query_pool_attention(self.out[None], query_pool_module)
            nat.check(lib.vlsa_head_forward(_p(pooled), 1, self.D, nat.POOL_GIVEN, None, None if self.identity_head else _p(W),
                                            None if self.identity_head else _p(b), k["That"], self.K, _p(logit_scale), k["ws"],
                                            k["pooled"], k["v"], k["vhat"], k["vnorm"], k["logits"], k["incidence"], s), "vlsa_head_forward")
        if outs and "That" in outs:
            if not reuse or getattr(self, "_That_out", None) is None:
                self._That_out = self.That.clone()   # one copy per parameter version, handed out (read-only) to every bag
            outs["That"] = self._That_out
        return outs["logits"] if outs and "logits" in outs else self.logits

    def hot_call(self, T, logit_scale, W, b, pool_w, That_out):
        """The per-bag call with EVERYTHING but the bag frozen: returns ``fn(X [1, N, D]) -> (logits [1, K], vhat [1, D], That)`` that
        issues ``vlsa_vlfan_forward_bag`` with a pre-built argument list (prepared queries / text features reused: Q = NULL) -- the
        reference handler's loop calls the model once per slide (runner/vlsa_handler.py:322-330) and that call is bound by its host side
        (tools/prof_single_slide.py: 28.5 us per call for ANY bag size in round 5, ~15 of it Python around the C call).  The caller
        (``VLSA._fused_vlfan``) keeps the closure only as long as the model state it was built under holds; the tensors it reads are
        kept alive here."""
        lib, k = self.lib, self._c
        keep = (T, logit_scale, W, b, pool_w, That_out)
        fn_c = lib.vlsa_vlfan_forward_bag
        dt_code = {torch.float32: nat.DT_F32, torch.bfloat16: nat.DT_BF16}
        N, D, K = self.N, self.D, self.K
        any_n = N is None
        nq = self.P + 1 if self.gated else self.P
        args = [None, 0, N or 0, 0, D, None, nq, int(self.gated), self.scale, _p(T), K, _p(logit_scale), self.pool, _p(pool_w),
                None if self.identity_head else _p(W), None if self.identity_head else _p(b), self.kernel, k["qprep"], k["That"],
                k["tnorm"], k["pm"], k["pl"], k["pacc"], self.G, k["m2"], k["l"], k["out"], k["scores"], k["A"], k["ws"], k["pooled"],
                k["v"], None, k["vnorm"], None, k["incidence"], None]
        I_X, I_DT, I_N, I_LD, I_G, I_VHAT, I_LOGITS, I_STREAM = 0, 1, 2, 3, 23, 32, 34, 36
        empty, cur = torch.empty, torch.cuda.current_stream
        partials = _partials_of

        def fn(X):
            dev = X.device
            logits = empty((1, K), dtype=torch.float32, device=dev)
            vhat = empty((1, D), dtype=torch.float32, device=dev)
            a = args
            if any_n:
                n = X.shape[1]
                a[I_N], a[I_G] = n, partials(n)
            a[I_X], a[I_DT], a[I_LD] = X.data_ptr(), dt_code[X.dtype], X.stride(1)
            a[I_VHAT], a[I_LOGITS], a[I_STREAM] = vhat.data_ptr(), logits.data_ptr(), cur(dev).cuda_stream
            rc = fn_c(*a)
            if rc != 0:
                nat.check(rc, "vlsa_vlfan_forward_bag")
            return logits, vhat, That_out
        fn.keep = keep
        return fn

    def run_partial_only(self, X: torch.Tensor):
        """Just the streaming kernel (for roofline timing); queries must have been prepared by a run()."""
        dt = nat.DT_F32 if X.dtype == torch.float32 else nat.DT_BF16
        nat.check(self.lib.vlsa_vlfan_partial(_p(X), dt, self.N if self.N is not None else X.shape[0], X.stride(0), self.D, _p(self.qprep), self.P,
                                              self.kernel, _p(self.pm), _p(self.pl), _p(self.pacc),
                                              _p(self.scores), _stream()), "vlfan_partial")


# ------------------------------------------------------------------------------------------------------
# autograd: cross-attention aggregation with gradient w.r.t. the queries (X has none in the reference)
# ------------------------------------------------------------------------------------------------------
def _vlfan_dx(bags, qbuf, P: int, scale: float, dout: torch.Tensor, out: torch.Tensor, m2: torch.Tensor, l: torch.Tensor):
    """dL/dX of the cross attention for a list of fp32 [N_i, 512] bags (vlsa_vlfan_backward_dx): dout / out [B, P, 512],
    m2 / l [B, 16].  Returns a list of fp32 [N_i, 512] gradients."""
    lib = nat.load()
    dev = bags[0].device
    for x in bags:
        if x.dtype != torch.float32 or x.shape[1] != 512:
            raise VlsaNativeError("the HIP aggregation produces dX for fp32 bags with D == 512 only (the output of a trainable "
                                  "Feat_Projecter); detach the bag or use such a bag")
    dxs = [torch.empty(x.shape[0], 512, dtype=torch.float32, device=dev) for x in bags]
    B = len(bags)
    keep, p_desc, p_dx, _, p_ts, n_tiles, _ = _row_tables(bags, 64, extra=dxs)
    delta = torch.empty(B, nat.P_STRIDE, dtype=torch.float32, device=dev)
    nat.check(lib.vlsa_vlfan_backward_dx(p_desc, p_dx, B, 512, _p(qbuf), P, float(scale), p_ts, n_tiles, _p(_f32c(dout)), _p(out), _p(m2),
                                         _p(l), _p(delta), _stream()), "vlsa_vlfan_backward_dx")
    return dxs


class _VlfanAggregateFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, X, Q, gated, coattn_scale, kernel, want_attn):
        X2 = _bag2d(X)
        qp = prepare_queries(Q, gated, coattn_scale)
        pm, pl, pacc, scores = vlfan_partial(X2, qp, kernel, want_scores=want_attn)
        m2, l, out = vlfan_merge(pm, pl, pacc, normalise=True)
        A = attn_normalise(scores, m2, l) if want_attn else torch.empty(0, device=X2.device)
        ctx.save_for_backward(X2, out, m2, l, qp.buf)
        ctx.meta = (qp.nq, qp.P, qp.D, bool(gated), float(coattn_scale))
        ctx.xshape = tuple(X.shape)
        ctx.mark_non_differentiable(A)
        return out, A

    @staticmethod
    def backward(ctx, dout, _dA):
        X2, out, m2, l, qbuf = ctx.saved_tensors
        nq, P, D, gated, scale = ctx.meta
        lib = nat.load()
        N = X2.shape[0]
        dev = X2.device
        dout = _f32c(dout)
        G = num_partials(N)
        pm = torch.empty(G, nat.P_STRIDE, dtype=torch.float32, device=dev)
        pl = torch.empty(G, nat.P_STRIDE, dtype=torch.float32, device=dev)
        pacc = torch.empty(G, P, D, dtype=torch.float32, device=dev)
        prep = torch.empty(lib.vlsa_bwd_prep_bytes(D), dtype=torch.uint8, device=dev)
        dt = nat.DT_F32 if X2.dtype == torch.float32 else nat.DT_BF16
        if N == 0:
            dE = torch.zeros(P, D, dtype=torch.float32, device=dev)
        else:
            nat.check(lib.vlsa_vlfan_backward(_p(X2), dt, N, X2.stride(0), D, _p(qbuf), P, scale, _p(dout), _p(out),
                                              _p(m2), _p(l), _p(prep), _p(pm), _p(pl), _p(pacc), _stream()),
                      "vlsa_vlfan_backward")
            _, _, dE = vlfan_merge(pm, pl, pacc, normalise=False)
        # chain rule through e_p = q^_p - gated * q^_gate and q^ = q / max(|q|, eps): P x D host-side math
        qp = PreparedQueries(qbuf, nq, P, D, gated)
        qhat, qnorm = qp.qhat, qp.qnorm
        dqh = torch.cat([dE, -dE.sum(dim=0, keepdim=True)], dim=0) if gated else dE
        dQ = (dqh - qhat * (dqh * qhat).sum(dim=-1, keepdim=True)) / qnorm[:, None]
        dX = None
        if ctx.needs_input_grad[0] and N > 0:       # the bag carries a gradient (trainable Feat_Projecter in front)
            dX = _vlfan_dx([X2], qbuf, P, scale, dout[None], out[None], m2[None], l[None])[0].reshape(ctx.xshape)
        return dX, dQ, None, None, None, None


def vlfan_cross_attention(X: torch.Tensor, Q: torch.Tensor, gated: bool = False, coattn_scale: float = COATTN_SCALE,
                          kernel: int = nat.KERNEL_AUTO, want_attn: bool = False):
    """out[P, D] = softmax_N(coattn_scale * cos(Q, X)) @ X (model/deepmil.py:187-200), differentiable w.r.t. Q.
    Returns (out, A) with A[P, N] the detached attention weights (None unless want_attn)."""
    _need_gpu(X, Q)
    if torch.is_grad_enabled() and X.requires_grad and not (X.dtype == torch.float32 and X.shape[-1] == 512):
        _no_bag_grad(X)       # dX exists for fp32 D == 512 bags (vlsa_vlfan_backward_dx): anything else is refused loudly
    out, A = _VlfanAggregateFn.apply(X, Q.float(), bool(gated), float(coattn_scale), int(kernel), bool(want_attn))
    return out, (A if want_attn else None)


_GROUPS_CACHE: dict = {}


def forward_max_bags() -> int:
    """bags one forward launch of the batched path takes (256; the backward kernels and the score / pooling launches take 64)"""
    return int(nat.load().vlsa_batch_forward_max_bags())


def choose_groups(sizes, reserved_cus: int = 0) -> int:
    """Bags the persistent kernels keep in flight for these bag sizes (vlsa_batch_groups: slowest-group model)."""
    key = (tuple(sizes), int(reserved_cus))
    g = _GROUPS_CACHE.get(key)
    if g is None:
        arr = (ctypes.c_int64 * len(sizes))(*[int(n) for n in sizes])
        g = int(nat.load().vlsa_batch_groups(arr, len(sizes), int(reserved_cus)))
        if len(_GROUPS_CACHE) > 4096:
            _GROUPS_CACHE.clear()
        _GROUPS_CACHE[key] = g
    return g


class AttnBuffers:
    """One fp32 [P, ld_i] matrix per bag of a batch (ld_i = N_i rounded up to 64) in ONE allocation, plus the device-side
    ``vlsa_rows_desc`` table the batched kernels take.  The streaming kernel stores the log2-domain scores there and
    ``vlsa_attn_normalise_batch`` turns them into the attention weights in place; ``views[i]`` is bag i's A [P, N_i]."""

    def __init__(self, sizes, P: int, device):
        import numpy as np
        self.sizes, self.P = [int(n) for n in sizes], int(P)
        B = len(self.sizes)
        n = np.asarray(self.sizes, dtype=np.int64)
        lds = (n + 63) // 64 * 64
        offs = np.zeros(B + 1, dtype=np.int64)
        np.cumsum(P * lds, out=offs[1:])
        self.buf = torch.empty(max(int(offs[B]), 4), dtype=torch.float32, device=device)
        base = self.buf.data_ptr()
        # ONE upload: [B, 2] vlsa_rows_desc (pointer, pitch) followed by a [B, 3] bag table holding only the N_i (vlsa_bag_desc
        # layout: lets the normalise launch outlive a later set_bags())
        host = np.zeros(5 * B, dtype=np.int64)
        host[0:2 * B:2] = np.where(n > 0, base + 4 * offs[:B], 0)
        host[1:2 * B:2] = lds
        host[2 * B + 1::3] = n
        dev = torch.from_numpy(host).to(device)
        self.desc, self.ndesc = dev[:2 * B].view(B, 2), dev[2 * B:].view(B, 3)
        self._offs, self._lds = offs, lds
        self._views = None
        self.max_n = max(self.sizes) if self.sizes else 0

    @property
    def views(self):
        """views[i] = bag i's matrix [P, N_i] (built on first use: callers that only feed the buffers to kernels skip it)"""
        if self._views is None:
            P = self.P
            self._views = [self.buf[int(o):int(o) + P * int(ld)].view(P, int(ld))[:, :n]
                           for o, ld, n in zip(self._offs[:-1], self._lds, self.sizes)]
        return self._views


def _stage_table(host_np, device) -> torch.Tensor:
    """int64 table -> device through a ring of pinned staging buffers (async; a pageable ``.to(device)`` blocks the host for ~60 us
    per call -- four such copies were 0.25 ms of a 2.1 ms optimizer step)"""
    import threading
    ring = _TABLE_RING.setdefault(threading.get_ident(), _PinnedRing())
    return ring.stage(host_np.reshape(-1), device).view(host_np.shape)


class BagSet(list):
    """A fixed collection of bags -- a split's slides resident in HBM -- checked ONCE: a ``list`` of ``[N_i, 512]`` device tensors (so every
    API that takes a list of bags takes it) that also carries the bags' descriptor rows (pointer, N_i, row stride).  ``forward_bags`` and
    the batched autograd functions then skip the per-bag validation and table building, which for slide-sized (2k - 12k patch) bags is
    most of a call: 3.4 -> ~0.8 us per 2 798-patch bag through ``net.forward_bags``.  ``take(indices)`` / ``chunk(i, n)`` give sub-sets
    (an optimizer step's 32 bags, a launch's 64) that share the tensors.  The tensors must stay where they are (views of a
    ``DeviceBagArena``, or tensors nobody frees): the rows hold their addresses."""

    def __init__(self, bags=(), D: int = 512, _rows=None):
        import numpy as np
        if _rows is not None:                   # internal: a sub-set of a checked set
            super().__init__(bags)
            self.rows, self.D = _rows, D
        else:
            keep = []
            for i, x in enumerate(bags):
                _need_gpu(x)
                x = _bag2d(x)
                if x.shape[1] != D or x.shape[0] < 1 or (i > 0 and (x.dtype != keep[0].dtype or x.device != keep[0].device)):
                    raise VlsaNativeError("a BagSet holds non-empty bags with D == 512, one dtype (bf16 or fp32) and one device")
                if torch.is_grad_enabled() and x.requires_grad:
                    raise VlsaNativeError("a BagSet holds bags without a gradient of their own (data, not activations)")
                keep.append(x)
            super().__init__(keep)
            self.rows = np.asarray([(x.data_ptr(), x.shape[0], x.stride(0)) for x in keep], dtype=np.int64).reshape(len(keep), 3)
            self.D = D
        self.sizes = tuple(int(n) for n in self.rows[:, 1])
        self.dt = (nat.DT_F32 if self[0].dtype == torch.float32 else nat.DT_BF16) if len(self) else nat.DT_BF16
        self._chunks, self._groups, self._desc = {}, {}, None

    def take(self, indices) -> "BagSet":
        idx = [int(i) for i in indices]
        return BagSet([self[i] for i in idx], self.D, _rows=self.rows[idx])

    def chunk(self, start: int, n: int) -> "BagSet":
        """bags [start, start + n) as a BagSet of their own (kept: an evaluation loop asks for the same chunks every epoch)"""
        if start == 0 and n >= len(self):
            return self
        c = self._chunks.get((start, n))
        if c is None:
            c = self._chunks[(start, n)] = BagSet(list.__getitem__(self, slice(start, start + n)), self.D, _rows=self.rows[start:start + n])
        return c

    def groups(self, reserved_cus: int = 0) -> int:
        g = self._groups.get(reserved_cus)
        if g is None:
            g = self._groups[reserved_cus] = choose_groups(self.sizes, reserved_cus)
        return g

    def desc(self) -> torch.Tensor:
        """the [B, 3] int64 descriptor table on the bags' device (uploaded once per set)"""
        if self._desc is None:
            self._desc = _stage_table(self.rows, self[0].device)
        return self._desc


class _BagTable:
    """Device-side descriptor table (pointer, N, row stride) of up to 64 bags for the batched kernels."""

    def __init__(self, bags, D=512):
        lib = nat.load()
        B = len(bags)
        if not (1 <= B <= lib.vlsa_batch_max_bags()):
            raise ValueError(f"batch size {B} outside [1, {lib.vlsa_batch_max_bags()}]")
        import numpy as np
        if isinstance(bags, BagSet) and bags.D == D:      # checked once, rows kept: no per-bag work
            self.bags, self.B, self.D, self.dt, self.desc = bags, B, D, bags.dt, bags.desc()
            self.sizes = bags.sizes
            return
        keep, rows = [], []
        for i, x in enumerate(bags):
            _need_gpu(x)
            x = _bag2d(x)
            if x.shape[1] != D or (i > 0 and x.dtype != keep[0].dtype):
                raise VlsaNativeError("the batched path takes bags with D == 512 and one dtype (bf16 or fp32) per batch")
            keep.append(x)
            n = x.shape[0]
            rows.append((x.data_ptr(), n, x.stride(0) if n > 0 else D))
        self.bags, self.B, self.D = keep, B, D
        self.dt = nat.DT_F32 if keep[0].dtype == torch.float32 else nat.DT_BF16
        self.desc = _stage_table(np.asarray(rows, dtype=np.int64).reshape(B, 3), keep[0].device)
        self.sizes = tuple(r[1] for r in rows)


class _VlfanBatchAggregateFn(torch.autograd.Function):
    """out[B, P, D] for B bags sharing the queries Q; differentiable w.r.t. Q (the bags carry no gradient)."""

    @staticmethod
    def forward(ctx, Q, gated, coattn_scale, table, attn, *bag_tensors):
        """bag_tensors: the bags of `table` once more, as autograd inputs (a bag that requires grad -- the output of a trainable
        Feat_Projecter -- receives dX in backward; fp32 only)"""
        lib, s = nat.load(), _stream()
        B, D = table.B, table.D
        dev = table.desc.device
        qp = prepare_queries(Q, gated, coattn_scale)
        P = qp.P
        ws = torch.empty(lib.vlsa_batch_workspace_bytes(B, P, D), dtype=torch.uint8, device=dev)
        groups = table.bags.groups(0) if isinstance(table.bags, BagSet) else choose_groups(table.sizes, 0)
        nat.check(lib.vlsa_vlfan_partial_batch_scores(_p(table.desc), B, table.dt, D, _p(qp.buf), P, _p(ws), 0, groups,
                                                      None if attn is None else _p(attn.desc), s),
                  "vlsa_vlfan_partial_batch")
        G = int(lib.vlsa_batch_partials_per_bag_ex(B, 0, groups))
        wf = ws.view(torch.float32)
        pm, pl, pacc = wf, wf[B * G * nat.P_STRIDE:], wf[2 * B * G * nat.P_STRIDE:]
        m2 = torch.empty(B, nat.P_STRIDE, dtype=torch.float32, device=dev)
        l = torch.empty(B, nat.P_STRIDE, dtype=torch.float32, device=dev)
        out = torch.empty(B, P, D, dtype=torch.float32, device=dev)
        st = (ctypes.c_int64 * 9)(nat.P_STRIDE, nat.P_STRIDE, P * D, G * nat.P_STRIDE, G * nat.P_STRIDE, G * P * D,
                                  nat.P_STRIDE, nat.P_STRIDE, P * D)
        nat.check(lib.vlsa_vlfan_merge_batch_strided(_p(pm), _p(pl), _p(pacc), B, G, P, D, 1, st, _p(m2), _p(l), _p(out), s),
                  "vlsa_vlfan_merge_batch_strided")
        if attn is not None:   # scores -> attention weights in place, with the bag-global (m2, l)
            nat.check(lib.vlsa_attn_normalise_batch(_p(table.desc), B, P, attn.max_n, _p(attn.desc), _p(m2), _p(l), _p(attn.desc), s),
                      "vlsa_attn_normalise_batch")
        ctx.save_for_backward(out, m2, l, qp.buf)
        ctx.table = table
        ctx.xshapes = [tuple(x.shape) for x in bag_tensors]
        ctx.meta = (qp.nq, P, D, bool(gated), float(coattn_scale))
        return out

    @staticmethod
    def backward(ctx, dout):
        out, m2, l, qbuf = ctx.saved_tensors
        table = ctx.table
        nq, P, D, gated, scale = ctx.meta
        lib, s = nat.load(), _stream()
        B, dev = table.B, out.device
        dout = _f32c(dout)
        if (table.dt == nat.DT_BF16 and P <= 12) or table.dt == nat.DT_F32:   # the persistent batch kernels (bf16: P <= 12; fp32: any P)
            G = lib.vlsa_bwd_batch_partials()
            pm = torch.empty(G, nat.P_STRIDE, dtype=torch.float32, device=dev)
            pl = torch.empty(G, nat.P_STRIDE, dtype=torch.float32, device=dev)
            pacc = torch.empty(G, P, D, dtype=torch.float32, device=dev)
            prep = torch.empty(lib.vlsa_bwd_batch_prep_bytes(B, D), dtype=torch.uint8, device=dev)
            nat.check(lib.vlsa_vlfan_backward_batch(_p(table.desc), B, table.dt, D, _p(qbuf), P, scale, _p(dout), _p(out),
                                                    _p(m2), _p(l), _p(prep), _p(pm), _p(pl), _p(pacc),
                                                    choose_groups([x.shape[0] for x in table.bags], 0), s),
                      "vlsa_vlfan_backward_batch")
            _, _, dE = vlfan_merge(pm, pl, pacc, normalise=False)
        else:  # fp32 bags or P > 12: the per-bag kernel over the bag table in ONE launch, partial sums of all bags reduced together
            G = max(num_partials(x.shape[0]) for x in table.bags)
            pm = torch.empty(B * G, nat.P_STRIDE, dtype=torch.float32, device=dev)
            pl = torch.empty(B * G, nat.P_STRIDE, dtype=torch.float32, device=dev)
            pacc = torch.empty(B * G, P, D, dtype=torch.float32, device=dev)
            prep = torch.empty(lib.vlsa_bwd_batch_prep_bytes(B, D), dtype=torch.uint8, device=dev)
            nat.check(lib.vlsa_vlfan_backward_bags(_p(table.desc), B, table.dt, D, _p(qbuf), P, scale, _p(dout), _p(out), _p(m2),
                                                   _p(l), _p(prep), _p(pm), _p(pl), _p(pacc), G, s), "vlsa_vlfan_backward_bags")
            _, _, dE = vlfan_merge(pm, pl, pacc, normalise=False)
        # chain rule to the raw queries (normalisation + gate row) in one launch (round 6; was five torch kernels on [P, 512])
        dQ = torch.empty(nq, D, dtype=torch.float32, device=dev)
        nat.check(lib.vlsa_query_chain(_p(dE.contiguous()), _p(qbuf), nq, int(gated), D, _p(dQ), s), "vlsa_query_chain")
        dxs = [None] * len(ctx.xshapes)         # (a BagSet's bags are not autograd inputs: none)
        if any(ctx.needs_input_grad[5:]):           # bags that carry a gradient (projected by a trainable Feat_Projecter)
            got = _vlfan_dx(table.bags, qbuf, P, scale, dout, out, m2, l)
            dxs = [g.reshape(shape) if need else None for g, shape, need in zip(got, ctx.xshapes, ctx.needs_input_grad[5:])]
        return (dQ, None, None, None, None, *dxs)


def vlfan_cross_attention_bags(bags, Q: torch.Tensor, gated: bool = False, coattn_scale: float = COATTN_SCALE,
                               want_attn: bool = False):
    """out[B, P, D]: ``vlfan_cross_attention`` for a list of up to 64 bags that share the queries, through the
    persistent multi-bag kernels (forward and backward); what one optimizer step of the reference does bag by bag
    (runner/vlsa_handler.py:260-289).  Bags: [N_i, 512] device tensors, N_i >= 1, one dtype per batch.
    want_attn: also return the detached attention weights, a list of [P, N_i] (model/deepmil.py:198,206-215)."""
    if isinstance(bags, BagSet):              # checked once: non-empty, no gradient of their own -> not autograd inputs either
        table = _BagTable(bags)
        if not want_attn:
            return _VlfanBatchAggregateFn.apply(Q.float(), bool(gated), float(coattn_scale), table, None)
        attn = AttnBuffers(bags.sizes, Q.shape[0] - (1 if gated else 0), table.desc.device)
        return _VlfanBatchAggregateFn.apply(Q.float(), bool(gated), float(coattn_scale), table, attn), attn.views
    if torch.is_grad_enabled():
        _no_bag_grad(*[x for x in bags if not (x.dtype == torch.float32 and x.shape[-1] == 512)])   # dX: fp32 D == 512 bags only
    table = _BagTable(bags)
    if any(x.shape[0] == 0 for x in table.bags):
        raise VlsaNativeError("empty bag in a batch")
    if not want_attn:
        return _VlfanBatchAggregateFn.apply(Q.float(), bool(gated), float(coattn_scale), table, None, *bags)
    P = Q.shape[0] - (1 if gated else 0)
    attn = AttnBuffers([x.shape[0] for x in table.bags], P, table.desc.device)
    out = _VlfanBatchAggregateFn.apply(Q.float(), bool(gated), float(coattn_scale), table, attn, *bags)
    return out, attn.views


# ------------------------------------------------------------------------------------------------------
# FeatMIL / DeepMIL / zero-shot pieces (model/deepmil.py:16-67,222-292; model/layers.py:85-153)
# ------------------------------------------------------------------------------------------------------
def _dt(X):
    return nat.DT_F32 if X.dtype == torch.float32 else nat.DT_BF16


def _scored_pool_raw(X2: torch.Tensor, scores: Optional[torch.Tensor]):
    lib = nat.load()
    N, D = X2.shape
    G = int(lib.vlsa_pool_num_partials(N))
    dev = X2.device
    pm = torch.empty(G, nat.P_STRIDE, dtype=torch.float32, device=dev)
    pl = torch.empty(G, nat.P_STRIDE, dtype=torch.float32, device=dev)
    pacc = torch.empty(G, 1, D, dtype=torch.float32, device=dev)
    nat.check(lib.vlsa_scored_pool_partial(_p(X2), _dt(X2), N, X2.stride(0), D, _p(scores), _p(pm), _p(pl), _p(pacc),
                                           _stream()), "vlsa_scored_pool_partial")
    return vlfan_merge(pm, pl, pacc, normalise=True)  # m2[16], l[16], out[1, D]


def rowdot(X: torch.Tensor, v: torch.Tensor) -> torch.Tensor:
    _need_gpu(X, v)
    lib = nat.load()
    X2 = _bag2d(X)
    N, D = X2.shape
    out = torch.empty(N, dtype=torch.float32, device=X2.device)
    nat.check(lib.vlsa_rowdot(_p(X2), _dt(X2), N, X2.stride(0), D, _p(_f32c(v)), _p(out), _stream()), "vlsa_rowdot")
    return out


class _ScoredPoolFn(torch.autograd.Function):
    """pooled[D] = softmax_N(a) @ X, differentiable w.r.t. the raw scores a[N] (X has no gradient)."""

    @staticmethod
    def forward(ctx, X2, a):
        a = _f32c(a).reshape(-1)
        m2, l, out = _scored_pool_raw(X2, a)
        ctx.save_for_backward(X2, a, m2, l, out)
        return out[0]

    @staticmethod
    def backward(ctx, dpooled):
        X2, a, m2, l, out = ctx.saved_tensors
        dp = _f32c(dpooled).reshape(-1)
        N, D = X2.shape
        da = torch.empty(N, dtype=torch.float32, device=X2.device)
        if D % (8 if X2.dtype == torch.bfloat16 else 4) == 0 and D <= nat.MAX_D:
            # da_n = A_n (x_n . dp - pooled . dp) in ONE pass over X (was: row dots + four [N]-sized torch kernels)
            nat.check(nat.load().vlsa_scored_pool_backward(_p(X2), _dt(X2), N, X2.stride(0), D, _p(a), _p(m2), _p(l), _p(out), _p(dp),
                                                           _p(da), _stream()), "vlsa_scored_pool_backward")
            return None, da
        xd = rowdot(X2, dp)                                    # odd feature widths: row dots in HIP, [N] vector math in torch
        A = torch.exp2(a * 1.4426950408889634 - m2[0]) / l[0]
        return None, A * (xd - (out[0] * dp).sum())


def scored_pool(X: torch.Tensor, scores: Optional[torch.Tensor]) -> torch.Tensor:
    """softmax_N(scores) @ X -> [D]; scores None => mean over the N rows."""
    _need_gpu(X, scores)
    _no_bag_grad(X)
    X2 = _bag2d(X)
    if X2.shape[0] == 0:
        raise ValueError("empty bag")
    if scores is None:
        return _scored_pool_raw(X2, None)[2][0]
    return _ScoredPoolFn.apply(X2, scores)


def colmax(X: torch.Tensor) -> torch.Tensor:
    _need_gpu(X)
    lib = nat.load()
    X2 = _bag2d(X)
    N, D = X2.shape
    G = int(lib.vlsa_pool_num_partials(N))
    part = torch.empty(G, D, dtype=torch.float32, device=X2.device)
    out = torch.empty(D, dtype=torch.float32, device=X2.device)
    nat.check(lib.vlsa_colmax(_p(X2), _dt(X2), N, X2.stride(0), D, _p(part), _p(out), _stream()), "vlsa_colmax")
    return out


def adapter_head(f: torch.Tensor, W1: torch.Tensor, W2: torch.Tensor, keep_ratio: float) -> torch.Tensor:
    """keep * f + (1 - keep) * relu(W2 relu(W1 f)) for one pooled bag vector f [D] (DeepMIL's Adapter head, inference)."""
    _need_gpu(f, W1, W2)
    lib = nat.load()
    f = _f32c(f).reshape(-1)
    R, D = W1.shape
    hid = torch.empty(R, dtype=torch.float32, device=f.device)
    out = torch.empty(D, dtype=torch.float32, device=f.device)
    nat.check(lib.vlsa_adapter_head(_p(f), D, _p(_f32c(W1)), R, _p(_f32c(W2)), float(keep_ratio), _p(hid), _p(out), _stream()),
              "vlsa_adapter_head")
    return out


def attn_scores(H: torch.Tensor, Hg: Optional[torch.Tensor], b1, bg, w2, b2) -> torch.Tensor:
    """a[n] = w2 . (tanh(H[n] + b1) [* sigmoid(Hg[n] + bg)]) + b2 (inference path; H = X W1^T from rocBLAS)."""
    _need_gpu(H)
    lib = nat.load()
    H = _f32c(H)
    N, hid = H.shape
    a = torch.empty(N, dtype=torch.float32, device=H.device)
    Hg_ = _f32c(Hg) if Hg is not None else None
    nat.check(lib.vlsa_attn_scores(_p(H), _p(Hg_), N, hid, _p(_f32c(b1)), _p(_f32c(bg)) if bg is not None else None,
                                   _p(_f32c(w2).reshape(-1)), _p(_f32c(b2).reshape(-1)), _p(a), _stream()),
              "vlsa_attn_scores")
    return a


_SCORE_TILING = {}


def _score_tiling(f32: bool, gated: bool):
    """(rows of the score kernel's largest tile, row tiles per round of the CUs) -- vlsa_gated_scores_tiling, asked once."""
    key = (f32, gated)
    t = _SCORE_TILING.get(key)
    if t is None:
        mr, rt = ctypes.c_int(0), ctypes.c_int(0)
        nat.check(nat.load().vlsa_gated_scores_tiling(nat.DT_F32 if f32 else nat.DT_BF16, int(gated), ctypes.addressof(mr),
                                                     ctypes.addressof(rt)), "vlsa_gated_scores_tiling")
        t = _SCORE_TILING[key] = (mr.value, rt.value)
    return t


_SCORE_BIG_TILE = {}
_NO_FUSED_POOL = os.environ.get("VLSA_GS_NO_FUSED_POOL", "") == "1"      # (A/B hook: scores and pooling as two launches)


def _score_big_tile(f32: bool, gated: bool):
    """(tile height of the persistent LDS-DMA score kernel, rows in a batch from which to use it) -- vlsa_gated_scores_big_tile;
    (0, 0): not for this dtype / module."""
    key = (f32, gated)
    t = _SCORE_BIG_TILE.get(key)
    if t is None:
        rows, mn = ctypes.c_int(0), ctypes.c_int64(0)
        nat.check(nat.load().vlsa_gated_scores_big_tile(nat.DT_F32 if f32 else nat.DT_BF16, int(gated), ctypes.addressof(rows),
                                                       ctypes.addressof(mn)), "vlsa_gated_scores_big_tile")
        t = _SCORE_BIG_TILE[key] = (rows.value, mn.value)
    return t


_POOL_WS: dict = {}        # N -> vlsa_gated_scores_pool_ws_floats(N)


class FusedAttnScores:
    """Raw attention scores a[N] of (Gated_)Attention_Pooling over all patches of a bf16 or fp32 bag in ONE MFMA kernel
    (vlsa_gated_scores; model/layers.py:85-153): the [N, 256] hidden activations never reach memory.  Holds the weights
    packed in MFMA-fragment order (bf16 hi + lo split) and re-packs them when a parameter changes."""

    def __init__(self):
        self._key, self._prep = None, None
        self._pool_ws = {}        # (device, stream) -> tile workspace of scores_and_pool (scratch, never returned)

    @staticmethod
    def supported(X2: torch.Tensor, dim_in: int, dim_hid: int) -> bool:
        return (X2.is_cuda and X2.dtype in (torch.bfloat16, torch.float32) and dim_in == 512 and dim_hid == 256
                and X2.shape[0] > 0)

    def _packed(self, device, Wa, ba, Wg, bg, w2, c) -> torch.Tensor:
        lib = nat.load()
        gated = Wg is not None
        params = [t for t in (Wa, ba, Wg, bg, w2, c) if t is not None]
        key = tuple((t.data_ptr(), t._version) for t in params)
        if key != self._key:
            prep = torch.empty(lib.vlsa_gated_prep_bytes(int(gated)), dtype=torch.uint8, device=device)
            f = lambda t: None if t is None else _f32c(t).reshape(-1)  # noqa: E731
            keep = [f(t) for t in (Wa, ba, Wg, bg, w2, c)]
            nat.check(lib.vlsa_prepare_gated_weights(*[_p(t) for t in keep], Wa.shape[1], Wa.shape[0], int(gated), _p(prep),
                                                     _stream()), "vlsa_prepare_gated_weights")
            self._key, self._prep = key, prep
        return self._prep

    def packed_t(self, device, Wa, Wg) -> torch.Tensor:
        """the un-scaled weights packed [hidden][column] for the dX contraction (vlsa_prepare_attn_dx_weights), per parameter version"""
        lib = nat.load()
        params = [t for t in (Wa, Wg) if t is not None]
        key = tuple((id(t), t._version) for t in params) + (device,)
        if key != getattr(self, "_key_t", None):
            gated = Wg is not None
            prep = torch.empty(lib.vlsa_attn_dx_prep_bytes(int(gated)), dtype=torch.uint8, device=device)
            keep = [_f32c(t) for t in params]
            nat.check(lib.vlsa_prepare_attn_dx_weights(_p(keep[0]), _p(keep[1]) if gated else None, int(gated), _p(prep), _stream()),
                      "vlsa_prepare_attn_dx_weights")
            self._key_t, self._prep_t, self._params_t = key, prep, params
        return self._prep_t

    def pool_bags(self, bags, Wa, ba, Wg, bg, w2, c):
        """The N-sized part of DeepMIL for up to 64 bags ([N_i, 512], one dtype, validated by the caller) in three launches:
        raw scores of all bags (vlsa_gated_scores_batch), softmax-weighted row sums (vlsa_scored_pool_partial_batch), fold.
        Returns (pooled [B, 512] fp32, scores [sum N_i] fp32, offsets [B + 1]) -- model/layers.py:103-122,137-153 per bag."""
        lib, s, dev = nat.load(), _stream(), bags[0].device
        gated, B = Wg is not None, len(bags)
        if not (1 <= B <= 64):
            raise ValueError("1..64 bags per call")
        prep = self._packed(dev, Wa, ba, Wg, bg, w2, c)
        rows = [x.shape[0] for x in bags]
        f32 = bags[0].dtype == torch.float32
        max_rows, round_tiles = _score_tiling(f32, bool(gated))
        rpt = max_rows
        big_rows, big_min = _score_big_tile(f32, bool(gated))
        fused = bool(big_rows) and sum(rows) >= big_min and not _NO_FUSED_POOL
        if big_rows and not fused and not gated:
            big_rows = 0                          # (the ungated module's plain score launches keep the fragment-order kernel)
        if fused:            # a large batch: the persistent LDS-DMA kernel, scores and pooling in ONE launch (a workgroup per row tile)
            rpt = big_rows
            if sum((n + rpt - 1) // rpt for n in rows) < 256:                    # less than one round: the lowest tiles that still fit it
                rpt = max_rows + 32
                while rpt < big_rows and sum((n + rpt - 1) // rpt for n in rows) > 256:
                    rpt += 32
        elif sum((n + max_rows - 1) // max_rows for n in rows) < round_tiles:    # less than one round of the 256 CUs: smaller tiles
            rpt = 16
            while rpt < max_rows and sum((n + rpt - 1) // rpt for n in rows) > round_tiles:
                rpt += 16
        import numpy as np
        stage = getattr(self, "_stage", None)
        if stage is None or stage[0] != B:
            meta = torch.empty(4 * B + (B + 2) // 2, dtype=torch.int64).pin_memory()
            stage = self._stage = [B, meta, None]
        elif stage[2] is not None:
            stage[2].synchronize()                 # the previous call's async table upload has read the staging buffer
        meta = stage[1]
        m = meta.numpy()
        m[:3 * B] = np.asarray([(x.data_ptr(), n, x.stride(0)) for x, n in zip(bags, rows)], dtype=np.int64).reshape(-1)
        offs = np.zeros(B + 1, dtype=np.int64)
        np.cumsum(rows, out=offs[1:])
        m[3 * B:4 * B] = offs[:B]
        ts = m[4 * B:].view(np.int32)
        ts[0] = 0
        np.cumsum([(n + rpt - 1) // rpt for n in rows], out=ts[1:B + 1])
        n_tiles, total = int(ts[B]), int(offs[B])
        meta_d = meta.to(dev, non_blocking=True)
        stage[2] = torch.cuda.Event()
        stage[2].record()
        base = meta_d.data_ptr()
        a = torch.empty(total, dtype=torch.float32, device=dev)
        dt = nat.DT_F32 if f32 else nat.DT_BF16
        if fused:
            ws = torch.empty(n_tiles * 514, dtype=torch.float32, device=dev)
            pooled = torch.empty(B, 512, dtype=torch.float32, device=dev)
            nat.check(lib.vlsa_gated_scores_pool_batch(base, B, dt, 512, _p(prep), int(gated), base + 32 * B, n_tiles, rpt, _p(a),
                                                       base + 24 * B, _p(ws), _p(pooled), s), "vlsa_gated_scores_pool_batch")
            self._keep = (meta_d, bags, ws)            # the kernels read these
            return pooled, a, offs
        G = max(1, min(64, 512 // B))
        pm = torch.empty(B * G, nat.P_STRIDE, dtype=torch.float32, device=dev)
        pl = torch.empty(B * G, nat.P_STRIDE, dtype=torch.float32, device=dev)
        pacc = torch.empty(B * G, 512, dtype=torch.float32, device=dev)
        m2 = torch.empty(B, nat.P_STRIDE, dtype=torch.float32, device=dev)
        l = torch.empty(B, nat.P_STRIDE, dtype=torch.float32, device=dev)
        pooled = torch.empty(B, 512, dtype=torch.float32, device=dev)
        nat.check(lib.vlsa_gated_scores_batch(base, B, dt, 512, _p(prep), int(gated), base + 32 * B, n_tiles, rpt, _p(a),
                                              base + 24 * B, total, s), "vlsa_gated_scores_batch")
        nat.check(lib.vlsa_scored_pool_partial_batch(base, B, dt, 512, _p(a), base + 24 * B, G, _p(pm), _p(pl), _p(pacc), s),
                  "vlsa_scored_pool_partial_batch")
        st = (ctypes.c_int64 * 9)(nat.P_STRIDE, nat.P_STRIDE, 512, G * nat.P_STRIDE, G * nat.P_STRIDE, G * 512,
                                  nat.P_STRIDE, nat.P_STRIDE, 512)
        nat.check(lib.vlsa_vlfan_merge_batch_strided(_p(pm), _p(pl), _p(pacc), B, G, 1, 512, 1, st, _p(m2), _p(l), _p(pooled), s),
                  "vlsa_vlfan_merge_batch_strided")
        self._keep = (meta_d, bags)                # the kernels read these
        return pooled, a, offs

    def scores_and_pool(self, X2, Wa, ba, Wg, bg, w2, c, adapter=None):
        """(pooled [1, 512] fp32, raw scores [N]) of ONE bag from ONE host call (vlsa_gated_scores_pool): bf16 bags -- scores and pooling
        in one launch of the persistent LDS-DMA kernel; fp32 bags -- score kernel, pooling partials and merge chained inside the
        library.  None where the bag's layout rules it out (or VLSA_GS_NO_FUSED_POOL=1): the caller then takes the score kernel and
        the pooling kernel one after the other.  adapter = (W1 [R, 512], W2 [512, R], keep_ratio): DeepMIL's Adapter head
        (model/deepmil.py:283-286) behind it in the same call -> (pooled, scores, logit [1, 512])."""
        esz = X2.element_size()
        if (_NO_FUSED_POOL or X2.stride(0) * 256 * esz >= (1 << 31) or X2.stride(1) != 1 or (X2.data_ptr() & 15)
                or (X2.stride(0) * esz) % 16):
            return None
        lib = nat.load()
        prep = self._packed(X2.device, Wa, ba, Wg, bg, w2, c)
        N = X2.shape[0]
        nws = _POOL_WS.get(N)
        if nws is None:
            if len(_POOL_WS) > 4096:
                _POOL_WS.clear()
            nws = _POOL_WS[N] = lib.vlsa_gated_scores_pool_ws_floats(N)
        # What the caller may keep alive (pooled / logit views end up as DeepMIL's outputs, the scores as its attention) must not pin the
        # tile workspace (~0.35 MB per bag: an evaluation loop that keeps features on the GPU grew by 0.5 MB per slide):
        # scores and the 512-float outputs are their own small tensors, the workspace is scratch of this object, one per (device, stream)
        # -- launches of one stream are ordered, so the next call may overwrite it.
        dev = X2.device
        key = (dev.index, torch.cuda.current_stream(dev).cuda_stream)
        R = 0 if adapter is None else adapter[0].shape[0]
        ws = self._pool_ws.get(key)
        if ws is None or ws.numel() < nws + R:
            if len(self._pool_ws) > 16:
                self._pool_ws.clear()
            ws = self._pool_ws[key] = torch.empty(max(nws + R, 1 << 16), dtype=torch.float32, device=dev)
        a = torch.empty(N, dtype=torch.float32, device=dev)
        if adapter is None:
            pooled = torch.empty(1, 512, dtype=torch.float32, device=dev)
            nat.check(lib.vlsa_gated_scores_pool(_p(X2), _dt(X2), N, X2.stride(0), X2.shape[1], _p(prep), int(Wg is not None), _p(a), _p(ws),
                                                 _p(pooled), _stream()), "vlsa_gated_scores_pool")
            return pooled, a
        W1, W2, keep = adapter                           # DeepMIL's Adapter head behind the pooling, same host call
        out = torch.empty(2, 1, 512, dtype=torch.float32, device=dev)                    # pooled | logit
        pooled, logit = out[0], out[1]
        nat.check(lib.vlsa_gated_scores_pool_adapter(_p(X2), _dt(X2), N, X2.stride(0), X2.shape[1], _p(prep), int(Wg is not None), _p(a),
                                                     _p(ws), _p(pooled), _p(_f32c(W1)), R, _p(_f32c(W2)), float(keep), _p(logit), _stream()),
                  "vlsa_gated_scores_pool_adapter")
        return pooled, a, logit

    def __call__(self, X2, Wa, ba, Wg, bg, w2, c, drop_p: float = 0.0, seed: int = 0) -> torch.Tensor:
        lib = nat.load()
        gated = Wg is not None
        self._packed(X2.device, Wa, ba, Wg, bg, w2, c)
        N = X2.shape[0]
        a = torch.empty(N, dtype=torch.float32, device=X2.device)
        if drop_p and gated:      # training-mode dropout of the gated module, counter-based masks (vlsa_gated_scores_train)
            nat.check(lib.vlsa_gated_scores_train(_p(X2), _dt(X2), N, X2.stride(0), X2.shape[1], _p(self._prep), 1, _p(a), float(drop_p),
                                                  int(seed) & 0xFFFFFFFF, _stream()), "vlsa_gated_scores_train")
            return a
        nat.check(lib.vlsa_gated_scores(_p(X2), _dt(X2), N, X2.stride(0), X2.shape[1], _p(self._prep), int(gated), _p(a),
                                        _stream()), "vlsa_gated_scores")
        return a


class _PinnedRing:
    """Small ring of pinned int64 staging buffers for the descriptor tables of the multi-bag kernels: the upload is an ASYNC copy
    on the current stream (a pageable `.to(device)` blocks the host for tens of microseconds per call); a slot is reused only after
    the copy that read it has completed (event)."""

    def __init__(self, slots: int = 8, words: int = 1024):
        self.slots, self.words, self.bufs, self.events, self.i = slots, words, None, None, 0

    def stage(self, host_np, device):
        import numpy as np
        n = int(host_np.shape[0])
        if n > self.words or not torch.cuda.is_available():
            return torch.from_numpy(np.ascontiguousarray(host_np)).to(device)
        if self.bufs is None:
            self.bufs = [torch.empty(self.words, dtype=torch.int64).pin_memory() for _ in range(self.slots)]
            self.events = [None] * self.slots
        k = self.i % self.slots
        self.i += 1
        if self.events[k] is not None:
            self.events[k].synchronize()
        self.bufs[k].numpy()[:n] = host_np
        dev = self.bufs[k][:n].to(device, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(device))
        self.events[k] = ev
        return dev


_TABLE_RING = {}     # per host thread (autograd runs backward functions on its own thread)


def _row_tables(bags, tile_rows: int, extra=None):
    """ONE upload for the tables the multi-bag backward kernels take: vlsa_bag_desc [B, 3] (pointer, N, row stride), optionally a
    second table (``extra``: e.g. the gradient rows), row offsets [B] (int64) and tile_start [B + 1] (int32, tiles of
    ``tile_rows`` rows).  Returns (device buffer, ptr of desc, ptr of extra desc or None, ptr of row_off, ptr of tile_start,
    n_tiles, row offsets as a list)."""
    import threading
    B = len(bags)
    if B == 1 and bags[0].shape[0] > 0:
        # one bag (the bag-by-bag training loop): the tables are written on the device from by-value arguments -- one launch
        # instead of numpy bookkeeping + a pinned staging copy + an event per call
        x, lib = bags[0], nat.load()
        s = _stream()
        buf = torch.empty(8, dtype=torch.int64, device=x.device)      # 64 bytes per call: no state shared between calls
        ex = extra[0] if extra is not None else None
        n_tiles = lib.vlsa_fill_one_bag_tables(_p(buf), _p(x), x.shape[0], x.stride(0), _p(ex), 0 if ex is None else ex.stride(0), int(tile_rows), s)
        if n_tiles < 0:
            nat.check(n_tiles, "vlsa_fill_one_bag_tables")
        base = buf.data_ptr()
        o = 6 if ex is not None else 3
        return buf, base, (base + 24 if ex is not None else None), base + 8 * o, base + 8 * (o + 1), int(n_tiles), [0, int(x.shape[0])]
    import numpy as np
    rows = [int(x.shape[0]) for x in bags]
    n64 = 3 * B + (3 * B if extra is not None else 0) + B + (B + 2) // 2
    host = np.zeros(n64, dtype=np.int64)
    host[:3 * B] = np.asarray([(x.data_ptr(), n, x.stride(0)) for x, n in zip(bags, rows)], dtype=np.int64).reshape(-1)
    o = 3 * B
    if extra is not None:
        host[o:o + 3 * B] = np.asarray([(x.data_ptr(), n, x.stride(0)) for x, n in zip(extra, rows)], dtype=np.int64).reshape(-1)
        o += 3 * B
    offs = np.zeros(B + 1, dtype=np.int64)
    np.cumsum(rows, out=offs[1:])
    host[o:o + B] = offs[:B]
    ts = host[o + B:].view(np.int32)
    ts[0] = 0
    np.cumsum([(n + tile_rows - 1) // tile_rows for n in rows], out=ts[1:B + 1])
    n_tiles = int(ts[B])
    ring = _TABLE_RING.setdefault(threading.get_ident(), _PinnedRing())
    dev = ring.stage(host, bags[0].device)
    base = dev.data_ptr()
    p_extra = base + 24 * B if extra is not None else None
    p_off = base + 8 * o
    return dev, base, p_extra, p_off, p_off + 8 * B, n_tiles, offs


class _AttnScoresFn(torch.autograd.Function):
    """Raw (gated) attention scores a [N] of one bag with a gradient for the pooling module's parameters: forward = the fused MFMA
    kernel (vlsa_gated_scores), backward = vlsa_attn_scores_backward (hidden activations recomputed tile by tile, dW = dH^T X in
    the same kernel) -- model/layers.py:103-122,137-153 under autograd without the [N, 256] activations in memory."""

    @staticmethod
    def forward(ctx, X2, fused, Wa, ba, Wg, bg, w2, c, drop_p, seed):
        a = fused(X2, Wa, ba, Wg, bg, w2, c, drop_p=drop_p, seed=seed)
        ctx.save_for_backward(X2, fused._prep)       # the packed weights of THIS parameter version
        ctx.gated = Wg is not None
        ctx.shapes = (w2.shape, c.shape)
        ctx.drop = (float(drop_p), int(seed))
        return a

    @staticmethod
    def backward(ctx, da):
        lib = nat.load()
        X2, prep = ctx.saved_tensors
        gated = ctx.gated
        dev = X2.device
        da = _f32c(da).reshape(-1)
        tile_rows = int(lib.vlsa_mlp_bwd_tile_rows(_dt(X2)))
        keep, p_desc, _, p_off, p_ts, n_tiles, _ = _row_tables([X2], tile_rows)
        ws = torch.empty(lib.vlsa_mlp_bwd_workspace_bytes(1 if gated else 0, n_tiles), dtype=torch.uint8, device=dev)
        dW = torch.empty(2 if gated else 1, 256, 512, dtype=torch.float32, device=dev)
        dvec = torch.empty(3, 512, dtype=torch.float32, device=dev)
        nat.check(lib.vlsa_attn_scores_backward(p_desc, 1, _dt(X2), 512, _p(prep), int(gated), p_ts, n_tiles, _p(da), p_off, _p(ws),
                                                _p(dW), _p(dvec), ctx.drop[0], ctx.drop[1], _stream()), "vlsa_attn_scores_backward")
        w2_shape, c_shape = ctx.shapes
        return (None, None, dW[0], dvec[0, :256], dW[1] if gated else None, dvec[0, 256:] if gated else None,
                dvec[1, :256].reshape(w2_shape), dvec[2, :1].reshape(c_shape), None, None)


class _AttnPoolFn(torch.autograd.Function):
    """pooled [512] = softmax_N(a) @ X with a = the (gated) attention scores, as ONE autograd node (model/deepmil.py:267-283): forward
    = score kernel + pooling kernels, backward = da in one pass (vlsa_scored_pool_backward), the parameter gradients
    (vlsa_attn_scores_backward) and -- only for a bag that ITSELF carries a gradient (a trainable Feat_Projecter in front) --
    dX = dHa Wa + dHg Wg + A dpooled (vlsa_attn_scores_backward_dx).  Returns (pooled, a); a is not differentiable here."""

    @staticmethod
    def forward(ctx, X2, fused, Wa, ba, Wg, bg, w2, c, drop_p, seed):
        a = fused(X2, Wa, ba, Wg, bg, w2, c, drop_p=drop_p, seed=seed)
        m2, l, out = _scored_pool_raw(X2, a)
        gated = Wg is not None
        ctx.save_for_backward(X2, a, m2, l, out, fused._prep, fused.packed_t(X2.device, Wa, Wg) if ctx.needs_input_grad[0] else None)
        ctx.gated, ctx.shapes, ctx.drop = gated, (w2.shape, c.shape), (float(drop_p), int(seed))
        ctx.mark_non_differentiable(a)
        return out[0], a

    @staticmethod
    def backward(ctx, dpooled, _da_unused):
        lib, s = nat.load(), _stream()
        X2, a, m2, l, out, prep, prep_t = ctx.saved_tensors
        gated, dev, N = ctx.gated, X2.device, X2.shape[0]
        dp = _f32c(dpooled).reshape(-1)
        da = torch.empty(N, dtype=torch.float32, device=dev)
        nat.check(lib.vlsa_scored_pool_backward(_p(X2), _dt(X2), N, X2.stride(0), 512, _p(a), _p(m2), _p(l), _p(out), _p(dp), _p(da), s),
                  "vlsa_scored_pool_backward")
        tile_rows = int(lib.vlsa_mlp_bwd_tile_rows(_dt(X2)))
        need_dx = ctx.needs_input_grad[0]          # the bag itself carries a gradient (a trainable Feat_Projecter in front)
        dX = torch.empty(N, 512, dtype=torch.float32, device=dev) if need_dx else None
        keep, p_desc, p_dx, p_off, p_ts, n_tiles, _ = _row_tables([X2], tile_rows, extra=[dX] if need_dx else None)
        ws = torch.empty(lib.vlsa_mlp_bwd_workspace_bytes(1 if gated else 0, n_tiles), dtype=torch.uint8, device=dev)
        dW = torch.empty(2 if gated else 1, 256, 512, dtype=torch.float32, device=dev)
        dvec = torch.empty(3, 512, dtype=torch.float32, device=dev)
        nat.check(lib.vlsa_attn_scores_backward(p_desc, 1, _dt(X2), 512, _p(prep), int(gated), p_ts, n_tiles, _p(da), p_off, _p(ws),
                                                _p(dW), _p(dvec), ctx.drop[0], ctx.drop[1], s), "vlsa_attn_scores_backward")
        if need_dx:
            aw = torch.exp2(a * 1.4426950408889634 - m2[0]) / l[0]          # [N] softmax weights of the pooling
            nat.check(lib.vlsa_attn_scores_backward_dx(p_desc, p_dx, 1, _dt(X2), 512, _p(prep), _p(prep_t), int(gated), p_ts, n_tiles, _p(da),
                                                       _p(aw), _p(dp), p_off, ctx.drop[0], ctx.drop[1], s), "vlsa_attn_scores_backward_dx")
            if X2.dtype != torch.float32:
                dX = dX.to(X2.dtype)
        w2_shape, c_shape = ctx.shapes
        return (dX, None, dW[0], dvec[0, :256], dW[1] if gated else None, dvec[0, 256:] if gated else None,
                dvec[1, :256].reshape(w2_shape), dvec[2, :1].reshape(c_shape), None, None)


def attn_pool_autograd(X2: torch.Tensor, fused: "FusedAttnScores", Wa, ba, Wg, bg, w2, c, drop_p: float = 0.0, seed: Optional[int] = None):
    """(pooled [512], raw scores a [N]) of (Gated_)Attention_Pooling over a bag, differentiable w.r.t. the module's parameters AND
    the bag (see _AttnPoolFn)."""
    _need_gpu(X2)
    if drop_p and seed is None:
        seed = int(torch.randint(0, 2 ** 31 - 1, (1,)).item())
    return _AttnPoolFn.apply(X2, fused, Wa, ba, Wg, bg, w2, c, float(drop_p or 0.0), int(seed or 0))


def attn_scores_autograd(X2: torch.Tensor, fused: "FusedAttnScores", Wa, ba, Wg, bg, w2, c, drop_p: float = 0.0,
                         seed: Optional[int] = None) -> torch.Tensor:
    """a [N] of (Gated_)Attention_Pooling over a bag, differentiable w.r.t. the module's parameters (not the bag).
    drop_p > 0 (gated only): the module's training-mode dropout behind tanh and sigmoid (model/layers.py:94,99), masks from a
    counter-based generator keyed on ``seed`` (None: drawn from torch's CPU generator, so ``torch.manual_seed`` governs it)."""
    _need_gpu(X2)
    _no_bag_grad(X2)
    if drop_p and seed is None:
        seed = int(torch.randint(0, 2 ** 31 - 1, (1,)).item())
    return _AttnScoresFn.apply(X2, fused, Wa, ba, Wg, bg, w2, c, float(drop_p or 0.0), int(seed or 0))


def mean_pool_bags(bags) -> torch.Tensor:
    """Row means [B, 512] of up to 64 validated [N_i, 512] device bags (one dtype) in two launches (the 'mean' pooling of
    FeatMIL / DeepMIL over a batch: model/deepmil.py:57-58,271-272 per bag)."""
    lib, s, dev, B = nat.load(), _stream(), bags[0].device, len(bags)
    if not (1 <= B <= 64):
        raise ValueError("1..64 bags per call")
    import numpy as np
    desc = torch.from_numpy(np.asarray([(x.data_ptr(), x.shape[0], x.stride(0)) for x in bags], dtype=np.int64)).to(dev)
    G = max(1, min(64, 512 // B))
    f = lambda *sh: torch.empty(*sh, dtype=torch.float32, device=dev)  # noqa: E731
    pm, pl, pacc, m2, l, out = f(B * G, nat.P_STRIDE), f(B * G, nat.P_STRIDE), f(B * G, 512), f(B, nat.P_STRIDE), f(B, nat.P_STRIDE), f(B, 512)
    dt = nat.DT_F32 if bags[0].dtype == torch.float32 else nat.DT_BF16
    nat.check(lib.vlsa_scored_pool_partial_batch(_p(desc), B, dt, 512, None, None, G, _p(pm), _p(pl), _p(pacc), s),
              "vlsa_scored_pool_partial_batch")
    st = (ctypes.c_int64 * 9)(nat.P_STRIDE, nat.P_STRIDE, 512, G * nat.P_STRIDE, G * nat.P_STRIDE, G * 512,
                              nat.P_STRIDE, nat.P_STRIDE, 512)
    nat.check(lib.vlsa_vlfan_merge_batch_strided(_p(pm), _p(pl), _p(pacc), B, G, 1, 512, 1, st, _p(m2), _p(l), _p(out), s),
              "vlsa_vlfan_merge_batch_strided")
    return out


class FusedFeatProjecter:
    """Feat_Projecter (Linear(512, 512) + LayerNorm, model/layers.py:65-82) over all patch rows of a bf16 or fp32 bag in ONE
    MFMA kernel (vlsa_feat_project): fp32 [N, 512] out, the pre-activations never reach memory.  Holds the weights packed
    in MFMA-fragment order (bf16 hi + lo split) and re-packs them when a parameter changes.  Inference only (no dX / dW)."""

    def __init__(self):
        self._key, self._prep = None, None

    @staticmethod
    def supported(X2: torch.Tensor, linear, norm) -> bool:
        return (X2.is_cuda and X2.dim() == 2 and X2.dtype in (torch.bfloat16, torch.float32) and X2.shape[0] > 0
                and linear.in_features == 512 and linear.out_features == 512 and tuple(norm.normalized_shape) == (512,))

    def packed(self, device, W, b, gamma, beta) -> torch.Tensor:
        """the fragment-packed weights of this parameter version (re-packed when a parameter changed in place or was replaced)"""
        lib = nat.load()
        params = [t for t in (W, b, gamma, beta) if t is not None]
        key = tuple((id(t), t._version) for t in params) + (device,)
        if key != self._key:
            prep = torch.empty(lib.vlsa_featproj_prep_bytes(), dtype=torch.uint8, device=device)
            keep = [None if t is None else _f32c(t).reshape(-1) for t in (W, b, gamma, beta)]
            nat.check(lib.vlsa_prepare_featproj(*[_p(t) for t in keep], W.shape[1], W.shape[0], _p(prep), _stream()),
                      "vlsa_prepare_featproj")
            self._key, self._prep, self._params = key, prep, params      # (the parameters stay alive: their ids are the key)
        return self._prep

    def autograd(self, X2: torch.Tensor, W, b, gamma, beta, eps: float) -> torch.Tensor:
        """Y = LayerNorm(X W^T + b) under autograd w.r.t. (W, b, gamma, beta): HIP forward and backward, see _FeatProjectFn."""
        _need_gpu(X2)
        _no_bag_grad(X2)
        return _FeatProjectFn.apply(X2, self, W, b, gamma, beta, eps)

    def __call__(self, X2: torch.Tensor, W, b, gamma, beta, eps: float) -> torch.Tensor:
        lib = nat.load()
        X2 = _bag2d(X2)
        self.packed(X2.device, W, b, gamma, beta)
        N = X2.shape[0]
        Y = torch.empty(N, 512, dtype=torch.float32, device=X2.device)
        nat.check(lib.vlsa_feat_project(_p(X2), _dt(X2), N, X2.stride(0), 512, _p(self._prep), float(eps), _p(Y), 512, _stream()),
                  "vlsa_feat_project")
        return Y


class _FeatProjectFn(torch.autograd.Function):
    """Feat_Projecter (Linear + LayerNorm, model/layers.py:65-82) over all rows of a bag under autograd: forward = ONE kernel
    that also stores the per-row LayerNorm statistics (vlsa_feat_project_train), backward = a row-statistics pass over (dY, Y)
    and ONE kernel for dW, db, dgamma, dbeta (vlsa_feat_project_backward).  The bag itself carries no gradient."""

    @staticmethod
    def forward(ctx, X2, fused, W, b, gamma, beta, eps):
        lib = nat.load()
        X2 = _bag2d(X2)
        prep = fused.packed(X2.device, W, b, gamma, beta)
        N = X2.shape[0]
        Y = torch.empty(N, 512, dtype=torch.float32, device=X2.device)
        stats = torch.empty(N, 4, dtype=torch.float32, device=X2.device)
        nat.check(lib.vlsa_feat_project_train(_p(X2), _dt(X2), N, X2.stride(0), 512, _p(prep), float(eps), _p(Y), 512, _p(stats),
                                              _stream()), "vlsa_feat_project_train")
        ctx.save_for_backward(X2, Y, stats, prep)
        ctx.has = (b is not None, gamma is not None, beta is not None)
        return Y

    @staticmethod
    def backward(ctx, dY):
        lib = nat.load()
        X2, Y, stats, prep = ctx.saved_tensors
        dev, N, s = X2.device, X2.shape[0], _stream()
        dY = _f32c(dY).reshape(N, 512)
        nat.check(lib.vlsa_feat_project_rowstats(_p(dY), dY.stride(0), _p(Y), Y.stride(0), N, _p(prep), _p(stats), s),
                  "vlsa_feat_project_rowstats")
        tile_rows = int(lib.vlsa_mlp_bwd_tile_rows(_dt(X2)))
        keep, p_desc, p_dy, p_off, p_ts, n_tiles, _ = _row_tables([X2], tile_rows, extra=[dY])
        ws = torch.empty(lib.vlsa_mlp_bwd_workspace_bytes(2, n_tiles), dtype=torch.uint8, device=dev)
        dW = torch.empty(512, 512, dtype=torch.float32, device=dev)
        dvec = torch.empty(3, 512, dtype=torch.float32, device=dev)
        nat.check(lib.vlsa_feat_project_backward(p_desc, p_dy, 1, _dt(X2), _p(prep), p_ts, n_tiles, _p(stats), p_off, _p(ws), _p(dW),
                                                 _p(dvec), s), "vlsa_feat_project_backward")
        hb, hg, hbt = ctx.has
        return None, None, dW, dvec[0] if hb else None, dvec[1] if hg else None, dvec[2] if hbt else None, None


def topk_mean(S: torch.Tensor, k: int, out_scale: float = 1.0) -> torch.Tensor:
    """Per-class mean of the k largest entries of S[C, N] (k >= N: plain mean), times out_scale."""
    _need_gpu(S)
    lib = nat.load()
    S = _f32c(S)
    C, N = S.shape
    out = torch.empty(C, dtype=torch.float32, device=S.device)
    ws = torch.empty(max(4, lib.vlsa_topk_workspace_bytes(C, N, int(k))), dtype=torch.uint8, device=S.device)
    nat.check(lib.vlsa_topk_mean_ws(_p(S), C, N, int(k), float(out_scale), _p(ws), _p(out), _stream()), "vlsa_topk_mean_ws")
    return out


def normalize_many(X: torch.Tensor) -> torch.Tensor:
    """F.normalize(X.float(), dim=-1) for all N patch rows of a bag in one pass (bf16 or fp32 in, fp32 out)."""
    _need_gpu(X)
    lib = nat.load()
    X2 = _bag2d(X)
    N, D = X2.shape
    out = torch.empty(N, D, dtype=torch.float32, device=X2.device)
    nat.check(lib.vlsa_normalize_many(_p(X2), _dt(X2), N, X2.stride(0), D, _p(out), _stream()), "vlsa_normalize_many")
    return out


def class_cosines(X: torch.Tensor, T: torch.Tensor) -> torch.Tensor:
    """cos(T_k, x_n) for every class / patch: [K, N] (zero-shot logits before the logit scale; model/vlsa.py:185-192).
    Runs the streaming MFMA kernel with the K text embeddings as queries and keeps its score output."""
    _need_gpu(X, T)
    K = T.shape[0]
    outs = []
    for k0 in range(0, K, nat.MAX_P):
        qp = prepare_queries(T[k0:k0 + nat.MAX_P], False, 1.0 / 1.4426950408889634)  # scores come back as plain cosines
        _, _, _, sc = vlfan_partial(X, qp, want_scores=True)
        outs.append(sc)
    return outs[0] if len(outs) == 1 else torch.cat(outs, dim=0)


class VlfanBatchPlan:
    """B bags per launch (bf16 rows, D = 512): one persistent streaming kernel over all bags + batched merge + batched
    head = 3 launches per batch (plus the shared query / text preparation).  Bags may have different N.

    ``set_bags`` writes the device-side descriptor table (pointer, N, row stride per bag); ``run`` enqueues the batch.
    Outputs are [B, ...] tensors owned by the plan (logits [B, K], incidence, vhat [B, D], out [B, P, D], m2/l [B, 16]).
    """

    def __init__(self, B: int, P: int, K: int, device, D: int = 512, gated: bool = False, pool: str = "mean",
                 identity_head: bool = False, coattn_scale: float = COATTN_SCALE, reserved_cus: int = 0,
                 want_attn: bool = False):
        """want_attn: also produce every bag's attention weights A [P, N_i] (``attn.views`` after ``run``): the streaming
        kernel stores its scores (48 B per patch at P = 12) and one more launch normalises them in place.
        reserved_cus: compute units left without a persistent streaming workgroup.  When batches are pipelined over two
        streams, 32 (4 per XCD) lets the merge / head / prepare kernels of batch i run on those CUs while batch i+1 streams
        on the other 224: the HBM-bound streaming kernel loses ~2 %, the step gains ~4 % (0 for a single stream)."""
        lib = nat.load()
        self.reserved_cus = int(reserved_cus)
        if not (1 <= B <= lib.vlsa_batch_forward_max_bags()):
            raise ValueError(f"batch size {B} outside [1, {lib.vlsa_batch_forward_max_bags()}]")
        self.lib, self.B, self.D, self.P, self.K = lib, B, D, P, K
        self.gated, self.pool, self.identity_head, self.scale = gated, _POOL_CODES[pool], identity_head, float(coattn_scale)
        f = lambda *s: torch.empty(*s, dtype=torch.float32, device=device)  # noqa: E731
        self.desc_host = torch.zeros(B, 3, dtype=torch.int64).pin_memory() if torch.cuda.is_available() else torch.zeros(B, 3, dtype=torch.int64)
        self.desc = torch.zeros(B, 3, dtype=torch.int64, device=device)
        self.ws = torch.zeros(lib.vlsa_batch_workspace_bytes(B, P, D), dtype=torch.uint8, device=device)
        self.qprep = torch.empty(lib.vlsa_qprep_bytes(D), dtype=torch.uint8, device=device)
        self.That, self.tnorm = f(K, D), f(K)
        self.m2, self.l, self.out = f(B, nat.P_STRIDE), f(B, nat.P_STRIDE), f(B, P, D)
        self.pooled, self.v, self.vhat, self.vnorm = f(B, D), f(B, D), f(B, D), f(B)
        self.logits, self.incidence = f(B, K), f(B, K)
        self._bags = None
        self.want_attn, self.attn = bool(want_attn), None
        self.dt = nat.DT_BF16
        self.groups = 0
        self._desc_np = self.desc_host.numpy()     # same (pinned) memory, cheap element writes
        self._desc_ev = None                       # recorded after the table's H2D copy: guards the pinned staging buffer
        # ctypes pointers of the plan-owned buffers, built once (the per-call cost of the eager path is host-side)
        self._c = {k: _p(getattr(self, k)) for k in ("desc", "ws", "qprep", "That", "tnorm", "m2", "l", "out", "pooled", "v",
                                                     "vhat", "vnorm", "logits", "incidence")}

    def set_bags(self, bags, validated: bool = False):
        """bags: list of B device tensors [N_i, 512], all bf16 or all fp32 (unit inner stride, 16-byte aligned rows).
        Kept alive by the plan.  validated: the caller already passed every bag through ``_bag2d`` and checked device,
        dtype and D (skips the per-bag checks: the eager path is host-bound for small bags)."""
        if len(bags) != self.B:
            raise ValueError(f"expected {self.B} bags, got {len(bags)}")
        if isinstance(bags, BagSet) and bags.D == self.D:      # checked once, descriptor rows kept: one array assignment
            if self._desc_ev is not None:
                self._desc_ev.synchronize()
            self._desc_np[:] = bags.rows
            self._bags, self.dt = bags, bags.dt
            self.groups = bags.groups(self.reserved_cus)
            if self.want_attn and (self.attn is None or tuple(self.attn.sizes) != bags.sizes):
                self.attn = AttnBuffers(bags.sizes, self.P, self.desc.device)
            self.desc.copy_(self.desc_host, non_blocking=True)
            if self.desc_host.is_pinned():
                self._desc_ev = torch.cuda.Event()
                self._desc_ev.record()
            return
        keep, rows = [], []
        for i, x in enumerate(bags):
            if not validated:
                _need_gpu(x)
                x = _bag2d(x)
                if x.shape[1] != self.D or (i > 0 and x.dtype != keep[0].dtype):
                    raise VlsaNativeError("the batched path takes bags with D == 512 and one dtype (bf16 or fp32) per batch")
            keep.append(x)
            n = x.shape[0]
            rows.append((x.data_ptr(), n, x.stride(0) if n > 0 else self.D))
        if self._desc_ev is not None:
            self._desc_ev.synchronize()            # the previous table's async copy has read the staging buffer
        self._desc_np[:] = rows
        self._bags = keep
        self.dt = nat.DT_F32 if keep[0].dtype == torch.float32 else nat.DT_BF16
        self.groups = choose_groups([r[1] for r in rows], self.reserved_cus)  # bags in flight
        if self.want_attn and (self.attn is None or self.attn.sizes != [r[1] for r in rows]):
            self.attn = AttnBuffers([r[1] for r in rows], self.P, self.desc.device)
        self.desc.copy_(self.desc_host, non_blocking=True)
        if self.desc_host.is_pinned():
            self._desc_ev = torch.cuda.Event()
            self._desc_ev.record()

    def run(self, Q, T, logit_scale, W=None, b=None, pool_w=None, outs: Optional[dict] = None, params_key=None):
        """outs: optional {'logits': [B, K], 'vhat': [B, D], 'That': [K, D]} tensors written instead of the plan's buffers.
        params_key: a hashable that changes whenever Q or T change (parameter versions); when it equals the previous call's, the
        query / text preparation launch is skipped and the plan's prepared block is reused -- the preparation is bag-independent
        (an evaluation loop prepares once, not once per launch; outs['That'] is then filled by a copy of the plan's)."""
        lib, s, c, k = self.lib, _stream(), nat.check, self._c
        reuse = params_key is not None and params_key == getattr(self, "_params_key", None)
        self._params_key = None          # set again below, once the preparation of THIS key has been enqueued (a failed
        own_That = None                  # prepare call followed by a retry with the same key must not reuse a stale block)
        if outs:
            k = dict(k)
            for name, t in outs.items():
                if name == "That" and params_key is not None:     # the prepared text features stay in the plan for later calls
                    own_That = t
                    continue
                k[name] = _p(t)
        nq = self.P + 1 if self.gated else self.P
        if not reuse:
            c(lib.vlsa_prepare_queries_and_text(_p(Q), nq, self.D, int(self.gated), self.scale, k["qprep"], _p(T), self.K,
                                                k["That"], k["tnorm"], s), "prepare_queries_and_text")
        self._params_key = params_key
        if own_That is not None:
            own_That.copy_(self.That)
        ad = _p(self.attn.desc) if self.want_attn else None
        c(lib.vlsa_vlfan_forward_batch_attn(k["desc"], self.B, self.dt, self.D, k["qprep"], self.P, self.pool,
                                            _p(pool_w), None if self.identity_head else _p(W),
                                            None if self.identity_head else _p(b), k["That"], self.K, _p(logit_scale),
                                            k["ws"], k["m2"], k["l"], k["out"], k["pooled"], k["v"], k["vhat"], k["vnorm"],
                                            k["logits"], k["incidence"], self.reserved_cus, self.groups, ad, ad,
                                            self.attn.max_n if self.want_attn else 0, s), "vlfan_forward_batch")
        return outs["logits"] if outs and "logits" in outs else self.logits

    def capture(self, Q, T, logit_scale, W=None, b=None, pool_w=None):
        """Record one ``run`` of the current bags into a hipGraph (``torch.cuda.CUDAGraph``) and return it: ``g.replay()``
        re-issues the 5 kernels with one host call.  Everything the kernels read is referenced by address (the bag table,
        Q, T, W, b, logit_scale), so in-place updates of those tensors are seen by later replays; the outputs land in the
        plan's buffers.  For many small bags per launch the eager launch sequence, not the GPU, is the limit."""
        side = torch.cuda.Stream(device=self.desc.device)
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            self.run(Q, T, logit_scale, W, b, pool_w)      # warm-up outside the capture (module load, attribute set-up)
        torch.cuda.current_stream().wait_stream(side)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            self.run(Q, T, logit_scale, W, b, pool_w)
        self._graph_keep = (Q, T, logit_scale, W, b, pool_w)
        return g

    def run_partial_only(self):
        """Only the persistent streaming kernel (roofline timing); queries must have been prepared by a run()."""
        nat.check(self.lib.vlsa_vlfan_partial_batch_scores(_p(self.desc), self.B, self.dt, self.D, _p(self.qprep), self.P,
                                                           _p(self.ws), self.reserved_cus, self.groups,
                                                           _p(self.attn.desc) if self.want_attn else None, _stream()),
                  "vlfan_partial_batch")


def zeroshot_pool_bags(bags, T: torch.Tensor, logit_scale: torch.Tensor, k: Optional[int]) -> torch.Tensor:
    """Zero-shot bag logits [B, K] for a list of up to 64 bags (bf16 or fp32 [N_i, 512], one dtype): the K text features go
    through the persistent multi-bag streaming kernel as queries (scale 1 / log2(e): the stored scores ARE the cosines), then
    ONE launch pools every (class, bag) row (top-k mean, k clamped to N_i; None: mean).  model/vlsa.py:185-196 per bag."""
    _need_gpu(T, logit_scale, *bags)
    lib, s = nat.load(), _stream()
    K = T.shape[0]
    table = _BagTable(bags)
    B, dev = table.B, table.desc.device
    out = torch.empty(B, K, dtype=torch.float32, device=dev)
    sizes = [x.shape[0] for x in table.bags]
    ls = _f32c(logit_scale).reshape(1)
    for k0 in range(0, K, nat.MAX_P):
        Tk = T[k0:k0 + nat.MAX_P]
        Pk = Tk.shape[0]
        qp = prepare_queries(Tk, False, 1.0 / 1.4426950408889634)
        sc = AttnBuffers(sizes, Pk, dev)
        ws = torch.empty(lib.vlsa_batch_workspace_bytes(B, Pk, table.D), dtype=torch.uint8, device=dev)
        nat.check(lib.vlsa_vlfan_partial_batch_scores(_p(table.desc), B, table.dt, table.D, _p(qp.buf), Pk, _p(ws), 0,
                                                      choose_groups(sizes, 0), _p(sc.desc), s), "vlsa_vlfan_partial_batch_scores")
        part = out if (k0 == 0 and Pk == K) else torch.empty(B, Pk, dtype=torch.float32, device=dev)
        nat.check(lib.vlsa_topk_mean_batch(_p(table.desc), _p(sc.desc), B, Pk, 0 if k is None else int(k), _p(ls), _p(part), s),
                  "vlsa_topk_mean_batch")
        if part is not out:
            out[:, k0:k0 + Pk] = part
    return out


def query_pool_attention(rows: torch.Tensor, module) -> Tuple[torch.Tensor, torch.Tensor]:
    """(Gated_)Attention_Pooling over the P aggregated rows of B bags in two launches (inference): rows [B, P, D] fp32 ->
    (pooled [B, D], scores [B, P]) with the module's own return convention (raw scores for Attention_Pooling, softmax weights
    for Gated_Attention_Pooling; model/layers.py:103-122,137-153).  `module` carries the parameters."""
    _need_gpu(rows)
    lib = nat.load()
    rows = _f32c(rows)
    B, P, D = rows.shape
    gated = hasattr(module, "fc1")
    if gated:
        la, lg, l2 = module.fc1[0], module.score[0], module.fc2
        Wg, bg = _f32c(lg.weight), _f32c(lg.bias)
    else:
        la, l2 = module.attention[0], module.attention[2]
        Wg = bg = None
    hid = la.weight.shape[0]
    ws = torch.empty(B * ((hid + 3) // 4) * nat.P_STRIDE, dtype=torch.float32, device=rows.device)
    pooled = torch.empty(B, D, dtype=torch.float32, device=rows.device)
    scores = torch.empty(B, P, dtype=torch.float32, device=rows.device)
    nat.check(lib.vlsa_query_pool_attention(_p(rows), B, P, D, _p(_f32c(la.weight)), _p(_f32c(la.bias)), _p(Wg), _p(bg),
                                            _p(_f32c(l2.weight).reshape(-1)), _p(_f32c(l2.bias).reshape(-1)), hid, int(not gated),
                                            _p(ws), _p(pooled), _p(scores), _stream()), "vlsa_query_pool_attention")
    return pooled, scores
