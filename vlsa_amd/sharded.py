"""Patch-sharded multi-GPU execution of one bag: one process per GPU, rows split contiguously.

Each rank streams its shard with the same HIP kernel, folds its workgroup partials into ONE compact record
``[m2(16) | l(16) | acc(P*D)]`` (24.7 KB at P=12, D=512), the ranks exchange the records with a single RCCL all-gather
over xGMI, and every rank merges the ``world`` records and runs the (replicated) incidence head.  The softmax over the
patches is permutation invariant, so any row partition gives the single-GPU result up to fp32 summation order.
Attention weights stay sharded: each rank normalises its own scores with the global (m2, l).

The exchange is latency bound (tens of microseconds for a 25 KB collective), so ``ShardedVlfanPlan`` software-pipelines
it: the all-gather of bag i runs on a side stream while the streaming kernel of bag i+1 runs; the merge + head of bag i
is enqueued after that.  ``finish()`` drains the pipeline.
"""
from __future__ import annotations

import ctypes
from typing import Optional, Tuple

import torch

from . import _native as nat
from . import functional as VF

REC_HDR = 2 * nat.P_STRIDE  # floats of (m2, l) in front of acc in a compact record


def shard_bounds(N: int, world: int, rank: int, align: int = 16) -> Tuple[int, int]:
    """Contiguous row range of `rank`: boundaries at multiples of `align` rows, balanced to within one unit."""
    units = (N + align - 1) // align
    q, r = divmod(units, world)
    ub = rank * q + min(rank, r)
    ue = ub + q + (1 if rank < r else 0)
    return min(ub * align, N), min(ue * align, N)


def record_floats(P: int, D: int) -> int:
    return REC_HDR + P * D


def all_gather_records(record: torch.Tensor, out: torch.Tensor, group=None, async_op: bool = False):
    """One collective: every rank contributes its compact record; out is [world, record_floats]."""
    import torch.distributed as dist
    return dist.all_gather_into_tensor(out.view(-1), record.view(-1), group=group, async_op=async_op)


class ShardedVlfanPlan:
    """Fused inference forward of a patch-sharded bag; same results on every rank."""

    def __init__(self, N_local: int, D: int, P: int, K: int, device, dist_module=None, group=None, gated: bool = False,
                 pool: str = "mean", identity_head: bool = False, want_attn: bool = False, pipeline: bool = True):
        import torch.distributed as dist
        self.dist = dist_module or dist
        self.group = group
        self.world = self.dist.get_world_size(group)
        self.local = VF.VlfanInferencePlan(N_local, D, P, K, device, gated=gated, pool=pool, identity_head=identity_head,
                                           want_attn=False)
        self.P, self.D, self.K = P, D, K
        rf = record_floats(P, D)
        f = lambda *s: torch.empty(*s, dtype=torch.float32, device=device)  # noqa: E731
        # Attention weights (model/deepmil.py:198,206-215) stay sharded: A[:, shard].  With the pipeline the streaming kernel
        # of bag i+1 runs BEFORE the tail of bag i, so everything the tail reads that the streaming kernel writes is held
        # per slot -- the record, the gathered records and the raw scores; A is per slot so that a caller can still read
        # bag i's weights after bag i+1 was enqueued.
        self.want_attn = want_attn
        self.scores = [f(P, N_local), f(P, N_local)] if want_attn else [None, None]
        self.A_slots = [f(P, N_local), f(P, N_local)] if want_attn else [None, None]
        self.A = None                                   # weights of the last bag whose tail was enqueued ([P, N_local])
        self.rec = [f(rf), f(rf)]                       # double-buffered: bag i's record is in flight while i+1 computes
        self.gathered = [f(self.world, rf), f(self.world, rf)]
        self.comm_stream = torch.cuda.Stream(device=device) if pipeline else None
        self.done_local = [torch.cuda.Event(), torch.cuda.Event()]
        self.done_comm = [torch.cuda.Event(), torch.cuda.Event()]
        self.pipeline = pipeline
        self._pending: Optional[tuple] = None
        self._i = 0
        self.lib = nat.load()

    # -- pieces ----------------------------------------------------------------------------------------------
    def _local_record(self, X, Q, slot):
        pl_, lib, s = self.local, self.lib, VF._stream()
        nq = self.P + 1 if pl_.gated else self.P
        c, p = nat.check, VF._p
        c(lib.vlsa_prepare_queries(p(Q), nq, self.D, int(pl_.gated), pl_.scale, p(pl_.qprep), s), "prepare_queries")
        dt = nat.DT_F32 if X.dtype == torch.float32 else nat.DT_BF16
        c(lib.vlsa_vlfan_partial(p(X), dt, pl_.N, X.stride(0), self.D, p(pl_.qprep), self.P, pl_.kernel, p(pl_.pm),
                                 p(pl_.pl), p(pl_.pacc), p(self.scores[slot]), s), "vlfan_partial")
        rec = self.rec[slot]
        # fold the workgroup partials into the compact record in place: m2 -> rec[0:16], l -> rec[16:32], acc -> rec[32:]
        c(lib.vlsa_vlfan_merge(p(pl_.pm), p(pl_.pl), p(pl_.pacc), pl_.G, self.P, self.D, 0, p(rec),
                               ctypes.c_void_p(rec.data_ptr() + 4 * nat.P_STRIDE),
                               ctypes.c_void_p(rec.data_ptr() + 4 * REC_HDR), s), "vlfan_merge(local)")

    def _tail(self, slot, T, ls, W, b, pool_w):
        pl_, lib, s = self.local, self.lib, VF._stream()
        c, p = nat.check, VF._p
        g = self.gathered[slot]
        rf = record_floats(self.P, self.D)
        base = g.data_ptr()
        c(lib.vlsa_vlfan_merge_strided(ctypes.c_void_p(base), rf, ctypes.c_void_p(base + 4 * nat.P_STRIDE), rf,
                                       ctypes.c_void_p(base + 4 * REC_HDR), rf, self.world, self.P, self.D, 1,
                                       p(pl_.m2), p(pl_.l), p(pl_.out), s), "vlfan_merge(global)")
        if self.want_attn:
            c(lib.vlsa_attn_normalise(p(self.scores[slot]), self.P, pl_.N, p(pl_.m2), p(pl_.l), p(self.A_slots[slot]), s),
              "attn_normalise")
            self.A = self.A_slots[slot]
        c(lib.vlsa_normalize_rows(p(T), self.K, self.D, p(pl_.That), p(pl_.tnorm), s), "normalize_rows")
        c(lib.vlsa_head_forward(p(pl_.out), self.P, self.D, pl_.pool, p(pool_w), None if pl_.identity_head else p(W),
                                None if pl_.identity_head else p(b), p(pl_.That), self.K, p(ls), p(pl_.ws), p(pl_.pooled),
                                p(pl_.v), p(pl_.vhat), p(pl_.vnorm), p(pl_.logits), p(pl_.incidence), s), "head_forward")

    # -- driver ----------------------------------------------------------------------------------------------
    def run(self, X_local, Q, T, logit_scale, W=None, b=None, pool_w=None):
        """Enqueue one bag. With pipeline=True the logits (``local.logits``) and attention weights (``A``, this rank's
        columns) of THIS bag are valid after the next run() or finish(); ``A`` stays intact until two more bags are run."""
        slot = self._i & 1
        self._i += 1
        cur = torch.cuda.current_stream()
        if self.pipeline:
            cur.wait_event(self.done_comm[slot])      # the gather buffers of bag i-2 are free again
        self._local_record(X_local, Q, slot)
        if not self.pipeline:
            all_gather_records(self.rec[slot], self.gathered[slot], self.group)
            self._tail(slot, T, logit_scale, W, b, pool_w)
            return self.local.logits
        self.done_local[slot].record(cur)
        with torch.cuda.stream(self.comm_stream):
            self.comm_stream.wait_event(self.done_local[slot])
            all_gather_records(self.rec[slot], self.gathered[slot], self.group)
            self.done_comm[slot].record(self.comm_stream)
        if self._pending is not None:
            self._drain()
        self._pending = (slot, T, logit_scale, W, b, pool_w)
        return self.local.logits

    def _drain(self):
        slot, T, ls, W, b, pw = self._pending
        torch.cuda.current_stream().wait_event(self.done_comm[slot])
        self._tail(slot, T, ls, W, b, pw)
        self._pending = None

    def finish(self):
        if self._pending is not None:
            self._drain()
        return self.local.logits


def sharded_vlfan_forward(X_local: torch.Tensor, Q: torch.Tensor, gated: bool = False, group=None,
                          want_attn: bool = False, coattn_scale: float = VF.COATTN_SCALE):
    """Functional, unpipelined form: returns (out[P, D] identical on every rank, A_local[P, N_local] or None)."""
    import torch.distributed as dist
    world = dist.get_world_size(group)
    qp = VF.prepare_queries(Q, gated, coattn_scale)
    pm, pl, pacc, scores = VF.vlfan_partial(X_local, qp, want_scores=want_attn)
    m2, l, acc = VF.vlfan_merge(pm, pl, pacc, normalise=False)
    rec = torch.cat([m2, l, acc.reshape(-1)])
    gathered = torch.empty(world, rec.numel(), dtype=torch.float32, device=rec.device)
    all_gather_records(rec, gathered, group)
    P, D = acc.shape
    m2g, lg, out = VF.vlfan_merge(gathered[:, :nat.P_STRIDE].contiguous(), gathered[:, nat.P_STRIDE:REC_HDR].contiguous(),
                                  gathered[:, REC_HDR:].reshape(world, P, D).contiguous(), normalise=True)
    A = VF.attn_normalise(scores, m2g, lg) if want_attn else None
    return out, A


class ShardedVlfanBatchPlan:
    """B patch-sharded bags per launch: the persistent streaming kernel walks this rank's shard of every bag, the
    workgroup partials are folded into B compact records, ONE all-gather moves ``world x B x 24.7 KB``, then a strided
    merge over the ranks and the batched head (replicated).  The collective of batch i overlaps the streaming kernel of
    batch i+1 (side stream); ``finish()`` drains.  Results (``logits [B, K]`` ...) are identical on every rank."""

    G = 256

    def __init__(self, B: int, P: int, K: int, device, dist_module=None, group=None, D: int = 512, gated: bool = False,
                 pool: str = "mean", identity_head: bool = False, pipeline: bool = True, reserved_cus: Optional[int] = None,
                 want_attn: bool = False):
        """want_attn: every rank also gets ITS columns of each bag's attention weights, ``A`` = list of [P, N_local_i]
        (valid after the next run() / finish(), like the logits; held per pipeline slot).
        reserved_cus: compute units the persistent streaming kernel leaves free so that the RCCL all-gather of the
        previous batch and the tail kernels can run next to it (default: 32 = four per XCD when pipelined, else 0;
        env ``VLSA_RESERVED_CUS`` overrides)."""
        import os
        import torch.distributed as dist
        self.dist = dist_module or dist
        self.group = group
        self.world = self.dist.get_world_size(group)
        self.local = VF.VlfanBatchPlan(B, P, K, device, D=D, gated=gated, pool=pool, identity_head=identity_head)
        self.B, self.P, self.K, self.D = B, P, K, D
        self.rf = record_floats(P, D)
        f = lambda *s: torch.empty(*s, dtype=torch.float32, device=device)  # noqa: E731
        self.rec = [f(B, self.rf), f(B, self.rf)]
        self.gathered = [f(self.world, B, self.rf), f(self.world, B, self.rf)]
        self.comm_stream = torch.cuda.Stream(device=device) if pipeline else None
        self.done_local = [torch.cuda.Event(), torch.cuda.Event()]
        self.done_comm = [torch.cuda.Event(), torch.cuda.Event()]
        self.pipeline = pipeline
        self.skip_exchange = False    # measurement aid (bench.py): leave the collective out -- timing of the local work only, results invalid
        self._pending = None
        self._i = 0
        self.lib = nat.load()
        self.want_attn, self.attn, self.A = bool(want_attn), [None, None], None
        if reserved_cus is None:
            reserved_cus = 32 if pipeline else 0
        self.reserved_cus = int(os.environ.get("VLSA_RESERVED_CUS", reserved_cus))
        self.local.reserved_cus = self.reserved_cus   # the local plan picks the bags in flight for the same workgroup count
        rf = self.rf
        self._st_global = (ctypes.c_int64 * 9)(B * rf, B * rf, B * rf, rf, rf, rf, nat.P_STRIDE, nat.P_STRIDE, P * D)
        self._set_groups(0)

    def _set_groups(self, groups):
        P, D, rf = self.P, self.D, self.rf
        self.G = G = int(self.lib.vlsa_batch_partials_per_bag_ex(self.B, self.reserved_cus, groups))
        self._st_local = (ctypes.c_int64 * 9)(nat.P_STRIDE, nat.P_STRIDE, P * D, G * nat.P_STRIDE, G * nat.P_STRIDE,
                                              G * P * D, rf, rf, rf)

    def set_bags(self, local_shards):
        self.local.set_bags(local_shards)
        self._set_groups(self.local.groups)
        if self.want_attn:   # one score / weight buffer per pipeline slot: batch i+1 streams before batch i's tail runs
            sizes = [x.shape[0] for x in self.local._bags]
            if self.attn[0] is None or self.attn[0].sizes != sizes:   # a pending batch keeps its own buffers (see run)
                self.attn = [VF.AttnBuffers(sizes, self.P, self.local.desc.device) for _ in range(2)]

    def _local(self, Q, slot, ab=None):
        pl_, lib, s, c, p = self.local, self.lib, VF._stream(), nat.check, VF._p
        nq = self.P + 1 if pl_.gated else self.P
        c(lib.vlsa_prepare_queries(p(Q), nq, self.D, int(pl_.gated), pl_.scale, p(pl_.qprep), s), "prepare_queries")
        c(lib.vlsa_vlfan_partial_batch_scores(p(pl_.desc), self.B, pl_.dt, self.D, p(pl_.qprep), self.P, p(pl_.ws),
                                              self.reserved_cus, pl_.groups,
                                              p(ab.desc) if ab is not None else None, s), "vlfan_partial_batch")
        base = pl_.ws.data_ptr()
        n_ml = self.B * self.G * nat.P_STRIDE * 4
        rec = self.rec[slot].data_ptr()
        c(lib.vlsa_vlfan_merge_batch_strided(ctypes.c_void_p(base), ctypes.c_void_p(base + n_ml),
                                             ctypes.c_void_p(base + 2 * n_ml), self.B, self.G, self.P, self.D, 0,
                                             self._st_local, ctypes.c_void_p(rec), ctypes.c_void_p(rec + 4 * nat.P_STRIDE),
                                             ctypes.c_void_p(rec + 4 * REC_HDR), s), "merge_batch(local)")

    def _tail(self, slot, T, ls, W, b, pool_w, ab=None):
        pl_, lib, s, c, p = self.local, self.lib, VF._stream(), nat.check, VF._p
        g = self.gathered[slot].data_ptr()
        c(lib.vlsa_normalize_rows(p(T), self.K, self.D, p(pl_.That), p(pl_.tnorm), s), "normalize_rows")
        c(lib.vlsa_vlfan_merge_head_batch_strided(ctypes.c_void_p(g), ctypes.c_void_p(g + 4 * nat.P_STRIDE),
                                                  ctypes.c_void_p(g + 4 * REC_HDR), self.B, self.world, self.P, self.D,
                                                  self._st_global, pl_.pool, p(pool_w),
                                                  None if pl_.identity_head else p(W), None if pl_.identity_head else p(b),
                                                  p(pl_.That), self.K, p(ls), p(pl_.m2), p(pl_.l), p(pl_.out), p(pl_.pooled),
                                                  p(pl_.v), p(pl_.vhat), p(pl_.vnorm), p(pl_.logits), p(pl_.incidence), s),
          "merge_head_batch(global)")
        if ab is not None:   # this rank's columns, normalised with the GLOBAL (m2, l) of the merged records
            c(lib.vlsa_attn_normalise_batch(p(ab.ndesc), self.B, self.P, ab.max_n, p(ab.desc), p(pl_.m2), p(pl_.l), p(ab.desc), s),
              "attn_normalise_batch")
            self.A = ab.views

    def run(self, Q, T, logit_scale, W=None, b=None, pool_w=None):
        slot = self._i & 1
        self._i += 1
        cur = torch.cuda.current_stream()
        if self.pipeline:
            cur.wait_event(self.done_comm[slot])
        ab = self.attn[slot] if self.want_attn else None
        self._local(Q, slot, ab)
        if not self.pipeline:
            if not self.skip_exchange:
                all_gather_records(self.rec[slot], self.gathered[slot], self.group)
            self._tail(slot, T, logit_scale, W, b, pool_w, ab)
            return self.local.logits
        self.done_local[slot].record(cur)
        with torch.cuda.stream(self.comm_stream):
            self.comm_stream.wait_event(self.done_local[slot])
            if not self.skip_exchange:
                all_gather_records(self.rec[slot], self.gathered[slot], self.group)
            self.done_comm[slot].record(self.comm_stream)
        if self._pending is not None:
            self._drain()
        self._pending = (slot, T, logit_scale, W, b, pool_w, ab)
        return self.local.logits

    def _drain(self):
        slot, T, ls, W, b, pw, ab = self._pending
        torch.cuda.current_stream().wait_event(self.done_comm[slot])
        self._tail(slot, T, ls, W, b, pw, ab)
        self._pending = None

    def finish(self):
        if self._pending is not None:
            self._drain()
        return self.local.logits


# ------------------------------------------------------------------------------------------------------------------
# The other encoders, patch-sharded: zero-shot top-k pooling and (gated-)attention pooling over N
# ------------------------------------------------------------------------------------------------------------------
def gather_rows(local: torch.Tensor, group=None) -> torch.Tensor:
    """One all-gather of equal-sized per-rank tensors: [..] -> [world, ..] (identical on every rank)."""
    import torch.distributed as dist
    world = dist.get_world_size(group)
    out = torch.empty((world,) + tuple(local.shape), dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(out.view(-1), local.contiguous().view(-1), group=group)
    return out


def merge_topk_candidates(gathered: torch.Tensor) -> torch.Tensor:
    """[world, C, k] per-rank winners -> [C, world * k] candidate rows for the final re-selection."""
    world, C, k = gathered.shape
    return gathered.permute(1, 0, 2).reshape(C, world * k).contiguous()


def sharded_zeroshot_logits(X_local: torch.Tensor, T: torch.Tensor, logit_scale: torch.Tensor, pooling: str, N_total: int,
                            group=None) -> torch.Tensor:
    """Zero-shot bag logits [1, K] of a patch-sharded bag (model/vlsa.py:188-196 with the identity FeatMIL encoder +
    logit_pooling, model/deepmil.py:16-37).  Every rank scores ITS patches against the K text features in the streaming
    kernel, keeps the k largest cosines per class (or the per-class sum for 'logit_mean'), ONE all-gather moves
    ``world x K x k`` floats, every rank re-selects the k winners and averages them.  Identical on every rank."""
    from .deepmil import _parse_logit_pooling
    import torch.distributed as dist
    lib = nat.load()
    topk = _parse_logit_pooling(pooling)
    K = T.shape[0]
    X2 = VF._bag2d(X_local)
    n_loc = X2.shape[0]
    dev = X2.device
    k = N_total if topk is None else min(topk, N_total)
    scale = logit_scale.detach().float().exp()
    if n_loc > 0:
        cos = VF.class_cosines(X2, T.detach())                                    # [K, n_loc]
    if k >= N_total:                                                              # mean over every patch of the bag
        part = cos.sum(dim=1) if n_loc > 0 else torch.zeros(K, device=dev)
        tot = gather_rows(part, group).sum(dim=0)
        return (scale * tot / float(N_total))[None, :]
    vals = torch.full((K, k), float("-inf"), dtype=torch.float32, device=dev)   # an empty shard contributes no candidate
    if n_loc > 0:
        ws = torch.empty(max(4, lib.vlsa_topk_workspace_bytes(K, n_loc, k)), dtype=torch.uint8, device=dev)
        nat.check(lib.vlsa_topk_values(VF._p(cos), K, n_loc, k, VF._p(ws), VF._p(vals), VF._stream()), "vlsa_topk_values")
    cand = merge_topk_candidates(gather_rows(vals, group))                        # [K, world * k], >= k finite entries
    return (scale * VF.topk_mean(cand, k))[None, :]


def sharded_scored_pool(X_local: torch.Tensor, scores_local: Optional[torch.Tensor], group=None):
    """softmax_N(scores) @ X over a patch-sharded bag (model/layers.py:115-116,146-147; scores None: the mean over N,
    model/deepmil.py:57-58,271-272): the local online-softmax partial is folded into one record [m | l | acc(D)], ONE
    all-gather, log-sum-exp merge.  Returns (pooled [D] identical on every rank, (m2, l) global normalisers [16] each --
    this rank's attention weights are exp2(scores_local * log2(e) - m2[0]) / l[0])."""
    import torch.distributed as dist
    X2 = VF._bag2d(X_local)
    D = X2.shape[1]
    dev = X2.device
    if X2.shape[0] > 0:
        lib = nat.load()
        N = X2.shape[0]
        G = int(lib.vlsa_pool_num_partials(N))
        pm = torch.empty(G, nat.P_STRIDE, dtype=torch.float32, device=dev)
        pl = torch.empty(G, nat.P_STRIDE, dtype=torch.float32, device=dev)
        pacc = torch.empty(G, 1, D, dtype=torch.float32, device=dev)
        sc = None if scores_local is None else VF._f32c(scores_local).reshape(-1)
        nat.check(lib.vlsa_scored_pool_partial(VF._p(X2), VF._dt(X2), N, X2.stride(0), D, VF._p(sc), VF._p(pm), VF._p(pl), VF._p(pacc),
                                               VF._stream()), "vlsa_scored_pool_partial")
        m2, l, acc = VF.vlfan_merge(pm, pl, pacc, normalise=False)
    else:   # an empty shard: the neutral element of the merge
        m2 = torch.full((nat.P_STRIDE,), float("-inf"), device=dev)
        l = torch.zeros(nat.P_STRIDE, device=dev)
        acc = torch.zeros(1, D, device=dev)
    rec = torch.cat([m2, l, acc.reshape(-1)])
    g = gather_rows(rec, group)
    world = g.shape[0]
    m2g, lg, out = VF.vlfan_merge(g[:, :nat.P_STRIDE].contiguous(), g[:, nat.P_STRIDE:REC_HDR].contiguous(),
                                  g[:, REC_HDR:].reshape(world, 1, D).contiguous(), normalise=True)
    return out[0], (m2g, lg)


def sharded_deepmil_forward(enc, X_local: torch.Tensor, group=None, ret_with_attn: bool = False):
    """``DeepMIL.forward`` (model/deepmil.py:261-292) on a patch-sharded bag in eval mode: the raw (gated-)attention scores of
    this rank's patches (fused MFMA kernel for bf16 bags), the sharded softmax pooling above, the replicated head.  Returns the
    bag vector [1, C] (identical on every rank) and, if asked, this rank's columns of what the reference returns as attention
    (raw scores for 'attention', softmax weights for 'gated_attention')."""
    import torch.distributed as dist
    from .layers import Attention_Pooling
    if enc.feat_proj is not None:
        X_local = enc.feat_proj(X_local)
    X2 = VF._bag2d(X_local)
    attn = None
    with torch.no_grad():
        if enc.sigma == "max":
            loc = VF.colmax(X2) if X2.shape[0] > 0 else torch.full((X2.shape[1],), float("-inf"), device=X2.device)
            feat = gather_rows(loc, group).max(dim=0).values
        elif enc.sigma == "mean":
            feat, _ = sharded_scored_pool(X2, None, group)
        else:
            a = enc._attention_scores(X2) if X2.shape[0] > 0 else torch.empty(0, device=X2.device)
            feat, (m2, l) = sharded_scored_pool(X2, a, group)
            if ret_with_attn:
                attn = a if isinstance(enc.sigma, Attention_Pooling) else torch.exp2(a * 1.4426950408889634 - m2[0]) / l[0]
        out_feat = feat[None, :]
        if enc.pred_head == "Adapter":
            logit = VF.adapter_head(out_feat, enc.visual_adapter.fc[0].weight, enc.visual_adapter.fc[2].weight, enc.keep_ratio)[None, :]
        else:
            logit = enc.g(out_feat)
    return (logit, attn[None, :] if attn is not None else None) if ret_with_attn else logit
