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
import os
from typing import Optional, Tuple

import torch

from . import _native as nat
from . import functional as VF
from ._native import VlsaNativeError

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


def owner_partition(B: int, world: int):
    """Bag-owner partition of a batch: bag b is OWNED (merged + headed) by rank ``b % world``.  Returns (perm, counts, starts):
    ``perm[t]`` = global index of the bag at local position t in owner-major order (rank 0's bags first, then rank 1's ...),
    ``counts[o]`` = bags rank o owns, ``starts[o]`` = first local position of rank o's bags."""
    perm = [b for o in range(world) for b in range(o, B, world)]
    counts = [(B - o + world - 1) // world for o in range(world)]
    starts = [0] * world
    for o in range(1, world):
        starts[o] = starts[o - 1] + counts[o - 1]
    return perm, counts, starts


class PeerBuffers:
    """This rank's exported exchange buffer + the mapped buffers of its peers (csrc/xchg.hip).  Layout of rank o's buffer, per
    pipeline slot s in {0, 1} (o owns ``counts[o]`` bags):

        flags  [4 kinds][2 slots][16 peers] uint32     rec (records from r landed) | res (results of owner r landed) |
                                                        ack_rec (owner r consumed my records) | ack_res (r consumed my results)
        inbox  [2 slots][world][counts[o]][record_floats]   records of o's bags, one block per source rank
        resbox [2 slots][world][result_floats]              every owner's results

    Handles travel ONCE through ``all_gather_object`` of the control group (set-up, not data path)."""

    FLAG_BYTES = 4 * 2 * 16 * 4

    def __init__(self, dist_module, group, rank: int, world: int, counts, rf: int, F: int):
        lib = nat.load()
        if world > int(lib.vlsa_xchg_max_peers()):
            raise VlsaNativeError(f"peer-write exchange: at most {lib.vlsa_xchg_max_peers()} ranks")
        self.lib, self.rank, self.world, self.counts, self.rf, self.F = lib, rank, world, list(counts), rf, F
        self.bytes = self.layout_bytes(rank)
        self.own, self.kind, self.base, self._opened = 0, -1, [0] * world, []
        # Every rank walks the SAME sequence of control-plane collectives whatever fails locally (a rank that raised on its own would
        # leave the others inside a collective it never joins): export, gather (handle or the error), map, gather the verdicts, and
        # only then raise -- on every rank together.
        ptr, kind = ctypes.c_void_p(), ctypes.c_int(-1)
        handle = ctypes.create_string_buffer(64)
        rc = lib.vlsa_xchg_alloc(self.bytes, ctypes.byref(ptr), handle, ctypes.byref(kind))
        if rc == 0:
            self.own, self.kind = int(ptr.value), int(kind.value)
            mine = (bytes(handle.raw), self.kind, os.getpid())
        else:
            mine = (None, -1, f"vlsa_xchg_alloc: {lib.vlsa_error_string(rc).decode()} ({rc})")
        infos = [None] * world
        dist_module.all_gather_object(infos, mine, group=group)
        err = None
        bad = [f"rank {o}: {p_}" for o, (h, _k, p_) in enumerate(infos) if h is None]
        if bad:
            err = "peer-write exchange: no exportable device buffer on " + "; ".join(bad)
        else:
            self.kinds = [k for _h, k, _p in infos]
            if min(self.kinds) == 0 and os.environ.get("VLSA_XCHG_ALLOW_COARSE") != "1":
                # plain device memory is coherent at kernel boundaries only: a peer's stores need not be visible to a kernel that is
                # already spinning on the flag -- refuse (the caller falls back to a collective) rather than risk stale records
                err = ("peer-write exchange: the driver gave no uncached / fine-grained device memory to rank(s) "
                       f"{[i for i, k in enumerate(self.kinds) if k == 0]}")
        if err is None:
            for o, (h, _k, _pid) in enumerate(infos):
                if o == rank:
                    self.base[o] = self.own
                    continue
                q = ctypes.c_void_p()
                rc = lib.vlsa_xchg_open(ctypes.create_string_buffer(h, 64), ctypes.byref(q))
                if rc != 0:
                    err = f"vlsa_xchg_open(rank {o}'s buffer) on rank {rank}: {lib.vlsa_error_string(rc).decode()} ({rc})"
                    break
                self.base[o] = int(q.value)
                self._opened.append(int(q.value))
        verdicts = [None] * world
        dist_module.all_gather_object(verdicts, err, group=group)
        errs = [e for e in verdicts if e]
        if errs:
            self.close()
            raise VlsaNativeError(errs[0] if len(set(errs)) == 1 else "; ".join(sorted(set(errs))))

    def inbox_floats(self, o: int) -> int:
        return self.world * self.counts[o] * self.rf

    def layout_bytes(self, o: int) -> int:
        return self.FLAG_BYTES + 4 * 2 * (self.inbox_floats(o) + self.world * self.F)

    # addresses inside rank o's buffer
    def flag(self, o: int, kind: int, slot: int, peer: int) -> int:
        return self.base[o] + 4 * ((kind * 2 + slot) * 16 + peer)

    def inbox(self, o: int, slot: int, src: int) -> int:
        return self.base[o] + self.FLAG_BYTES + 4 * (slot * self.inbox_floats(o) + src * self.counts[o] * self.rf)

    def resbox(self, o: int, slot: int, owner: int) -> int:
        return self.base[o] + self.FLAG_BYTES + 4 * (2 * self.inbox_floats(o) + (slot * self.world + owner) * self.F)

    def close_peers(self):
        """unmap the OTHER ranks' buffers (local, safe at any time: this rank stops writing into them)"""
        for q in self._opened:
            self.lib.vlsa_xchg_close(ctypes.c_void_p(q))
        self._opened = []

    def close(self):
        """unmap the peers AND free this rank's exported buffer.  The caller must have made sure that no peer still writes into it:
        ``ShardedVlfanBatchPlan.close`` (device sync + barrier) is the collective way to get here."""
        self.close_peers()
        if self.own:
            self.lib.vlsa_xchg_free(ctypes.c_void_p(self.own))
            self.own = 0

    def __del__(self):
        # Garbage collection is not a collective: peers may still be writing into this rank's exported buffer, so only the peer
        # mappings go here; the buffer itself is released by close() or with the process.
        try:
            self.close_peers()
        except Exception:
            pass


REC, RES, ACK_REC, ACK_RES = 0, 1, 2, 3          # flag kinds of PeerBuffers
EXCHANGES = ("owner", "ipc", "allgather")


class ShardedVlfanBatchPlan:
    """B patch-sharded bags per launch: the persistent streaming kernel walks this rank's shard of every bag and the workgroup
    partials are folded into B compact records ``[m2 | l | acc]`` (24.7 KB at P = 12).  What happens next is the ``exchange``:

    ``"owner"`` (default): bag b is owned by rank ``b % world``.  ONE ``all_to_all_single`` hands every owner the ``world`` records of
        its bags (this rank sends ``(world - 1) / world x B x 24.7 KB`` and receives as much: 1.4 MB at B = 64, world = 8, where the
        all-gather moved 12.6 MB into every rank); the owner folds them, runs the head for ITS bags only (merge + head work
        / world), and ONE all-gather of the packed results ``[logits | incidence | v^ | m2 | l]`` (2.3 KB per bag) gives every rank
        every bag's outputs -- the global (m2, l) included, so the attention weights stay sharded as before.
    ``"ipc"``: the same data flow without a collective library on the data path: every rank exports one fine-grained buffer through
        hipIpc at construction (``PeerBuffers``), a put kernel stores the records straight into the owners' inboxes over xGMI and
        raises an epoch flag, the owner waits on its flags, folds, and puts its results into every peer's result box
        (csrc/xchg.hip).  Back-pressure through acknowledgement flags; every wait is bounded and reports through ``status()``.
    ``"allgather"``: round 1-4's exchange -- every rank receives all ``world x B`` records and merges all bags redundantly.

    The exchange of batch i overlaps the streaming kernel of batch i+1 (``pipeline``): results of a ``run`` are valid after the
    next ``run`` or ``finish()``.  ``logits [B, K]``, ``incidence``, ``vhat [B, D]``, ``m2`` / ``l [B, 16]`` are identical on every rank
    and in the caller's bag order; ``A`` (``want_attn``) = this rank's columns of every bag's attention weights."""

    G = 256

    def __init__(self, B: int, P: int, K: int, device, dist_module=None, group=None, D: int = 512, gated: bool = False,
                 pool: str = "mean", identity_head: bool = False, pipeline: bool = True, reserved_cus: Optional[int] = None,
                 want_attn: bool = False, exchange: str = "owner", timeout_s: float = 5.0):
        """want_attn: every rank also gets ITS columns of each bag's attention weights, ``A`` = list of [P, N_local_i]
        (valid after the next run() / finish(), like the logits; held per pipeline slot).
        reserved_cus: compute units the persistent streaming kernel leaves free so that the exchange and the tail kernels of the
        previous batch can run next to it (default: 32 = four per XCD when pipelined, else 0; env ``VLSA_RESERVED_CUS`` overrides).
        timeout_s: bound of every flag wait of the ``"ipc"`` exchange (a peer that never arrives sets a ``status()`` bit)."""
        import torch.distributed as dist
        if exchange not in EXCHANGES:
            raise ValueError(f"exchange must be one of {EXCHANGES}")
        self.dist = dist_module or dist
        self.group = group
        self.world = self.dist.get_world_size(group)
        self.rank = self.dist.get_rank(group)
        self.exchange = exchange
        self.local = VF.VlfanBatchPlan(B, P, K, device, D=D, gated=gated, pool=pool, identity_head=identity_head)
        self.B, self.P, self.K, self.D = B, P, K, D
        self.rf = record_floats(P, D)
        self.lib = nat.load()
        f = lambda *s: torch.empty(*s, dtype=torch.float32, device=device)  # noqa: E731
        self.rec = [f(B, self.rf), f(B, self.rf)]
        self.comm_stream = torch.cuda.Stream(device=device) if (pipeline and exchange != "ipc") else None
        ev = lambda: [torch.cuda.Event(), torch.cuda.Event()]  # noqa: E731
        self.done_local, self.done_comm, self.done_tail, self.done_res = ev(), ev(), ev(), ev()
        self.pipeline = pipeline
        self.skip_exchange = False    # measurement aid (bench.py): leave the exchange out -- timing of the local work only, results invalid
        self._pending = None
        self._i = 0
        self.want_attn, self.attn, self.A = bool(want_attn), [None, None], None
        if reserved_cus is None:
            reserved_cus = 32 if pipeline else 0
        self.reserved_cus = int(os.environ.get("VLSA_RESERVED_CUS", reserved_cus))
        self.local.reserved_cus = self.reserved_cus   # the local plan picks the bags in flight for the same workgroup count
        rf, world = self.rf, self.world
        self.perm, self.counts, self.starts = owner_partition(B, world)
        self.n_me, self.nmax = self.counts[self.rank], max(self.counts)
        if exchange == "allgather":
            self.perm = list(range(B))
            self.gathered = [f(world, B, rf), f(world, B, rf)]
            self._st_global = (ctypes.c_int64 * 9)(B * rf, B * rf, B * rf, rf, rf, rf, nat.P_STRIDE, nat.P_STRIDE, P * D)
            self.logits, self.incidence, self.vhat, self.m2, self.l = (self.local.logits, self.local.incidence, self.local.vhat,
                                                                       self.local.m2, self.local.l)
        else:
            n = self.n_me
            offs = (ctypes.c_int64 * 5)()
            self.F = int(self.lib.vlsa_xchg_result_floats(self.nmax, K, D, offs))
            self._res_off = [int(x) for x in offs]
            self.res = [torch.zeros(self.F, dtype=torch.float32, device=device) for _ in range(2)]
            # the owner's fold: `world` records per owned bag; between the source ranks n * rf floats, between bags rf
            self._st_global = (ctypes.c_int64 * 9)(n * rf, n * rf, n * rf, rf, rf, rf, nat.P_STRIDE, nat.P_STRIDE, P * D)
            self.logits, self.incidence, self.vhat = f(B, K), f(B, K), f(B, D)
            self.m2, self.l = f(B, nat.P_STRIDE), f(B, nat.P_STRIDE)                    # global (m2, l), caller's bag order
            self._ml_local = (f(B, nat.P_STRIDE), f(B, nat.P_STRIDE))                  # ... and in the local (owner-major) order
            self._counts_c = (ctypes.c_int * world)(*self.counts)
            self._starts_c = (ctypes.c_int * world)(*self.starts)
            self.status_word = torch.zeros(1, dtype=torch.int32, device=device)
            self.timeout_ticks = int(timeout_s * 100e6)
            if exchange == "owner":
                self.recv = [f(world, max(n, 1), rf), f(world, max(n, 1), rf)]
                self.resg = [f(world, self.F), f(world, self.F)]
                self._in_splits = [c * rf for c in self.counts]
                self._out_splits = [n * rf] * world
                self.peers = None
            else:
                self.peers = PeerBuffers(self.dist, group, self.rank, world, self.counts, rf, self.F)
        self._inv = [0] * B
        for t, b in enumerate(self.perm):
            self._inv[b] = t
        self._set_groups(0)

    # -- bookkeeping -----------------------------------------------------------------------------------------------------
    def exchange_bytes(self) -> dict:
        """bytes this rank sends / receives per launch over the links (its own share stays local)"""
        w, rf, B = self.world, self.rf, self.B
        if self.exchange == "allgather":
            return {"sent": 4 * B * rf * (w - 1), "received": 4 * B * rf * (w - 1), "what": "all-gather of B records"}
        rec_out = 4 * rf * (B - self.n_me)
        rec_in = 4 * rf * self.n_me * (w - 1)
        res = 4 * self.F * (w - 1)
        return {"sent": rec_out + res, "received": rec_in + res,
                "what": f"records to their owners ({rec_out} B out / {rec_in} B in) + packed results of {self.nmax} bags per owner ({res} B each way)"}

    def status(self) -> int:
        """time-out bits of THIS rank's flag waits of the "ipc" exchange (0 = every wait was served; syncs the stream)"""
        return int(self.status_word.item()) if self.exchange != "allgather" else 0

    def status_all(self) -> int:
        """COLLECTIVE: the OR of every rank's time-out bits.  A sender whose gate timed out sends nothing (csrc/xchg.hip), so the owner
        of those bags times out in turn and its merge is void -- but a third rank that only RECEIVES that owner's results has a clean
        word of its own.  Results of a launch may be used once this is 0 on every rank (`finish(check=True)` raises otherwise)."""
        if self.exchange != "ipc" or self.world == 1:
            return self.status()
        t = torch.tensor([self.status()], dtype=torch.int32)
        ctl = self.group
        if self.dist.get_backend(ctl) == "nccl":
            t = t.to(self.status_word.device)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX, group=ctl)      # (bits: MAX of small masks loses which bit, not whether)
        return int(t.item())

    def _set_groups(self, groups):
        P, D, rf = self.P, self.D, self.rf
        self.G = G = int(self.lib.vlsa_batch_partials_per_bag_ex(self.B, self.reserved_cus, groups))
        self._st_local = (ctypes.c_int64 * 9)(nat.P_STRIDE, nat.P_STRIDE, P * D, G * nat.P_STRIDE, G * nat.P_STRIDE,
                                              G * P * D, rf, rf, rf)

    def set_bags(self, local_shards):
        """local_shards: this rank's rows of every bag, in the caller's bag order"""
        if len(local_shards) != self.B:
            raise ValueError(f"expected {self.B} bags, got {len(local_shards)}")
        self.local.set_bags([local_shards[b] for b in self.perm])
        self._set_groups(self.local.groups)
        if self.want_attn:   # one score / weight buffer per pipeline slot: batch i+1 streams before batch i's tail runs
            sizes = [x.shape[0] for x in self.local._bags]
            if self.attn[0] is None or self.attn[0].sizes != sizes:   # a pending batch keeps its own buffers (see run)
                self.attn = [VF.AttnBuffers(sizes, self.P, self.local.desc.device) for _ in range(2)]

    # -- stages ------------------------------------------------------------------------------------------------------------
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

    def _put_records(self, slot, epoch):
        """"ipc": this rank's records of owner d's bags -> d's inbox (block d of ONE launch), d's REC flag for this rank"""
        pb, w, r, rf = self.peers, self.world, self.rank, self.rf
        V = ctypes.c_void_p * w
        base = self.rec[slot].data_ptr()
        src = V(*[base + 4 * self.starts[d] * rf for d in range(w)])
        dst = V(*[pb.inbox(d, slot, r) for d in range(w)])
        n16 = (ctypes.c_uint32 * w)(*[self.counts[d] * rf // 4 for d in range(w)])
        flag = V(*[pb.flag(d, REC, slot, r) for d in range(w)])
        gate = V(*[pb.flag(r, ACK_REC, slot, d) for d in range(w)])       # owner d has consumed my records of this slot's last use
        nat.check(self.lib.vlsa_xchg_put(w, src, dst, n16, flag, None, gate, epoch, (epoch - 2) & 0xFFFFFFFF, self.timeout_ticks,
                                         VF._p(self.status_word), VF._stream()), "vlsa_xchg_put(records)")

    def _merge_head_owned(self, slot, src_ptr, T, ls, W, b, pool_w):
        """fold the `world` records of each of this rank's bags and run the head for them: results -> res[slot] sections"""
        pl_, lib, s, c, p = self.local, self.lib, VF._stream(), nat.check, VF._p
        c(lib.vlsa_normalize_rows(p(T), self.K, self.D, p(pl_.That), p(pl_.tnorm), s), "normalize_rows")
        if self.n_me == 0:
            return
        rb = self.res[slot].data_ptr()
        o_log, o_inc, o_vh, o_m2, o_l = (ctypes.c_void_p(rb + 4 * o) for o in self._res_off)
        c(lib.vlsa_vlfan_merge_head_batch_strided(ctypes.c_void_p(src_ptr), ctypes.c_void_p(src_ptr + 4 * nat.P_STRIDE),
                                                  ctypes.c_void_p(src_ptr + 4 * REC_HDR), self.n_me, self.world, self.P, self.D,
                                                  self._st_global, pl_.pool, p(pool_w),
                                                  None if pl_.identity_head else p(W), None if pl_.identity_head else p(b),
                                                  p(pl_.That), self.K, p(ls), o_m2, o_l, p(pl_.out), p(pl_.pooled),
                                                  p(pl_.v), o_vh, p(pl_.vnorm), o_log, o_inc, s), "merge_head_batch(owned)")

    def _put_results(self, slot, epoch):
        pb, w, r = self.peers, self.world, self.rank
        V = ctypes.c_void_p * w
        src = V(*[self.res[slot].data_ptr()] * w)
        dst = V(*[pb.resbox(d, slot, r) for d in range(w)])
        n16 = (ctypes.c_uint32 * w)(*[self.F // 4] * w)
        flag = V(*[pb.flag(d, RES, slot, r) for d in range(w)])
        ack = V(*[pb.flag(d, ACK_REC, slot, r) for d in range(w)])         # "I have folded your records of this slot"
        gate = V(*[pb.flag(r, ACK_RES, slot, d) for d in range(w)])        # peer d has collected my results of this slot's last use
        nat.check(self.lib.vlsa_xchg_put(w, src, dst, n16, flag, ack, gate, epoch, (epoch - 2) & 0xFFFFFFFF, self.timeout_ticks,
                                         VF._p(self.status_word), VF._stream()), "vlsa_xchg_put(results)")

    def _collect(self, slot, epoch, boxes, flags, acks):
        w, p = self.world, VF._p
        V = ctypes.c_void_p * w
        nat.check(self.lib.vlsa_xchg_collect(w, V(*boxes), None if flags is None else V(*flags), None if acks is None else V(*acks),
                                             self._counts_c, self._starts_c, self.nmax, self.K, self.D, epoch, self.timeout_ticks,
                                             p(self.logits), p(self.incidence), p(self.vhat), p(self.m2), p(self.l),
                                             p(self._ml_local[0]), p(self._ml_local[1]), p(self.status_word), VF._stream()),
                  "vlsa_xchg_collect")

    def _finish_attn(self, ab, m2, l):
        if ab is not None:   # this rank's columns, normalised with the GLOBAL (m2, l) of the merged records
            nat.check(self.lib.vlsa_attn_normalise_batch(VF._p(ab.ndesc), self.B, self.P, ab.max_n, VF._p(ab.desc), VF._p(m2), VF._p(l),
                                                         VF._p(ab.desc), VF._stream()), "attn_normalise_batch")
            views = ab.views
            self.A = [views[t] for t in self._inv]

    def _tail(self, slot, epoch, T, ls, W, b, pool_w, ab=None):
        """everything behind the record exchange of launch `epoch`, on the current stream"""
        pl_, lib, s, c, p = self.local, self.lib, VF._stream(), nat.check, VF._p
        cur = torch.cuda.current_stream()
        if self.exchange == "allgather":
            g = self.gathered[slot].data_ptr()
            c(lib.vlsa_normalize_rows(p(T), self.K, self.D, p(pl_.That), p(pl_.tnorm), s), "normalize_rows")
            c(lib.vlsa_vlfan_merge_head_batch_strided(ctypes.c_void_p(g), ctypes.c_void_p(g + 4 * nat.P_STRIDE),
                                                      ctypes.c_void_p(g + 4 * REC_HDR), self.B, self.world, self.P, self.D,
                                                      self._st_global, pl_.pool, p(pool_w),
                                                      None if pl_.identity_head else p(W), None if pl_.identity_head else p(b),
                                                      p(pl_.That), self.K, p(ls), p(pl_.m2), p(pl_.l), p(pl_.out), p(pl_.pooled),
                                                      p(pl_.v), p(pl_.vhat), p(pl_.vnorm), p(pl_.logits), p(pl_.incidence), s),
              "merge_head_batch(global)")
            self._finish_attn(ab, pl_.m2, pl_.l)
            return
        w, r = self.world, self.rank
        if self.exchange == "owner":
            self._merge_head_owned(slot, self.recv[slot].data_ptr(), T, ls, W, b, pool_w)
            if not self.skip_exchange:
                if self.comm_stream is not None:
                    self.done_tail[slot].record(cur)
                    with torch.cuda.stream(self.comm_stream):
                        self.comm_stream.wait_event(self.done_tail[slot])
                        self.dist.all_gather_into_tensor(self.resg[slot].view(-1), self.res[slot], group=self.group)
                        self.done_res[slot].record(self.comm_stream)
                    cur.wait_event(self.done_res[slot])
                else:
                    self.dist.all_gather_into_tensor(self.resg[slot].view(-1), self.res[slot], group=self.group)
            g = self.resg[slot].data_ptr()
            self._collect(slot, epoch, [g + 4 * o * self.F for o in range(w)], None, None)
        else:
            pb = self.peers
            V = ctypes.c_void_p * w
            nat.check(lib.vlsa_xchg_wait(w, V(*[pb.flag(r, REC, slot, d) for d in range(w)]), epoch, self.timeout_ticks,
                                         p(self.status_word), s), "vlsa_xchg_wait(records)")
            self._merge_head_owned(slot, pb.inbox(r, slot, 0), T, ls, W, b, pool_w)
            self._put_results(slot, epoch)
            self._collect(slot, epoch, [pb.resbox(r, slot, o) for o in range(w)], [pb.flag(r, RES, slot, o) for o in range(w)],
                          [pb.flag(o, ACK_RES, slot, r) for o in range(w)])
        self._finish_attn(ab, *self._ml_local)

    def _exchange_records(self, slot, epoch):
        """launch `epoch`'s records on their way (pipelined collectives: side stream; "ipc": a put kernel on this stream)"""
        if self.exchange == "ipc":              # (`skip_exchange` does not apply: the flags of a skipped launch would be missed later)
            self._put_records(slot, epoch)
            return
        if self.skip_exchange:
            return
        if self.exchange == "allgather":
            all_gather_records(self.rec[slot], self.gathered[slot], self.group)
        else:
            self.dist.all_to_all_single(self.recv[slot].view(-1)[:self.world * self.n_me * self.rf], self.rec[slot].view(-1),
                                        output_split_sizes=self._out_splits, input_split_sizes=self._in_splits, group=self.group)

    # -- driver -------------------------------------------------------------------------------------------------------------
    def run(self, Q, T, logit_scale, W=None, b=None, pool_w=None):
        slot = self._i & 1
        self._i += 1
        epoch = self._i & 0xFFFFFFFF
        cur = torch.cuda.current_stream()
        ab = self.attn[slot] if self.want_attn else None
        self._local(Q, slot, ab)
        if not self.pipeline:
            self._exchange_records(slot, epoch)
            self._tail(slot, epoch, T, logit_scale, W, b, pool_w, ab)
            return self.logits
        if self.comm_stream is not None:
            self.done_local[slot].record(cur)
            with torch.cuda.stream(self.comm_stream):
                self.comm_stream.wait_event(self.done_local[slot])
                self._exchange_records(slot, epoch)
                self.done_comm[slot].record(self.comm_stream)
        else:
            self._exchange_records(slot, epoch)
        if self._pending is not None:
            self._drain()
        self._pending = (slot, epoch, T, logit_scale, W, b, pool_w, ab)
        return self.logits

    def _drain(self):
        slot, epoch, T, ls, W, b, pw, ab = self._pending
        if self.comm_stream is not None:
            torch.cuda.current_stream().wait_event(self.done_comm[slot])
        self._tail(slot, epoch, T, ls, W, b, pw, ab)
        self._pending = None

    def finish(self, check: bool = False):
        """logits of the last launch.  ``check=True`` (COLLECTIVE, syncs): raise unless every rank's flag waits were served -- what a
        caller outside a timed region should use; the throughput loops check once after their last step (bench.py)."""
        if self._pending is not None:
            self._drain()
        if check:
            st = self.status_all()
            if st != 0:
                raise VlsaNativeError(f"sharded exchange: a flag wait timed out on some rank (status bits up to {st}); the results of this "
                                      "launch are void on every rank")
        return self.logits

    def close(self):
        """release the peer mappings and this rank's exchange buffer ("ipc"); COLLECTIVE when there is something to release: every
        rank's kernels must have finished writing into the others' buffers before any of them goes away"""
        if getattr(self, "peers", None) is not None:
            torch.cuda.synchronize()
            try:
                self.dist.barrier(group=self.group)
            except Exception:  # noqa: BLE001  (a torn-down group: nothing left to wait for)
                pass
            self.peers.close()
            self.peers = None


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
