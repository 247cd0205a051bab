"""One optimizer step of the reference's training loop as ONE object (runner/vlsa_handler.py:260-289: 32 x ``net(X)``, ``torch.cat``,
``calc_objective_loss``, one ``backward``, ``optimizer.step``; cfg_vlsa_conch.yaml:111-118).

Why: the step is ~190 dependent kernel launches of 5-15 us (text tower forward + backward 150 of them) under 1.6-1.9 ms of Python -- host
and GPU are level, so neither a faster kernel nor a leaner host shows up alone.  ``TrainStep`` takes the host out of the
repeated case: the first time a batch -- the same bag tensors, the same label tensors -- comes, the step runs eagerly; the second time
it is captured into a hipGraph (``torch.cuda.CUDAGraph``: forward_bags, the fused loss kernel, the whole autograd backward and a
capturable Adam), and from then on the batch costs ONE host call.  Everything the kernels read is referenced by address (resident bags,
parameters, optimizer state, the label tensors), so replays see the parameters the previous step left.

What a replay cannot do is run Python: the in-place version counters of the parameters do not move.  ``step`` bumps them itself after
every replay (``torch._C._increment_version``), so every cache keyed on parameter versions (text features, prepared queries, look-ahead
windows: vlsa_amd/vlsa.py) misses exactly as after an eager ``optimizer.step()``.

Batches that never repeat (a sampler that reshuffles every epoch) stay eager -- same numbers, no capture.  ``graph=False`` disables
capturing altogether.  Data parallel (``dist``): bags are the unit (SURVEY.md 8(e) "Training DP"); gradients are averaged with one
all-reduce of a flat buffer per step (eager steps only: a collective inside a captured step is left to the day a node exists).
"""
from __future__ import annotations

import os
from collections import OrderedDict
from typing import Optional, Sequence

import torch

from . import functional as VF


def _dbg(msg):
    if os.environ.get("VLSA_TRAINSTEP_DEBUG"):
        import sys
        sys.stderr.write(f"[TrainStep] {msg}\n")
        sys.stderr.flush()


def _bump_versions(tensors):
    try:
        torch._C._increment_version(tensors)            # torch >= 2.4: an iterable of tensors
    except (RuntimeError, TypeError):
        for t in tensors:
            torch._C._increment_version(t)


class TrainStep:
    def __init__(self, net, objective, optimizer, dist=None, group=None, world: int = 1, graph: bool = True, max_graphs: int = 8,
                 capture_after: int = 1):
        self.net, self.objective, self.opt = net, objective, optimizer
        self.dist, self.group, self.world = dist, group, int(world)
        self.params = [p for g in optimizer.param_groups for p in g["params"]]
        from .losses import SurvObjective
        self._obj_takes_log = isinstance(objective, SurvObjective)
        self.graph_enabled = bool(graph) and self.world == 1 and self._capturable()
        self.max_graphs, self.capture_after = int(max_graphs), int(capture_after)
        self._seen = OrderedDict()          # batch key -> eager steps seen
        self._graphs = OrderedDict()        # batch key -> (graph, loss, kept)
        self._pool = None
        self._cap_stream = None
        self.why_eager: Optional[str] = None if self.graph_enabled else ("graph=False" if not graph else
                                                                        "data parallel" if self.world > 1 else "optimizer is not capturable")
        self.n_eager = self.n_replay = self.n_capture = 0

    def _capturable(self) -> bool:
        return all(g.get("capturable", False) for g in self.opt.param_groups)

    # -- eager ------------------------------------------------------------------------------------------------------------------
    def _eager(self, bags, t, e):
        net = self.net
        logits = net.forward_bags(bags)[0]
        self.opt.zero_grad(set_to_none=True)
        if self._obj_takes_log and logits.dim() == 2 and logits.shape[0] <= 4096:
            # vlsa_amd.losses.SurvObjective: value AND gradient come out of its one launch (with the exp of the logit scale), so the
            # backward pass starts at the logits -- no ones_like seed, no product with the saved gradient (two launches of ~5 us)
            loss, g = self.objective.value_and_grad(logits, t, e, log_logit_scale=net.logit_scale)
            logits.backward(g)
        else:
            loss = self.objective(logits, t, e, net.get_logit_scale())
            loss.backward()
        if self.world > 1:
            self._allreduce_grads()
        self.opt.step()
        return loss.detach()          # (detached: a caller that keeps the loss must not keep the step's autograd graph -- and its
                                      #  AccumulateGrad nodes, bound to the stream they were created on -- alive into a capture)

    def _allreduce_grads(self):
        grads = [p.grad for p in self.params if p.grad is not None]
        if not grads:
            return
        flat = torch.cat([g.reshape(-1) for g in grads])
        if self.dist.get_backend(self.group) != "nccl":
            host = flat.cpu()
            self.dist.all_reduce(host, group=self.group)
            flat.copy_(host)
        else:
            self.dist.all_reduce(flat, group=self.group)
        flat.div_(self.world)
        o = 0
        for g in grads:
            n = g.numel()
            g.copy_(flat[o:o + n].view_as(g))
            o += n

    # -- graph ------------------------------------------------------------------------------------------------------------------
    @staticmethod
    def _key(bags, t, e):
        rows = bags.rows.tobytes() if isinstance(bags, VF.BagSet) else tuple((x.data_ptr(), x.shape[0], x.stride(0)) for x in bags)
        return (rows, t.data_ptr(), None if e is None else e.data_ptr(), t.shape)

    def _side(self, bags, t, e):
        """one EAGER step on the capture stream (the step before a capture): whatever the step creates lazily -- allocator pools of
        that stream, autograd's per-parameter AccumulateGrad nodes, library handles -- then exists on the stream the capture runs on"""
        if self._cap_stream is None:
            self._cap_stream = torch.cuda.Stream()
        cur = torch.cuda.current_stream()
        self._cap_stream.wait_stream(cur)
        with torch.cuda.stream(self._cap_stream):
            loss = self._eager(bags, t, e)
        cur.wait_stream(self._cap_stream)
        return loss

    def _capture(self, key, bags, t, e):
        bagset = bags if isinstance(bags, VF.BagSet) else VF.BagSet(bags)
        bagset.desc()                                   # the descriptor table goes up outside the capture
        g = torch.cuda.CUDAGraph()
        self.opt.zero_grad(set_to_none=True)
        torch.cuda.synchronize()
        _dbg("capture begins")
        try:
            with torch.cuda.graph(g, pool=self._pool, stream=self._cap_stream,
                                  capture_error_mode=os.environ.get("VLSA_GRAPH_CAPTURE_MODE", "global")):
                loss = self._eager(bagset, t, e)
        except Exception as exc:  # noqa: BLE001  (whatever refuses to be captured: this batch -- and the object -- stay eager)
            self.graph_enabled = False
            self.why_eager = f"capture failed: {type(exc).__name__}: {str(exc)[:300]}"
            _dbg(self.why_eager)
            try:
                torch.cuda.synchronize()
            except Exception:  # noqa: BLE001
                pass
            self.net._drop_text_cache()
            self.opt.zero_grad(set_to_none=True)
            return None
        _dbg("capture ended")
        if self._pool is None:
            self._pool = g.pool()
        self.n_capture += 1
        # a capture enqueues nothing: the step it recorded runs now
        self._graphs[key] = (g, loss, (bagset, t, e))
        while len(self._graphs) > self.max_graphs:
            self._graphs.popitem(last=False)
        return self._replay(key)

    def _replay(self, key):
        g, loss, _ = self._graphs[key]
        self._graphs.move_to_end(key)
        sync = getattr(self.opt, "sync_hyper", None)
        if sync is not None:
            sync()                                      # vlsa_amd.optim.FusedAdam: a changed learning rate reaches the device table
        g.replay()
        _bump_versions(self.params)                     # what optimizer.step() does to the version counters, without a kernel
        self.net._drop_text_cache()                     # (the capture left the text cache pointing at a graph-owned tensor)
        self.n_replay += 1
        return loss

    def step(self, bags: Sequence[torch.Tensor], t: torch.Tensor, e: Optional[torch.Tensor] = None):
        """bags: the step's resident bags (a list or a ``BagSet``); t / e: the labels on the device.  Returns the loss (a 0-dim device
        tensor; in graph mode the SAME tensor every time, overwritten by the next replay of that batch)."""
        if not self.graph_enabled:
            self.n_eager += 1
            return self._eager(bags, t, e)
        key = self._key(bags, t, e)
        if key in self._graphs:
            return self._replay(key)
        n = self._seen.get(key, 0)
        if n > self.capture_after:
            self._seen.pop(key, None)
            loss = self._capture(key, bags, t, e)
            if loss is not None:
                return loss
            self.n_eager += 1
            return self._eager(bags, t, e)
        self._seen[key] = n + 1
        if n == self.capture_after:         # the step before the capture: eager, on the capture stream
            while len(self._seen) > 4096:
                self._seen.popitem(last=False)
            self.n_eager += 1
            return self._side(bags, t, e)
        while len(self._seen) > 4096:
            self._seen.popitem(last=False)
        self.n_eager += 1
        return self._eager(bags, t, e)

    def describe(self) -> dict:
        return {"mode": "hipGraph replay of the whole step" if self.n_replay else "eager", "eager_steps": self.n_eager,
                "captures": self.n_capture, "replays": self.n_replay, "why_eager": self.why_eager,
                "optimizer": type(self.opt).__name__ + ("(fused, capturable)" if self._capturable() else "")}

    def close(self):
        self._graphs.clear()
        self._seen.clear()
        self._pool = None
