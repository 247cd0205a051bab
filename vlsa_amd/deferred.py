"""Training calls of the reference's handler at batched speed -- without touching the handler.

``runner/vlsa_handler.py:260-289`` (``_update_network``) calls the model once per bag::

    for i in range(n_sample):
        pred, *_ = self.net(xs[i]);  y_hat.append(pred)
    self.optimizer.zero_grad()
    bag_preds = torch.cat(y_hat, dim=0)                 # the FIRST time anything looks at a prediction
    pred_loss = self.calc_objective_loss(bag_preds, bag_label);  pred_loss.backward();  self.optimizer.step()

Bag by bag that is 32 autograd nodes and ~130 launches each way per step, host-bound at 4.3-4.8 ms, where ONE
``VLSA.forward_bags`` over the same 32 bags (one persistent forward launch, one backward launch, one head node) is 1.8 ms
(DESIGN.md 5c).  With ``net.defer_training_calls = True`` (``patch_reference()`` sets it) a grad-enabled ``net(X)`` in training mode
therefore does NOT run: it records the bag and hands back three ``DeferredOutput`` tensors -- ``torch.Tensor`` subclass instances of the
right shape / dtype / device that carry (batch, index).  The first torch operation that touches ANY output of the batch -- the
handler's ``torch.cat`` -- first runs ``forward_bags`` over every bag recorded so far (gradients enabled, the model's normal batched
training route) and then executes on the real rows (``logits[i:i+1]`` ...), so autograd sees exactly the graph of a ``forward_bags``
call.  Shape / dtype / device queries do not trigger anything.

What is guaranteed: an output is only ever computed under the parameters its call saw.  A batch remembers the in-place versions of
every trainable tensor at each call (a later call after an optimizer step opens a new batch) and the full state key of the model
(text-side modules, encoder, flags) at its first call; materialising under a different key raises instead of returning numbers the
bag-by-bag call would not have produced.  Models with an active dropout layer, opaque text providers, CPU bags or anything but a
``[1, N, D]`` / ``[N, D]`` device tensor are not deferred.  Off by default: code that hands a deferred output to an API that
bypasses ``__torch_function__`` (``torch.autograd.grad`` on it) would see a placeholder leaf.
"""
from __future__ import annotations

import torch

_META = None


def _meta_funcs():
    """attribute reads / methods that only ask for metadata: answered from the placeholder, nothing is computed"""
    global _META
    if _META is None:
        T = torch.Tensor
        # (`is_leaf` is NOT here: the placeholder is a leaf, the real row is a slice of the batched result -- asking materialises)
        names = ("shape", "dtype", "device", "requires_grad", "is_cuda", "ndim", "layout", "names", "is_sparse", "is_quantized",
                 "is_meta")
        fs = {getattr(T, n).__get__ for n in names if hasattr(T, n)}
        fs |= {T.dim, T.size, T.numel, T.__len__, T.nelement, T.ndimension, T.element_size, T.is_floating_point, T.is_complex,
               T.get_device, T.stride, T.is_contiguous, T.storage_offset}
        _META = fs
    return _META


class DeferredOutput(torch.Tensor):
    """One output of a deferred ``net(X)`` call (see the module docstring)."""

    _vlsa_batch = None
    _vlsa_slot = None          # (bag index, which): which = 0 logits [1, K], 1 image features [1, D], 2 text features [K, D]

    @staticmethod
    def make(batch, index, which, base):
        # a placeholder of the output's shape / dtype / device: aliases the batch's (never read) base tensor -- no allocation per call
        t = torch.Tensor._make_subclass(DeferredOutput, base, True)
        t._vlsa_batch, t._vlsa_slot = batch, (index, which)
        return t

    @classmethod
    def __torch_function__(cls, func, types, args=(), kwargs=None):
        kwargs = kwargs or {}
        if func in _meta_funcs():
            with torch._C.DisableTorchFunctionSubclass():
                return func(*args, **kwargs)
        if func in (torch.cat, torch.concat, torch.concatenate):
            # the handler's `torch.cat(y_hat, dim=0)`: consecutive rows of one batch are ONE slice of the batched result -- as 32
            # single-row slices the backward pass would be 32 slice nodes + 31 adds of zero-padded [B, K] gradients (~1 ms per step)
            seq = args[0] if args else kwargs.get("tensors")
            dim = args[1] if len(args) > 1 else kwargs.get("dim", 0)
            if isinstance(seq, (list, tuple)) and dim in (0, -2) and kwargs.get("out") is None:
                merged = _merge_rows(seq)
                if merged is not None:
                    with torch._C.DisableTorchFunctionSubclass():
                        if len(merged) == 1:
                            # `torch.cat` hands out FRESH storage: an in-place op on the handler's `bag_preds` must not reach what the
                            # other outputs of the batch will read later -- one [B, K] copy per step
                            return merged[0].clone()
                        return torch.cat(merged, dim=0)
        args = _real(args)
        kwargs = {k: _real(v) for k, v in kwargs.items()}
        with torch._C.DisableTorchFunctionSubclass():
            return func(*args, **kwargs)

    def __repr__(self):                      # (printing a prediction is a use: show the real row)
        return repr(_real(self))


def _merge_rows(seq):
    """the real tensors of a sequence to be concatenated along dim 0, runs of consecutive rows (logits or image features) of one
    batch as one slice each; None when the sequence holds anything this does not understand (the generic route then applies)"""
    out, i, n = [], 0, len(seq)
    while i < n:
        x = seq[i]
        if isinstance(x, DeferredOutput) and x._vlsa_slot[1] in (0, 1):
            b, (r0, which) = x._vlsa_batch, x._vlsa_slot
            j = i + 1
            while (j < n and isinstance(seq[j], DeferredOutput) and seq[j]._vlsa_batch is b
                   and seq[j]._vlsa_slot == (r0 + (j - i), which)):
                j += 1
            if b.real is None:
                b._materialise()
            full = b.real[which]
            out.append(full if (r0 == 0 and j - i == full.shape[0]) else full[r0:r0 + (j - i)])
            i = j
        elif isinstance(x, DeferredOutput):
            out.append(_real(x))
            i += 1
        elif isinstance(x, torch.Tensor):
            out.append(x)
            i += 1
        else:
            return None
    return out


def _real(x):
    if isinstance(x, DeferredOutput):
        return x._vlsa_batch.value(x._vlsa_slot)
    if isinstance(x, (list, tuple)):
        return type(x)(_real(y) for y in x)
    if isinstance(x, dict):
        return {k: _real(v) for k, v in x.items()}
    return x


class TrainingCalls:
    """The bags of the ``net(X)`` calls since the last materialisation, under ONE parameter state."""

    def __init__(self, model, key, trainable, K, D, device):
        self.model, self.key = model, key
        self.trainable = trainable                                  # tensors an optimizer may move
        self.versions = [t._version for t in trainable]
        self.bags, self.real, self.error = [], None, None
        self.device = device
        self._base = (torch.empty(1, K, device=device), torch.empty(1, D, device=device))
        self._text = DeferredOutput.make(self, 0, 2, torch.empty(K, D, device=device))      # one object for every call of the batch

    def same_state(self) -> bool:
        """cheap per-call check: no trainable tensor was modified in place since the batch was opened"""
        return self.real is None and self.error is None and [t._version for t in self.trainable] == self.versions

    @staticmethod
    def takes(X) -> bool:
        return (isinstance(X, torch.Tensor) and X.is_cuda and not X.requires_grad
                and ((X.dim() == 3 and X.shape[0] == 1) or X.dim() == 2) and X.shape[-2] > 0)

    def add(self, X):
        if X.device != self.device:
            raise RuntimeError("vlsa_amd: bags of one deferred training batch must live on one device")
        if type(X) is not torch.Tensor:
            X = X.as_subclass(torch.Tensor)
        i = len(self.bags)
        self.bags.append(X)
        return DeferredOutput.make(self, i, 0, self._base[0]), DeferredOutput.make(self, i, 1, self._base[1]), self._text

    def value(self, slot):
        if self.real is None:
            self._materialise()
        i, which = slot
        logits, feats, text = self.real
        return logits[i:i + 1] if which == 0 else (feats[i:i + 1] if which == 1 else text)

    def _materialise(self):
        m = self.model
        if self.error is not None:
            raise RuntimeError(self.error)
        if m._pending_calls is self:
            m._pending_calls = None
        if [t._version for t in self.trainable] != self.versions or m._defer_key() != self.key:
            self.error = ("vlsa_amd: a deferred net(X) output is being used after the model changed (a parameter was modified in place, "
                          "or a module switched train / eval) -- its value under the parameters of the call can no longer be computed. "
                          "Use the prediction before stepping the optimizer, or set net.defer_training_calls = False.")
            self.bags = None
            raise RuntimeError(self.error)
        prev = m._materialising
        m._materialising = True
        try:
            with torch.enable_grad():
                self.real = m.forward_bags(self.bags)
        finally:
            m._materialising = prev
        from . import vlsa as _v
        if _v.ENV_PARANOID:
            import random
            j = random.randrange(len(self.bags))
            m._paranoid_check(self.bags[j], self.real[0][j], f"deferred training batch of {len(self.bags)} bags, bag {j}")
        self.bags = None
