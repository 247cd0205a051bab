"""Bag ingest: host feature files -> bags resident in HBM.

The reference reads one fp32 ``[n, 512]`` feature file per slide, concatenates a patient's slides on the host, and copies
the bag to the device with a blocking transfer at every step of every epoch (dataset/PatchWSI.py:197-215,
utils/io.py:16-42, runner/vlsa_handler.py:205,324): 102 MB over PCIe per 50k bag, ~2 ms, against ~10 us of kernel time.
An MI355X has 288 GB of HBM3E -- the whole TCGA-BLCA CONCH release (31.9 GB fp32, 16 GB as bf16) fits many times over --
so here every bag is uploaded ONCE into one arena and epochs then run at kernel speed:

* ``ArenaLayout``: host bookkeeping (row offsets, 64-row aligned so every bag starts on a tile boundary).
* ``DeviceBagArena``: one ``[capacity_rows, 512]`` bf16 tensor; ``add(key, slides)`` streams a patient's slides through two
  pinned staging buffers on a copy stream (the next chunk's host memcpy overlaps the previous chunk's DMA) and packs them
  back to back into the arena (``vlsa_pack_rows_bf16``: fp32 -> bf16 RNE on the device, or converted on the host to halve
  the PCIe bytes); the multi-slide concat of PatchWSI.py:214 becomes consecutive row ranges -- no extra copy.
* ``bag(key)`` -> ``[N, 512]`` view for ``VLSA.forward`` / ``forward_bags``; ``batches(keys, B)`` -> lists of views.
"""
from __future__ import annotations

import ctypes
from typing import Dict, Iterable, List, Sequence, Tuple, Union

import torch
import torch.utils.data

from . import _native as nat
from ._native import VlsaNativeError

ROW_ALIGN = 64  # rows; one workgroup iteration of the streaming kernels


class ArenaLayout:
    """Row bookkeeping of the arena (pure host logic)."""

    def __init__(self, capacity_rows: int, align: int = ROW_ALIGN):
        if capacity_rows <= 0 or align <= 0:
            raise ValueError("capacity_rows and align must be positive")
        self.capacity, self.align = int(capacity_rows), int(align)
        self.cursor = 0
        self.ranges: Dict[object, Tuple[int, int]] = {}

    def reserve(self, key, n_rows: int) -> int:
        """Reserve ``n_rows`` consecutive rows for ``key``; returns the first row."""
        if key in self.ranges:
            raise KeyError(f"bag {key!r} is already in the arena")
        if n_rows < 0:
            raise ValueError("n_rows must be >= 0")
        start = self.cursor
        if start + n_rows > self.capacity:
            raise MemoryError(f"arena full: {self.capacity - start} rows left, bag {key!r} needs {n_rows}")
        self.ranges[key] = (start, n_rows)
        self.cursor = min(self.capacity, -(-(start + n_rows) // self.align) * self.align)
        return start

    def rows_free(self) -> int:
        return self.capacity - self.cursor

    def reset(self):
        self.cursor = 0
        self.ranges.clear()

    def __contains__(self, key) -> bool:
        return key in self.ranges

    def __len__(self) -> int:
        return len(self.ranges)

    @staticmethod
    def rows_needed(sizes: Iterable[int], align: int = ROW_ALIGN) -> int:
        return sum(-(-int(n) // align) * align for n in sizes)


def read_patch_data(path: str) -> torch.Tensor:
    """One slide's features as a host tensor: ``.pt`` (torch.load on CPU) or ``.npy`` (utils/io.py:16-42)."""
    if path.endswith(".pt"):
        t = torch.load(path, map_location="cpu", weights_only=True)   # feature files hold one tensor: never unpickle code
    elif path.endswith(".npy"):
        import numpy as np
        t = torch.from_numpy(np.load(path))
    else:
        raise ValueError(f"Not support {path.rsplit('.', 1)[-1]}")
    return t


class DeviceBagArena:
    def __init__(self, capacity_rows: int, device, D: int = 512, chunk_rows: int = 16384, convert: str = "host",
                 host_threads: int = 8, dtype: torch.dtype = torch.bfloat16):
        """convert='host': slides are cast to bf16 while being copied into the pinned staging buffer, so only 1 KB/patch
        crosses PCIe (measured 1.3 ms per 50k-patch bag); 'device': fp32 crosses PCIe and ``vlsa_pack_rows_bf16`` casts
        in HBM (2.3 ms; for hosts short on cores).  ``host_threads`` caps torch's intra-op threads during the staging
        copies: with one thread per core of a 256-core host the 16 MB copies were 10x slower and erratic.
        dtype=torch.float32 keeps the reference's fp32 features bit for bit (2 KB per patch: the exact-fp32 kernels then
        run at ~4.6 TB/s instead of the bf16 kernels' 6.4, and results match the reference to 1e-5 instead of the
        ~1e-2 logit shift of rounding the features to bf16 once)."""
        if dtype not in (torch.bfloat16, torch.float32):
            raise ValueError("arena dtype must be torch.bfloat16 or torch.float32")
        self.dtype = dtype
        if dtype == torch.float32:
            convert = "host"          # nothing to convert: slides are copied as they are
        if convert not in ("device", "host"):
            raise ValueError("convert must be 'device' (fp32 over PCIe, bf16 cast in HBM) or 'host' (bf16 over PCIe)")
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise VlsaNativeError("DeviceBagArena needs an MI355X device (there is no CPU fallback)")
        self.lib = nat.load()
        self.D, self.chunk, self.convert = D, int(chunk_rows), convert
        self.host_threads = int(host_threads)
        self.layout = ArenaLayout(capacity_rows)
        self.data = torch.empty(capacity_rows, D, dtype=dtype, device=self.device)
        sdt = torch.float32 if (convert == "device" or dtype == torch.float32) else torch.bfloat16
        self._pinned = [torch.empty(self.chunk, D, dtype=sdt).pin_memory() for _ in range(2)]
        self._dev_stage = ([torch.empty(self.chunk, D, dtype=torch.float32, device=self.device) for _ in range(2)]
                           if convert == "device" else None)
        self._stream = torch.cuda.Stream(device=self.device)
        self._slot_free = [torch.cuda.Event(), torch.cuda.Event()]   # recorded when a slot's DMA (and pack) completed
        self._slot_used = [False, False]
        self._slot = 0
        self._ready: Dict[object, torch.cuda.Event] = {}
        self.bytes_h2d = 0

    # -- upload ------------------------------------------------------------------------------------------
    def _push(self, rows: torch.Tensor, dst_row: int):
        """rows: host [n <= chunk, D] (any float dtype) -> arena rows [dst_row, dst_row + n)."""
        n = rows.shape[0]
        s = self._slot
        if self._slot_used[s]:
            self._slot_free[s].synchronize()          # the staging buffer is still being read by an earlier DMA
        stage = self._pinned[s][:n]
        stage.copy_(rows)                             # host memcpy (+ cast to bf16 when convert == 'host')
        with torch.cuda.stream(self._stream):
            if self.convert == "host":
                self.data[dst_row:dst_row + n].copy_(stage, non_blocking=True)
            else:
                dev = self._dev_stage[s][:n]
                dev.copy_(stage, non_blocking=True)
                nat.check(self.lib.vlsa_pack_rows_bf16(ctypes.c_void_p(dev.data_ptr()), nat.DT_F32, n, self.D, self.D,
                                                       ctypes.c_void_p(self.data[dst_row:].data_ptr()), self.D,
                                                       ctypes.c_void_p(self._stream.cuda_stream)), "vlsa_pack_rows_bf16")
            self._slot_free[s].record(self._stream)
        self._slot_used[s] = True
        self.bytes_h2d += n * self.D * stage.element_size()
        self._slot ^= 1

    def add(self, key, slides: Union[torch.Tensor, Sequence[torch.Tensor]]) -> torch.Tensor:
        """Upload one bag = one patient's slides (a host tensor ``[n, D]`` or a list of them, concatenated in order as
        dataset/PatchWSI.py:214 does).  Returns the device view; the upload is asynchronous (``bag`` / ``wait`` order it)."""
        if isinstance(slides, torch.Tensor):
            slides = [slides]
        slides = [t.reshape(-1, t.shape[-1]) for t in slides]
        for t in slides:
            if t.is_cuda or t.shape[1] != self.D or not t.is_floating_point():
                raise ValueError(f"slides must be host float tensors [n, {self.D}]")
        total = sum(t.shape[0] for t in slides)
        row = self.layout.reserve(key, total)
        v_before = self.data._version               # only the arena's OWN bumps move `clean_version` (below)
        prev = torch.get_num_threads()
        if prev > self.host_threads > 0:
            torch.set_num_threads(self.host_threads)
        try:
            for t in slides:
                for a in range(0, t.shape[0], self.chunk):
                    part = t[a:a + self.chunk]
                    self._push(part, row)
                    row += part.shape[0]
        finally:
            if torch.get_num_threads() != prev:
                torch.set_num_threads(prev)
        ev = torch.cuda.Event()
        ev.record(self._stream)
        self._ready[key] = ev
        # the arena's in-place version after ITS OWN writes (see `modified_in_place`): advanced by this upload's bumps only -- a
        # user's in-place write to a resident view BEFORE a later (lazy, first-epoch) upload into the same segment must not be
        # absorbed into the snapshot
        self.clean_version = getattr(self, "clean_version", v_before) + (self.data._version - v_before)
        return self._view(key)

    def modified_in_place(self) -> bool:
        """True when a torch in-place op wrote to the arena (through ANY view of it: they share one version counter) since the arena's
        own last upload -- resident bags are handed out as views, not copies, so `X.mul_(2)` on one of them would silently corrupt the
        bag for every later epoch.  (Writes by raw-pointer kernels do not count; nothing in this package issues one on resident rows.)"""
        return self.data._version != getattr(self, "clean_version", self.data._version)

    def add_files(self, key, paths: Sequence[str]) -> torch.Tensor:
        return self.add(key, [read_patch_data(p) for p in paths])

    # -- access ------------------------------------------------------------------------------------------
    def _view(self, key) -> torch.Tensor:
        start, n = self.layout.ranges[key]
        return self.data[start:start + n]

    def bag(self, key) -> torch.Tensor:
        """[N, D] bf16 view of a bag; the current stream is ordered after the bag's upload."""
        ev = self._ready.get(key)
        if ev is not None:
            if ev.query():        # the upload has completed (seen by the host): every later launch on any stream is behind it
                del self._ready[key]
            else:
                torch.cuda.current_stream(self.device).wait_event(ev)
        return self._view(key)

    def uploaded(self, key) -> bool:
        """True once the bag's upload has completed (``bag`` then needs no event wait any more)"""
        ev = self._ready.get(key)
        if ev is not None and ev.query():
            del self._ready[key]
            ev = None
        return ev is None

    def bag_set(self, keys: Sequence):
        """the bags of ``keys`` as a checked ``vlsa_amd.functional.BagSet`` (validated once, descriptor rows kept) for ``forward_bags``"""
        from .functional import BagSet
        return BagSet([self.bag(k) for k in keys])

    def batches(self, keys: Sequence, batch_size: int = 32) -> Iterable[List[torch.Tensor]]:
        for i in range(0, len(keys), batch_size):
            yield [self.bag(k) for k in keys[i:i + batch_size]]

    def wait(self):
        self._stream.synchronize()

    def reset(self):
        """Forget all bags (e.g. before loading the next fold); the arena memory and staging buffers are kept."""
        self.wait()
        if self.data.is_cuda:
            # kernels on ANY stream may still be reading bag views handed out earlier; the rows are about to be overwritten
            torch.cuda.synchronize(self.device)
        self.layout.reset()
        self._ready.clear()

    def __contains__(self, key) -> bool:
        return key in self.layout

    def __len__(self) -> int:
        return len(self.layout)


class ResidentBagView(torch.Tensor):
    """A resident bag's rows as the handler's loader sees them: an ordinary device tensor that additionally KNOWS which item of
    which ``ResidentBags`` it is (``_vlsa_src = (resident_bags, index)``).

    Why a subclass: the reference's loaders collate with ``default_collate`` (runner/base_handler.py:233), i.e. ``torch.stack([feats])``
    -- for a bag that already sits in HBM that is a 51 MB device copy per 50k-patch bag, after which nothing links the tensor to
    the dataset item any more.  Here ``torch.stack`` of ONE such tensor along dim 0 returns the view ``feats[None]`` (still tagged),
    and ``.cuda()`` / ``.to(its own device)`` return it unchanged (runner/vlsa_handler.py:205,324), so ``net(X)`` receives the resident
    rows themselves plus their identity -- which is what lets an evaluation loop that calls the model once per bag be served from
    ONE batched launch over the next bags of the dataset (``VLSA`` look-ahead).  Every other operation yields a
    plain, untagged ``torch.Tensor``: the tag can never end up on data that is not bit for bit the resident bag."""

    _vlsa_src = None

    @staticmethod
    def wrap(t: torch.Tensor, src) -> "ResidentBagView":
        v = t.as_subclass(ResidentBagView)
        v._vlsa_src = src
        return v

    @classmethod
    def __torch_function__(cls, func, types, args=(), kwargs=None):
        kwargs = kwargs or {}
        if func in _view_meta_funcs():          # shape / dtype / device / dim / ...: no tensor comes back, nothing to untag
            with torch._C.DisableTorchFunctionSubclass():
                return func(*args, **kwargs)
        with torch._C.DisableTorchFunctionSubclass():
            if func is torch.stack and not kwargs.get("out"):
                seq = args[0] if args else kwargs.get("tensors")
                dim = args[1] if len(args) > 1 else kwargs.get("dim", 0)
                if isinstance(seq, (list, tuple)) and len(seq) == 1 and dim == 0 and isinstance(seq[0], ResidentBagView) and seq[0].dim() == 2:
                    return ResidentBagView.wrap(seq[0].as_subclass(torch.Tensor)[None], seq[0]._vlsa_src)
            elif func in (torch.Tensor.cuda, torch.Tensor.to) and isinstance(args[0], ResidentBagView):
                out = func(*args, **kwargs)
                me = args[0]
                if out.data_ptr() == me.data_ptr() and out.dtype == me.dtype and out.shape == me.shape and out.stride() == me.stride():
                    return me                        # a no-op move: the very same rows
                return out.as_subclass(torch.Tensor)
            out = func(*args, **kwargs)
        return _untag(out)


def _view_meta_funcs():
    from .deferred import _meta_funcs
    return _meta_funcs()


def _untag(out):
    if isinstance(out, ResidentBagView):
        return out.as_subclass(torch.Tensor)
    if isinstance(out, (list, tuple)):
        return type(out)(_untag(o) for o in out)
    return out


class ResidentBags(torch.utils.data.Dataset):
    """Drop-in wrapper for the reference's bag datasets (``WSIPatchSurv`` in 'patch' mode, ``FewShot_WSIPatchSurv``:
    dataset/PatchWSI.py:143-215): items are ``(index, (feats [N, 512], extra...), label)``.  The first time an item is asked for it is
    read through the wrapped dataset -- ``.pt`` files, multi-slide concat, ``.to(float)`` exactly as there -- and its features go
    into HBM once (``DeviceBagArena``: bf16 by default, pinned double-buffered async copies); afterwards the item comes back with
    ``feats`` as a view of the resident rows, so the handler's ``data_x[0].cuda()`` (runner/vlsa_handler.py:205,324) costs nothing
    and epochs 2.. never touch the disk or PCIe again.  No handler change: wrap what ``prepare_surv_dataset`` returns
    (``vlsa_amd.model_utils.patch_reference(resident_bags=True)`` does) and run the loaders with ``num_workers: 0`` -- inside a
    DataLoader worker process (no device there) items pass through unchanged.

    ``dtype=torch.float32`` keeps the features bit for bit (2 KB per patch).  Segments are allocated as the data arrives and grow
    geometrically -- the first holds ``first_segment_rows`` rows (128 MiB as bf16), each further one twice the last up to
    ``segment_rows`` -- so that a small validation split does not pin gigabytes (a bag larger than a segment gets its own)."""

    def __init__(self, dataset, device="cuda", dtype: torch.dtype = torch.bfloat16, segment_rows: int = 1 << 21, D: int = 512,
                 first_segment_rows: int = 1 << 17, tag_views: bool = True):
        """tag_views: resident features come back as ``ResidentBagView`` (see there): no collate copy, and the model can serve a
        bag-by-bag evaluation loop from batched launches.  NOTE the aliasing that comes with it: what the loader hands the handler IS
        the resident rows (the reference's host loader + ``.cuda()`` hand out a private copy per step) -- an in-place op on ``X``
        would change the bag for good.  Nothing in the reference's handlers writes to ``X``; a write through torch is detected at
        the next access (``DeviceBagArena.modified_in_place`` -> RuntimeError).  ``tag_views=False``: plain views; the loader's
        ``default_collate`` then copies every item (the reference's isolation, at one device copy per bag)."""
        self.dataset = dataset
        self._tag_views = bool(tag_views)
        self._device, self._dtype, self._segment_rows, self._D = torch.device(device), dtype, int(segment_rows), int(D)
        self._next_rows = min(int(first_segment_rows), int(segment_rows))
        self._warned_worker = False
        self._segments: List[DeviceBagArena] = []
        self._where: Dict[int, DeviceBagArena] = {}
        self._rest: Dict[int, tuple] = {}
        self._views: Dict[int, torch.Tensor] = {}
        self.reads = 0                    # items fetched from the wrapped dataset so far

    def __len__(self):
        return len(self.dataset)

    def __getattr__(self, name):          # uid, get_meta_data, summary, ...: whatever the handler asks the dataset for
        if name in ("dataset", "_segments", "_where", "_rest", "_views", "_next_rows", "_warned_worker", "_tag_views", "_la_sets"):
            raise AttributeError(name)
        return getattr(self.dataset, name)

    def _segment_for(self, n_rows: int) -> DeviceBagArena:
        need = -(-n_rows // ROW_ALIGN) * ROW_ALIGN
        if self._segments and self._segments[-1].layout.rows_free() >= need:
            return self._segments[-1]
        seg = DeviceBagArena(max(self._next_rows, need), self._device, D=self._D, dtype=self._dtype)
        self._next_rows = min(2 * max(self._next_rows, need), self._segment_rows)
        self._segments.append(seg)
        return seg

    def bag_set(self, indices=None):
        """The resident bags (all, or those of ``indices``) as a checked ``vlsa_amd.functional.BagSet`` for ``net.forward_bags``: an
        evaluation / training loop over a split then pays the per-bag validation and descriptor building once, not per call."""
        from .functional import BagSet
        idx = range(len(self)) if indices is None else indices
        views = []
        for i in idx:
            v = self.resident_view(i)
            if v is None:
                self[i]                                  # reads + uploads the item
                v = self.resident_view(i)
            if v is None:
                raise VlsaNativeError(f"item {i} is not a bag of patch features: it cannot be made resident")
            views.append(v)
        return BagSet(views)

    def resident_bytes(self) -> int:
        return sum(s.data.numel() * s.data.element_size() for s in self._segments)

    def __getitem__(self, i):
        i = int(i)
        if torch.utils.data.get_worker_info() is not None:
            if not self._warned_worker:   # a loader worker process: no device here
                self._warned_worker = True
                import warnings
                warnings.warn("vlsa_amd.ResidentBags is being read from a DataLoader worker process: bags pass through unchanged "
                              "(disk + PCIe every epoch).  Set `num_workers: 0` to keep them resident in HBM.", RuntimeWarning)
            return self.dataset[i]
        if i not in self._rest:
            item = self.dataset[i]
            self.reads += 1
            try:
                idx, data_x, label = item
                feats = data_x[0]
                ok = (isinstance(feats, torch.Tensor) and not feats.is_cuda and feats.dim() == 2 and feats.shape[1] == self._D
                      and feats.is_floating_point() and feats.shape[0] > 0)
            except (TypeError, ValueError, IndexError):
                ok = False
            if not ok:
                return item               # not a bag of patch features (cluster / graph modes, empty bags): untouched
            seg = self._segment_for(feats.shape[0])
            seg.add(i, feats)
            self._where[i], self._rest[i] = seg, (idx, tuple(data_x[1:]), label)
        idx, rest, label = self._rest[i]
        seg = self._where[i]
        if seg.modified_in_place():
            raise RuntimeError("vlsa_amd.ResidentBags: a resident bag was modified IN PLACE since it was uploaded (the handler's loader + "
                               "`.cuda()` hand out VIEWS of the rows in HBM, not private copies as the reference's host loader does): "
                               "every later epoch would read the modified rows.  Clone before an in-place op (`X = X.clone()`), or "
                               "build the dataset with ResidentBags(..., tag_views=False) to get a private copy per item.")
        feats = seg.bag(i)
        if self._tag_views:
            feats = ResidentBagView.wrap(feats, (self, i))
        return idx, (feats, *rest), label

    # -- what the model's look-ahead asks (vlsa_amd/vlsa.py) ----------------------------------------------------------------
    def resident_view(self, i: int):
        """plain [N, 512] view of item i if it is resident, else None (never reads the wrapped dataset)"""
        i = int(i)
        v = self._views.get(i)
        if v is None:
            seg = self._where.get(i)
            if seg is None:
                return None
            v = seg.bag(i)
            if seg.uploaded(i):
                self._views[i] = v       # the same view object from now on (no event, no slicing per call)
        return v
