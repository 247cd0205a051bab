"""Text tower of the VL model on the GPU: drop-in counterpart of the reference's ``CONCHPromptEncoder``
(model/prompt_encoder.py:209-322) -- same constructor contract (wraps a CoCa model's ``.text`` tower), same attribute
and state-dict names (``positional_embedding``, ``cls_emb``, ``transformer.resblocks.N.{ln_1,attn,ln_2,mlp}.*``,
``ln_final.*``, ``text_projection``, ``token_embedding.weight``), same ``forward(prompts_text | prompts_embedding,
prompts_pseudo_tokens)`` -> ``[n_prompts, output_dim]``.

The 12 pre-LN blocks run in libvlsa_hip.so (vlsa_amd/csrc/text_tower.hip) on the COMPACT rows of the prompts: only the
positions that can reach the pooled CLS token are evaluated (a 13-row problem per rank prompt instead of 128 -- exact, see
the kernel file), f32 MFMA throughout.  Differentiable w.r.t. ``prompts_embedding`` (the learnable context / rank
embeddings of the prompt learner); the tower itself is frozen in every shipped configuration
(``vlsa_txt_encoder_frozen: True``, cfg_vlsa_conch.yaml:69).  A tower whose own parameters require grad
(``vlsa_txt_encoder_frozen: False``) runs the same kernels plus the weight-gradient products of ``vlsa_tt_backward_train``
(``_TextTowerTrainFn``; until round 4 that case was a torch route).
"""
from __future__ import annotations

import ctypes
import os
from collections import OrderedDict

import torch
import torch.nn as nn

from . import _native as nat
from ._native import VlsaNativeError


class _TTLayer(ctypes.Structure):
    _fields_ = [(n, ctypes.c_void_p) for n in ("ln1_w", "ln1_b", "in_w", "in_b", "out_w", "out_b", "ln2_w", "ln2_b", "fc_w", "fc_b",
                                               "proj_w", "proj_b")]


class _TTModel(ctypes.Structure):
    _fields_ = [("width", ctypes.c_int), ("heads", ctypes.c_int), ("layers", ctypes.c_int), ("out_dim", ctypes.c_int),
                ("ctx_len", ctypes.c_int), ("layer", ctypes.POINTER(_TTLayer)), ("pos_emb", ctypes.c_void_p),
                ("cls_emb", ctypes.c_void_p), ("lnf_w", ctypes.c_void_p), ("lnf_b", ctypes.c_void_p), ("text_proj", ctypes.c_void_p)]


class _TTRows(ctypes.Structure):
    _fields_ = [("n_seq", ctypes.c_int), ("M", ctypes.c_int), ("M_pad", ctypes.c_int), ("max_len", ctypes.c_int),
                ("row_seq", ctypes.c_void_p), ("row_pos", ctypes.c_void_p), ("row_src", ctypes.c_void_p),
                ("seq_row0", ctypes.c_void_p), ("cls_keep", ctypes.c_void_p), ("prefix_len", ctypes.c_int)]


# -- parameter holders with the reference's names (model/conch/transformer.py:191-247,290-322) ---------------------------
class _SelfAttention(nn.Module):          # the parameters of nn.MultiheadAttention(width, heads)
    def __init__(self, width):
        super().__init__()
        self.in_proj_weight = nn.Parameter(torch.empty(3 * width, width))
        self.in_proj_bias = nn.Parameter(torch.zeros(3 * width))
        self.out_proj = nn.Linear(width, width)


class _Block(nn.Module):
    def __init__(self, width):
        super().__init__()
        self.ln_1 = nn.LayerNorm(width)
        self.attn = _SelfAttention(width)
        self.ln_2 = nn.LayerNorm(width)
        self.mlp = nn.Sequential(OrderedDict([("c_fc", nn.Linear(width, 4 * width)), ("gelu", nn.GELU()),
                                              ("c_proj", nn.Linear(4 * width, width))]))


class _Transformer(nn.Module):
    def __init__(self, width, layers, heads):
        super().__init__()
        self.width, self.layers, self.heads = width, layers, heads
        self.resblocks = nn.ModuleList([_Block(width) for _ in range(layers)])


def compact_rows(pseudo_tokens: torch.Tensor, ctx_len: int, prefix_len: int = 0):
    """Host-side row plan for a [n, ctx_len - 1] pseudo-token matrix (0 = pad).  The CLS row (appended last) may attend to
    column 0 and to column j + 1 for every non-pad token j (build_cls_mask, model/prompt_encoder.py:245-252: the pad mask is
    shifted right by one); token rows are causal.  Prompt s therefore needs positions 0..m_s (m_s = last column < ctx_len - 1
    the CLS row can see) plus the CLS row.

    prefix_len = L > 0: the caller guarantees that positions 0 .. L-1 carry the same embedding in every prompt (<sot> and the
    shared context tokens of the CoOp learners).  Under the causal mask those rows are identical in every layer, so they are
    planned ONCE (rows 0 .. L-1) and prompt s keeps only positions L .. m_s + its CLS row; every CLS row must see all L prefix
    positions (true for real tokens) and every prompt must reach past the prefix, else the plan falls back to L = 0.
    Returns dict of int lists (+ 'prefix_len': the L actually used)."""
    pt = pseudo_tokens.detach().to("cpu")
    n, L_tok = pt.shape
    if L_tok != ctx_len - 1:
        raise ValueError(f"expected {ctx_len - 1} token positions, got {L_tok}")
    keeps, ms = [], []
    for s in range(n):
        keep = [True] + [bool(v) for v in (pt[s] != 0).tolist()]          # keep[c]: CLS attends to column c (c = L_tok: itself)
        keeps.append(keep)
        ms.append(max(c for c in range(L_tok) if keep[c]))
    L = int(prefix_len or 0)
    if L > 0 and not (n > 1 and all(m >= L for m in ms) and all(all(k[:L]) for k in keeps)):
        L = 0
    row_seq, row_pos, row_src, seq_row0, cls_keep = [], [], [], [], []
    for c in range(L):                                                    # the shared prefix, once
        row_seq.append(0); row_pos.append(c); row_src.append(c); cls_keep.append(1)
    seq_row0.append(len(row_seq))
    max_len = 0
    for s in range(n):
        keep, m = keeps[s], ms[s]
        for c in range(L, m + 1):
            row_seq.append(s); row_pos.append(c); row_src.append(c); cls_keep.append(1 if keep[c] else 0)
        row_seq.append(s); row_pos.append(ctx_len - 1); row_src.append(-1); cls_keep.append(1 if keep[L_tok] else 0)
        seq_row0.append(len(row_seq))
        max_len = max(max_len, m + 2)                                     # keys of the prompt's CLS row incl. the prefix
    return dict(row_seq=row_seq, row_pos=row_pos, row_src=row_src, seq_row0=seq_row0, cls_keep=cls_keep, max_len=max_len,
                n_seq=n, M=len(row_seq), prefix_len=L)


class _RowPlan:
    """Device copies of the compact-row tables + the C struct, cached per pseudo-token pattern."""

    def __init__(self, pseudo_tokens, ctx_len, device, prefix_len=0):
        rp = compact_rows(pseudo_tokens, ctx_len, prefix_len)
        self.n_seq, self.M, self.max_len, self.prefix_len = rp["n_seq"], rp["M"], rp["max_len"], rp["prefix_len"]
        self.M_pad = (self.M + 95) // 96 * 96      # whole 16-, 32- and 48-row workgroup tiles
        pad = self.M_pad - self.M
        i32 = lambda v, fill: torch.tensor(v + [fill] * pad, dtype=torch.int32).to(device)   # noqa: E731
        self.row_seq, self.row_pos, self.row_src = i32(rp["row_seq"], 0), i32(rp["row_pos"], 0), i32(rp["row_src"], -1)
        self.seq_row0 = torch.tensor(rp["seq_row0"], dtype=torch.int32).to(device)
        self.cls_keep = torch.tensor(rp["cls_keep"] + [0] * pad, dtype=torch.uint8).to(device)
        self.c = _TTRows(self.n_seq, self.M, self.M_pad, self.max_len, self.row_seq.data_ptr(), self.row_pos.data_ptr(),
                         self.row_src.data_ptr(), self.seq_row0.data_ptr(), self.cls_keep.data_ptr(), self.prefix_len)
        self.ws = None    # inference workspace (no activations kept), allocated on first use
        self._rp = rp
        self._status = []  # (pinned int32[4], event): status words of persistent launches on their way to the host
        self.ws_free = {}  # (bytes, stream) -> workspaces of finished forward + backward passes, ready for the next training forward

    def watch_status(self, words: torch.Tensor):
        host = torch.empty(4, dtype=torch.int32).pin_memory()
        host.copy_(words, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        self._status.append((host, ev))
        if len(self._status) > 64:
            self.check_status(wait=True)

    def check_status(self, wait: bool = False):
        """Raise if a persistent text-tower launch reported a timed-out in-kernel wait (vlsa_tt_status_offset): its text features
        -- and everything computed from them -- are void."""
        keep = []
        for host, ev in self._status:
            if wait:
                ev.synchronize()
            if not ev.query():
                keep.append((host, ev))
                continue
            code, stage, wg, seen = (int(v) for v in host)
            if code != 0:
                self._status = []
                raise VlsaNativeError(f"text tower: the persistent forward launch timed out in stage {stage} (workgroup {wg}, counter {seen}): "
                                      "not all of its workgroups were resident (another kernel held CUs).  Its outputs are void; "
                                      "VLSA_TT_PERSIST=0 selects the launch-per-stage path")
        self._status = keep


class _TextTowerFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, emb, enc, plan):
        lib = nat.load()
        model = enc._c_model(emb.device)
        save = 1 if ctx.needs_input_grad[0] else 0
        nbytes = lib.vlsa_tt_workspace_bytes(ctypes.byref(model), ctypes.byref(plan.c), save)
        if nbytes == 0:
            raise VlsaNativeError("text tower: unsupported shape (width % 128, width <= 768, 64 features per head, <= 128 rows per prompt)")
        if save:
            # holds the activations until backward; taken from the plan's pool of workspaces a finished backward handed back (same
            # stream: ordered behind that backward's kernels) -- a fresh one costs a 41 MB fill per step (15 us, round 4)
            key = (nbytes, torch.cuda.current_stream(emb.device).cuda_stream)
            free = plan.ws_free.get(key)
            ws = free.pop() if free else torch.zeros(nbytes, dtype=torch.uint8, device=emb.device)
            ctx.ws_key = key
        else:
            if plan.ws is None or plan.ws.numel() < nbytes:
                plan.ws = torch.zeros(nbytes, dtype=torch.uint8, device=emb.device)
            ws = plan.ws
        x = emb.detach()
        if x.dtype != torch.float32 or x.stride(-1) != 1:
            x = x.float().contiguous()
        out = torch.empty(plan.n_seq, enc.output_dim, dtype=torch.float32, device=emb.device)
        s = ctypes.c_void_p(torch.cuda.current_stream(emb.device).cuda_stream)
        packed = enc._packed_weights(emb.device, with_backward=bool(save))
        plan.check_status()                  # a time-out of an EARLIER persistent launch (see below) surfaces here at the latest
        # opt-in persistent forward: asked for per call through a flag bit -- the switch lives HERE (VLSA_TT_PERSIST=1, read per call:
        # tests and benches flip it inside one process); libvlsa_hip.so reads no environment
        flags = save | (nat.TT_PERSISTENT if os.environ.get("VLSA_TT_PERSIST", "0") not in ("", "0") else 0)
        nat.check(lib.vlsa_tt_forward(ctypes.byref(model), ctypes.byref(plan.c), ctypes.c_void_p(packed.data_ptr()),
                                      ctypes.c_void_p(x.data_ptr()), x.stride(0), x.stride(1), ctypes.c_void_p(ws.data_ptr()), flags,
                                      ctypes.c_void_p(out.data_ptr()), s), "vlsa_tt_forward")
        off = lib.vlsa_tt_status_offset(ctypes.byref(model), ctypes.byref(plan.c), flags)
        if off >= 0:
            # the persistent launch reports a timed-out wait in its workspace: read the four words back WITHOUT stalling the
            # caller (pinned buffer + event) and look at them once the copy has landed -- here at the next call, in backward,
            # or through ``check_status(wait=True)``
            plan.watch_status(ws[off:off + 16].view(torch.int32))
        if save:
            ctx.ws, ctx.plan, ctx.enc, ctx.shape, ctx.packed = ws, plan, enc, tuple(emb.shape), packed
        ctx.keep = x
        return out

    @staticmethod
    def backward(ctx, dout):
        lib = nat.load()
        enc, plan = ctx.enc, ctx.plan
        plan.check_status()
        model = enc._c_model(dout.device)
        g = dout.detach().float().contiguous()
        demb = torch.empty(ctx.shape, dtype=torch.float32, device=dout.device)
        s = ctypes.c_void_p(torch.cuda.current_stream(dout.device).cuda_stream)
        nat.check(lib.vlsa_tt_backward(ctypes.byref(model), ctypes.byref(plan.c), ctypes.c_void_p(ctx.packed.data_ptr()),
                                       ctypes.c_void_p(g.data_ptr()), ctypes.c_void_p(ctx.ws.data_ptr()),
                                       ctypes.c_void_p(demb.data_ptr()), demb.stride(0), demb.stride(1), demb.numel(), s),
                  "vlsa_tt_backward")
        if torch.cuda.current_stream(dout.device).cuda_stream == ctx.ws_key[1]:
            free = plan.ws_free.setdefault(ctx.ws_key, [])
            if len(free) < 2:
                free.append(ctx.ws)          # (the kernels leave their tickets / counters zeroed: reusable as it is)
        ctx.ws = ctx.packed = None
        return demb, None, None


class _TextTowerTrainFn(torch.autograd.Function):
    """The tower with its OWN parameters under training (``vlsa_txt_encoder_frozen: False``, runner/vlsa_handler.py:131; off in
    every shipped configuration).  Same kernels as ``_TextTowerFn``; the forward keeps every block's attention output
    (``save_for_backward == 2``) and the backward is ONE C call (``vlsa_tt_backward_train``) that writes d prompts_embedding and
    the gradient of every tower parameter -- four dW = dY^T act products per block over the compact rows on the f32 matrix pipe,
    bias / LayerNorm gradients as fixed-order column sums (vlsa_amd/csrc/text_tower.hip: k_tt_dw).  ``tensors`` = the tower's
    parameters in ``_tower_tensors`` order (autograd inputs: their ``.grad`` is what the optimizer reads)."""

    @staticmethod
    def forward(ctx, emb, enc, plan, *tensors):
        lib = nat.load()
        model = enc._c_model(emb.device)
        nbytes = lib.vlsa_tt_workspace_bytes(ctypes.byref(model), ctypes.byref(plan.c), 2)
        if nbytes == 0:
            raise VlsaNativeError("text tower: unsupported shape (width % 128, width <= 768, 64 features per head, <= 128 rows per prompt)")
        ws = torch.zeros(nbytes, dtype=torch.uint8, device=emb.device)
        x = emb.detach()
        if x.dtype != torch.float32 or x.stride(-1) != 1:
            x = x.float().contiguous()
        out = torch.empty(plan.n_seq, enc.output_dim, dtype=torch.float32, device=emb.device)
        s = ctypes.c_void_p(torch.cuda.current_stream(emb.device).cuda_stream)
        packed = enc._packed_weights(emb.device, with_backward=True)       # of the CURRENT weights (keyed on their versions)
        nat.check(lib.vlsa_tt_forward(ctypes.byref(model), ctypes.byref(plan.c), ctypes.c_void_p(packed.data_ptr()),
                                      ctypes.c_void_p(x.data_ptr()), x.stride(0), x.stride(1), ctypes.c_void_p(ws.data_ptr()), 2,
                                      ctypes.c_void_p(out.data_ptr()), s), "vlsa_tt_forward")
        ctx.ws, ctx.plan, ctx.enc, ctx.shape, ctx.packed = ws, plan, enc, tuple(emb.shape), packed
        ctx.versions = [t._version for t in tensors]
        ctx.keep = x
        return out

    @staticmethod
    def backward(ctx, dout):
        lib = nat.load()
        enc, plan = ctx.enc, ctx.plan
        ts = enc._tower_tensors()
        if [t._version for t in ts] != ctx.versions:
            raise RuntimeError("text tower: a weight was modified in place between forward and backward (the packed copies the "
                               "backward reads belong to the forward's weights)")
        model = enc._c_model(dout.device)
        g = dout.detach().float().contiguous()
        demb = torch.empty(ctx.shape, dtype=torch.float32, device=dout.device)
        # one allocation for all parameter gradients, handed out as views shaped like the parameters; the C struct mirrors _c_model
        sizes = [t.numel() for t in ts]
        flat = torch.empty(sum(sizes), dtype=torch.float32, device=dout.device)
        grads, o = [], 0
        for t, n in zip(ts, sizes):
            grads.append(flat[o:o + n].view(t.shape))
            o += n
        nl = (len(ts) - 5) // 12
        arr = (_TTLayer * nl)()
        for i in range(nl):
            for j, (name, _) in enumerate(_TTLayer._fields_):
                setattr(arr[i], name, grads[5 + 12 * i + j].data_ptr())
        gm = _TTModel(model.width, model.heads, model.layers, model.out_dim, model.ctx_len, arr, grads[0].data_ptr(), grads[1].data_ptr(),
                      grads[2].data_ptr(), grads[3].data_ptr(), grads[4].data_ptr())
        s = ctypes.c_void_p(torch.cuda.current_stream(dout.device).cuda_stream)
        nat.check(lib.vlsa_tt_backward_train(ctypes.byref(model), ctypes.byref(plan.c), ctypes.c_void_p(ctx.packed.data_ptr()),
                                             ctypes.c_void_p(g.data_ptr()), ctypes.c_void_p(ctx.ws.data_ptr()),
                                             ctypes.c_void_p(demb.data_ptr()), demb.stride(0), demb.stride(1), demb.numel(),
                                             ctypes.byref(gm), s), "vlsa_tt_backward_train")
        ctx.ws = ctx.packed = None
        need = ctx.needs_input_grad
        return (demb if need[0] else None, None, None) + tuple(gr if need[3 + i] else None for i, gr in enumerate(grads))


class CONCHPromptEncoder(nat.TransientCaches, nn.Module):
    """``CONCHPromptEncoder(coca_model)`` adopts the text tower of a CoCa model exactly like the reference does
    (``coca_model.text.{positional_embedding, transformer, ln_final, cls_emb, text_projection, token_embedding, heads,
    pad_id}``, model/prompt_encoder.py:213-243); ``CONCHPromptEncoder(width=..., heads=..., layers=...)`` builds an empty tower
    of the CONCH text architecture to ``load_state_dict`` into (model/conch/model_configs/conch_ViT-B-16.json: 768 / 12 / 12,
    128 positions, 32007 tokens, 512 outputs)."""

    _transient = {"_cm": lambda: None, "_cm_key": lambda: None, "_cm_arr": None, "_plans": dict, "_pk": lambda: None, "_pk_key": lambda: None,
                  "_pk_bwd": lambda: False, "_tt_cache": None, "_plan_last": None}

    def __init__(self, coca_model=None, *, width: int = 768, heads: int = 12, layers: int = 12, context_length: int = 128,
                 vocab_size: int = 32007, output_dim: int = 512):
        super().__init__()
        if coca_model is not None:
            t = coca_model.text
            self.pad_id = t.pad_id
            assert self.pad_id == 0, "Assume pad_id = 0 in CONCH to encode prompts as expected."
            self.heads = t.heads
            self.positional_embedding = t.positional_embedding
            self.transformer = t.transformer
            self.ln_final = t.ln_final
            self.cls_emb = t.cls_emb
            self.text_projection = t.text_projection
            self.token_embedding = t.token_embedding
            if self.cls_emb is None:
                raise NotImplementedError("the CONCH text tower embeds a CLS token (embed_cls=True); towers without one are not supported")
            self._check_architecture(t)
        else:
            self.pad_id, self.heads = 0, heads
            self.positional_embedding = nn.Parameter(torch.empty(context_length, width))
            self.transformer = _Transformer(width, layers, heads)
            self.ln_final = nn.LayerNorm(width)
            self.cls_emb = nn.Parameter(torch.empty(width))
            self.text_projection = nn.Parameter(torch.empty(width, output_dim))
            self.token_embedding = nn.Embedding(vocab_size, width)
            self.reset_parameters()
        self.context_length = self.positional_embedding.shape[0]
        self.output_dim = self.text_projection.shape[1]
        self.text_config = {"max_num_tokens": self.context_length - 1, "embedding_dim": self.token_embedding.embedding_dim,
                            "embedding_dtype": self.token_embedding.weight.dtype}
        self._cm, self._cm_key, self._plans = None, None, {}
        self._pk, self._pk_key, self._pk_bwd = None, None, False

    def _check_architecture(self, tower):
        """The kernels hard-code the CoCa text block of model/conch/transformer.py:191-247: pre-LN with eps 1e-5, exact (erf)
        GELU, no LayerScale, MLP ratio 4, causal mask.  Anything else would give wrong text features silently: refuse it."""
        def bad(what):
            raise NotImplementedError(f"text tower: {what} -- the HIP tower implements the CONCH text block only "
                                      "(LayerNorm eps 1e-5, erf GELU, no LayerScale, mlp ratio 4, causal attention)")
        width = self.positional_embedding.shape[1]
        lns = [self.ln_final]
        for blk in self.transformer.resblocks:
            lns += [blk.ln_1, blk.ln_2]
            for name in ("ls_1", "ls_2"):
                ls = getattr(blk, name, None)
                if ls is not None and not isinstance(ls, nn.Identity):
                    bad(f"LayerScale ({name})")
            if getattr(blk, "ln_1_kv", None) is not None:
                bad("a cross-attention block")
            act = getattr(blk.mlp, "gelu", None)
            if not isinstance(act, nn.GELU) or getattr(act, "approximate", "none") != "none":
                bad(f"activation {type(act).__name__}")
            if blk.mlp.c_fc.out_features != 4 * width or blk.mlp.c_proj.in_features != 4 * width:
                bad("mlp ratio != 4")
        for ln in lns:
            if abs(float(ln.eps) - 1e-5) > 1e-12 or not getattr(ln, "elementwise_affine", True):
                bad(f"LayerNorm eps {ln.eps}")
        mask = getattr(tower, "attn_mask", None)
        if mask is not None:
            n = mask.shape[-1]
            causal = torch.full((n, n), float("-inf")).triu_(1)
            if not torch.equal(mask.detach().float().cpu(), causal):
                bad("a non-causal attention mask")

    def reset_parameters(self):
        """TextTransformer.init_parameters (model/conch/transformer.py:376-392)."""
        d, L = self.transformer.width, self.transformer.layers
        nn.init.normal_(self.token_embedding.weight, std=0.02)
        nn.init.normal_(self.positional_embedding, std=0.01)
        nn.init.normal_(self.cls_emb, std=0.01)
        proj_std, attn_std, fc_std = (d ** -0.5) * ((2 * L) ** -0.5), d ** -0.5, (2 * d) ** -0.5
        for blk in self.transformer.resblocks:
            nn.init.normal_(blk.attn.in_proj_weight, std=attn_std)
            nn.init.normal_(blk.attn.out_proj.weight, std=proj_std)
            nn.init.normal_(blk.mlp.c_fc.weight, std=fc_std)
            nn.init.normal_(blk.mlp.c_proj.weight, std=proj_std)
        nn.init.normal_(self.text_projection, std=d ** -0.5)

    # -- C-side view of the weights ------------------------------------------------------------------------------------
    def _tower_tensors(self):
        # cached: ~300 nn.Module attribute look-ups per call otherwise, several calls per step.  The Parameter objects survive
        # .to(device) / load_state_dict (in-place), whose effects the callers' (data_ptr, _version) keys still see; re-assigning a
        # parameter attribute is caught by the identity check on the first and last tensors.
        blocks = self.transformer.resblocks
        c = self.__dict__.get("_tt_cache")
        if (c is not None and c[0] == len(blocks) and c[1][0] is self.positional_embedding and c[1][4] is self.text_projection
                and (len(blocks) == 0 or c[1][-1] is blocks[len(blocks) - 1].mlp.c_proj.bias)):
            return c[1]
        ts = [self.positional_embedding, self.cls_emb, self.ln_final.weight, self.ln_final.bias, self.text_projection]
        for blk in blocks:
            ts += [blk.ln_1.weight, blk.ln_1.bias, blk.attn.in_proj_weight, blk.attn.in_proj_bias, blk.attn.out_proj.weight,
                   blk.attn.out_proj.bias, blk.ln_2.weight, blk.ln_2.bias, blk.mlp.c_fc.weight, blk.mlp.c_fc.bias,
                   blk.mlp.c_proj.weight, blk.mlp.c_proj.bias]
        self.__dict__["_tt_cache"] = (len(blocks), ts)
        return ts

    def _c_model(self, device):
        ts = self._tower_tensors()
        key = tuple(t.data_ptr() for t in ts)
        if self._cm is not None and key == self._cm_key:
            return self._cm
        for t in ts:
            if not t.is_cuda or t.device != device:
                raise VlsaNativeError("the text tower runs on the MI355X only: its weights must be on the device of the prompts "
                                      "(there is no CPU fallback)")
            if t.dtype != torch.float32 or not t.is_contiguous():
                raise VlsaNativeError("text tower weights must be contiguous fp32 tensors")
        blocks = list(self.transformer.resblocks)
        arr = (_TTLayer * len(blocks))()
        for i in range(len(blocks)):
            for j, name in enumerate(n for n, _ in _TTLayer._fields_):
                setattr(arr[i], name, ts[5 + 12 * i + j].data_ptr())
        width = self.positional_embedding.shape[1]
        self._cm = _TTModel(width, self.heads, len(blocks), self.output_dim, self.context_length, arr, ts[0].data_ptr(),
                            ts[1].data_ptr(), ts[2].data_ptr(), ts[3].data_ptr(), ts[4].data_ptr())
        self._cm_arr, self._cm_key = arr, key
        return self._cm

    def _packed_weights(self, device, with_backward: bool) -> torch.Tensor:
        """The tiled (MFMA-fragment-major) copies of the tower's matrices the product kernels read, rebuilt when a weight's
        storage or in-place version changes (load_state_dict, .to(device)); the backward set (transposes) only once a
        gradient is asked for."""
        ts = self._tower_tensors()
        mats = [ts[4]] + [ts[5 + 12 * i + j] for i in range((len(ts) - 5) // 12) for j in (2, 4, 8, 10)]   # projection, in / out / fc / proj
        key = tuple((t.data_ptr(), t._version) for t in mats)
        if self._pk is not None and self._pk_key == key and (self._pk_bwd or not with_backward):
            return self._pk
        lib = nat.load()
        model = self._c_model(device)
        nbytes = lib.vlsa_tt_packed_bytes(ctypes.byref(model), int(with_backward))
        if nbytes == 0:
            raise VlsaNativeError("text tower: unsupported shape (width % 128, width <= 768, 64 features per head, out_dim % 64)")
        buf = torch.empty(nbytes, dtype=torch.uint8, device=device)
        nat.check(lib.vlsa_tt_pack_weights(ctypes.byref(model), ctypes.c_void_p(buf.data_ptr()), int(with_backward),
                                           ctypes.c_void_p(torch.cuda.current_stream(device).cuda_stream)), "vlsa_tt_pack_weights")
        self._pk, self._pk_key, self._pk_bwd = buf, key, bool(with_backward)
        return buf

    def _plan(self, pseudo_tokens, device, prefix_len=0) -> _RowPlan:
        # fast path: the very tensor object (a learner's `pseudo_sentence_tokens` buffer) seen last time, unchanged in place;
        # otherwise keyed on CONTENT (the pattern of non-pad positions): `generate_pseudo_tokens` makes a new tensor per call
        prefix_len = int(prefix_len or 0)
        last = self.__dict__.get("_plan_last")
        if (last is not None and last[0] is pseudo_tokens and last[1] == pseudo_tokens._version and last[2].row_seq.device == device
                and last[3] == prefix_len):
            return last[2]
        key = (bytes((pseudo_tokens != 0).to("cpu", torch.uint8).contiguous().numpy().data), tuple(pseudo_tokens.shape), str(device),
               prefix_len)
        plan = self._plans.get(key)
        if plan is None:
            if len(self._plans) > 16:
                self._plans.clear()
            plan = _RowPlan(pseudo_tokens, self.context_length, device, prefix_len)
            self._plans[key] = plan
        self.__dict__["_plan_last"] = (pseudo_tokens, pseudo_tokens._version, plan, prefix_len)
        return plan

    # -- reference API ---------------------------------------------------------------------------------------------------
    def generate_pseudo_tokens(self, text):
        """1, 2, ... up to the token before the first pad, zeros behind it (model/prompt_encoder.py:257-265).  A row without
        any pad yields all zeros there (argmax of an all-False row is 0) -- kept as is."""
        first_pad = (text == self.pad_id).int().argmax(dim=-1)
        pos = torch.arange(text.shape[1], device=text.device)[None, :]
        return torch.where(pos < first_pad[:, None], pos + 1, torch.zeros_like(pos)).to(text.dtype)

    def forward(self, prompts_text=None, prompts_embedding=None, prompts_pseudo_tokens=None, shared_prefix_len=0):
        """prompts_text [n, 128] token ids (last slot = CLS placeholder) or prompts_embedding [n, 127, width] with
        prompts_pseudo_tokens [n, 127] (0 = pad) -> [n, output_dim]   (model/prompt_encoder.py:267-322).
        shared_prefix_len (beyond the reference's signature; 0 = off): the caller's promise that positions 0 .. L-1 hold the SAME
        embedding in every prompt (what the CoOp learners produce: <sot> + shared context, ``learner.shared_prefix_len``); those
        rows are then evaluated once for all prompts -- exact under the causal mask (see ``compact_rows``)."""
        L = self.context_length - 1
        if prompts_text is not None:
            assert prompts_text.shape[1] == L + 1, "Found invalid input of `prompts_text`."
            ids = prompts_text[:, :-1]                              # make space for the CLS token
            if prompts_pseudo_tokens is None:
                prompts_pseudo_tokens = self.generate_pseudo_tokens(ids)
            x = self.token_embedding(ids.to(self.token_embedding.weight.device))
        else:
            assert prompts_embedding is not None, "Found null `prompts_text`, please specify `prompts_embedding`."
            assert prompts_pseudo_tokens is not None, "Found null `prompts_text`, please specify `prompts_pseudo_tokens`."
            x = prompts_embedding
        assert x.shape[1] == L
        if not x.is_cuda:
            raise VlsaNativeError("vlsa_amd runs on MI355X only: got CPU prompt embeddings (there is no CPU fallback; the CPU "
                                  "oracle under oracle/ is test infrastructure)")
        plan = self._plan(prompts_pseudo_tokens, x.device, shared_prefix_len if prompts_text is None else 0)
        if torch.is_grad_enabled() and any(t.requires_grad for t in self._tower_tensors()):
            return _TextTowerTrainFn.apply(x, self, plan, *self._tower_tensors())      # vlsa_txt_encoder_frozen: False
        return _TextTowerFn.apply(x, self, plan)
