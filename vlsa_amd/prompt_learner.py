"""CoOp-style prompt learners of the text side -- drop-in counterparts of the reference's ``PlainPromptLearner`` /
``RankPromptLearner`` (model/prompt_learners/plain_prompt_learner.py:13-279, rank_prompt_learner.py:14-156): same
constructor keywords, same parameters (``context_embeds``, ``rank_embeds``: the state-dict keys of the shipped checkpoint,
``prompt_learner.context_embeds [4, 768]`` / ``prompt_learner.rank_embeds [4, 4, 768]``), same non-persistent buffers
(``pseudo_sentence_tokens``, ``sentence_embeds``, ``interpolation_weights``), same ``forward() -> [num_ranks, 127, dim]``.

A prompt is ``<sot> context.. rank.. . <eot> <pad>..`` in embedding space; the rank learner derives the K rank embeddings
from ``num_base_ranks`` learnable ones by fixed interpolation weights (ordinality prior).  ``forward`` assembles all K
sentences with one gather (the reference loops over the ranks); it is differentiable w.r.t. both parameters.
The tokenizer is duck-typed exactly as the reference uses it: ``tok(text | [texts], return_raw_tokens, return_num_tokens)``
with ``bos_token_id / eos_token_id / pad_token_id`` (model/utils_vl.py:19-75).
"""
from __future__ import annotations

import json
from typing import List, Optional, Sequence, Union

import torch
import torch.nn as nn

from ._native import TransientCaches as _TransientCaches

__all__ = ["PlainPromptLearner", "RankPromptLearner", "load_prompt_learner"]


def _read_init_prompt(path, context_idx, rank_idx):
    """{"context_templates": [...], "class_names": {"0": [...], ...}} -> (context, [rank names])  (utils/io.py:151-173)."""
    if path is None:
        return None, None
    with open(path, "r") as f:
        spec = json.load(f)
    return spec["context_templates"][context_idx], [names[rank_idx] for names in spec["class_names"].values()]


def _spread_names(names: Sequence[str], n: int) -> List[str]:
    """n rank names out of the candidates: evenly spaced picks when there are too many, block-wise repeats when too few."""
    c = len(names)
    if c > n:
        picks = torch.linspace(0, c - 1, n).to(torch.int32).tolist()      # truncation, as numpy's astype(int32)
        return [names[i] for i in picks]
    if c < n:
        block = n // c
        return [names[min(i // block, c - 1)] for i in range(n)]
    return list(names)


class _SentenceFn(torch.autograd.Function):
    """Sentence assembly on the device (vlsa_prompt_sentences / _backward): what ``forward`` below does with torch ops, in one
    launch each way -- the optimizer step is bound by the number of dependent launches."""

    @staticmethod
    def forward(ctx, context, rank, module):
        from . import _native as nat
        lib = nat.load()
        order, pos = module._hip_plan()
        templ = module.sentence_embeds
        R, L, dim = templ.shape
        S = order.shape[1]
        T = rank.shape[1]
        C = S - T
        interp = module._interp()
        cont, rk = context.detach().contiguous(), rank.detach().contiguous()
        ip = None if interp is None else interp.detach().float().contiguous()
        out = torch.empty_like(templ)
        st = torch.cuda.current_stream(templ.device).cuda_stream
        nat.check(lib.vlsa_prompt_sentences(templ.data_ptr(), cont.data_ptr(), int(context.dim() == 3), rk.data_ptr(),
                                            None if ip is None else ip.data_ptr(), 0 if ip is None else ip.shape[1], order.data_ptr(),
                                            R, L, S, C, T, dim, out.data_ptr(), st), "vlsa_prompt_sentences")
        ctx.meta = (module, ip, context.shape, rank.shape, (R, L, S, C, T, dim))
        return out

    @staticmethod
    def backward(ctx, dout):
        from . import _native as nat
        lib = nat.load()
        module, ip, cshape, rshape, (R, L, S, C, T, dim) = ctx.meta
        _, pos = module._hip_plan()
        dout = dout.contiguous()
        dcontext = torch.empty(cshape, dtype=torch.float32, device=dout.device)
        drank = torch.empty(rshape, dtype=torch.float32, device=dout.device)
        st = torch.cuda.current_stream(dout.device).cuda_stream
        nat.check(lib.vlsa_prompt_sentences_backward(dout.data_ptr(), pos.data_ptr(), int(len(cshape) == 3),
                                                     None if ip is None else ip.data_ptr(), 0 if ip is None else ip.shape[1],
                                                     rshape[0], R, L, S, C, T, dim, dcontext.data_ptr(), drank.data_ptr(), st),
                  "vlsa_prompt_sentences_backward")
        return dcontext, drank, None


class PlainPromptLearner(_TransientCaches, nn.Module):
    """One learnable embedding block per rank (model/prompt_learners/plain_prompt_learner.py)."""
    _transient = {"_hip_tables": None}

    rank_tokens_position_candidates = {"tail", "middle", "front"}

    def __init__(self, text_config, tokenizer, token_embedding, num_ranks: int, num_tokens_per_rank: Union[int, List[int]],
                 num_context_tokens: int, rank_tokens_position: str = "tail", init_prompt_path: Optional[str] = None,
                 init_prompt_context_idx: int = 0, init_prompt_rank_idx: int = 0, rank_specific_context: bool = False,
                 init_context: Optional[str] = None, init_rank_names: Optional[Sequence[str]] = None, **kwargs):
        super().__init__()
        self._setup(text_config, tokenizer, token_embedding, num_ranks, num_ranks, num_tokens_per_rank, num_context_tokens,
                    rank_tokens_position, init_prompt_path, init_prompt_context_idx, init_prompt_rank_idx, rank_specific_context,
                    init_context, init_rank_names, uniform_rank_length=False)

    # shared by both learners: `num_embed_ranks` blocks of rank embeddings serve `num_ranks` sentences
    def _setup(self, text_config, tokenizer, token_embedding, num_ranks, num_embed_ranks, num_tokens_per_rank, num_context_tokens,
               position, init_prompt_path, ctx_idx, rank_idx, rank_specific_context, init_context, init_rank_names,
               uniform_rank_length):
        self.cfg_max_num_tokens = text_config["max_num_tokens"]
        self.cfg_embedding_dim = text_config["embedding_dim"]
        self.cfg_embedding_dtype = text_config["embedding_dtype"]
        if position not in self.rank_tokens_position_candidates:
            raise ValueError(f"Got an invalid rank_tokens_position: {position}.")
        if init_prompt_path is not None:
            init_context, init_rank_names = _read_init_prompt(init_prompt_path, ctx_idx, rank_idx)
        dim, dt = self.cfg_embedding_dim, self.cfg_embedding_dtype
        dev = token_embedding.weight.device          # the learner is built where the token embedding lives (CPU or GPU)
        # ---- context: embeddings of the tokenised init text, or N(0, 0.02) ---------------------------------------------
        if init_context is not None:
            ids, n = tokenizer(init_context.replace("_", " "), return_raw_tokens=True, return_num_tokens=True)
            num_context_tokens = int(n)
            with torch.no_grad():
                ctx = token_embedding(ids.to(dev)).detach().clone()
            assert ctx.shape[0] == num_context_tokens
            if rank_specific_context:
                ctx = ctx[None].repeat(num_ranks, 1, 1)
        else:
            shape = (num_ranks, num_context_tokens, dim) if rank_specific_context else (num_context_tokens, dim)
            ctx = torch.empty(shape, dtype=dt, device=dev)
            nn.init.normal_(ctx, std=0.02)
        self.context_embeds = nn.Parameter(ctx)
        # ---- rank names: embeddings of the (right-padded) raw token rows, or N(0, 0.02) ---------------------------------
        if isinstance(num_tokens_per_rank, int):
            num_tokens_per_rank = [num_tokens_per_rank] * num_embed_ranks
        if init_rank_names is not None:
            names = _spread_names(list(init_rank_names), num_embed_ranks)
            ids, counts = tokenizer(names, return_raw_tokens=True, return_num_tokens=True)
            counts = [int(c) for c in counts]
            if max(counts) > self.cfg_max_num_tokens - num_context_tokens - 3:      # <sot>, <full stop>, <eot>
                raise ValueError(f"The rank name is too long: {names[counts.index(max(counts))]}.")
            with torch.no_grad():
                rk = token_embedding(ids.to(dev)).detach().clone()
            assert rk.shape[1] == max(counts)
        else:
            counts = list(num_tokens_per_rank)
            if self.cfg_max_num_tokens < num_context_tokens + max(counts) + 3:
                raise ValueError(f"The value of `max_num_tokens_per_rank` ({max(counts)}) is too large.")
            rk = torch.empty((num_embed_ranks, max(counts), dim), dtype=dt, device=dev)
            nn.init.normal_(rk, std=0.02)
        self.rank_embeds = nn.Parameter(rk)
        assert len(rk) == num_embed_ranks
        # the rank learner gives every sentence the longest rank-name length (rank_prompt_learner.py:70)
        self.num_tokens_per_rank = [max(counts)] * num_ranks if uniform_rank_length else counts
        self.num_context_tokens = num_context_tokens
        self.rank_tokens_position = position
        self.rank_tokens_positon = position          # (sic) the plain learner's attribute name in the reference
        self.num_ranks = num_ranks
        # ---- pseudo tokens: 1.. over <sot> context rank . <eot>, 0 behind ---------------------------------------------
        L = self.cfg_max_num_tokens
        pseudo = torch.zeros(num_ranks, L, dtype=torch.long)
        for i, nt in enumerate(self.num_tokens_per_rank):
            n = 1 + num_context_tokens + nt + 2
            pseudo[i, :n] = torch.arange(1, n + 1)
        self.register_buffer("pseudo_sentence_tokens", pseudo.to(dev), persistent=False)
        # ---- sentence template: pad everywhere, <sot> first, "." and <eot> closing the sentence --------------------------
        with torch.no_grad():
            row = tokenizer("X.", return_raw_tokens=False, return_num_tokens=False)
            row = row if row.dim() == 1 else row[0]
            assert int(row[0]) == tokenizer.bos_token_id and int(row[3]) == tokenizer.eos_token_id, 'expected "X." -> <sot> X . <eot>'
            emb = lambda i: token_embedding(torch.tensor([int(i)], dtype=torch.long, device=token_embedding.weight.device))[0].detach()  # noqa: E731
            pad_e, sot_e, dot_e, eot_e = emb(tokenizer.pad_token_id), emb(row[0]), emb(row[2]), emb(row[3])
        template = pad_e[None, None].repeat(num_ranks, L, 1).clone()
        last = pseudo.argmax(dim=-1)
        ar = torch.arange(num_ranks)
        template[ar, 0] = sot_e
        template[ar, last] = eot_e
        template[ar, last - 1] = dot_e
        self.register_buffer("sentence_embeds", template, persistent=False)
        # ---- gather plan of forward(): slot p of sentence i takes row order[i, p] of cat([context_i, rank_i]) ----------------
        C, Tmax = num_context_tokens, rk.shape[1]
        order = torch.zeros(num_ranks, C + Tmax, dtype=torch.long)
        valid = torch.zeros(num_ranks, C + Tmax, dtype=torch.bool)
        for i, nt in enumerate(self.num_tokens_per_rank):
            c_idx, r_idx = list(range(C)), [C + t for t in range(nt)]
            if position == "tail":
                seq = c_idx + r_idx
            elif position == "front":
                seq = r_idx + c_idx
            else:
                seq = c_idx[:C // 2] + r_idx + c_idx[C // 2:]
            order[i, :len(seq)] = torch.tensor(seq, dtype=torch.long)
            valid[i, :len(seq)] = True
        self.register_buffer("_order", order.to(dev), persistent=False)
        self.register_buffer("_valid", valid.to(dev), persistent=False)
        # leading sentence positions that hold the same embedding in EVERY rank's sentence: <sot>, then the context tokens that
        # come before the first rank token (all of them with the rank tokens at the tail, the first half in the middle position,
        # none at the front) -- unless every rank has its own context.  The text tower evaluates those rows once.
        lead = {"tail": C, "middle": C // 2, "front": 0}[position]
        self.shared_prefix_len = 1 + (0 if rank_specific_context else lead)

    def _rank_rows(self) -> torch.Tensor:
        return self.rank_embeds                       # [num_ranks, T, dim]

    def _interp(self):
        return None                                   # plain learner: one embedding row per rank

    def _hip_plan(self):
        """int32 slot -> source table (-1: unused) and its inverse (source -> sentence position) for vlsa_prompt_sentences"""
        p = self.__dict__.get("_hip_tables")
        if p is None or p[0].device != self._order.device:
            order = torch.where(self._valid, self._order, torch.full_like(self._order, -1)).to(torch.int32).contiguous()
            R, S = order.shape
            pos = torch.full((R, S), -1, dtype=torch.int32)
            oc, vc = self._order.cpu(), self._valid.cpu()
            for i in range(R):
                for s_ in range(S):
                    if bool(vc[i, s_]):
                        pos[i, int(oc[i, s_])] = s_ + 1
            p = self.__dict__["_hip_tables"] = (order, pos.to(order.device))
        return p

    def forward(self):
        ctx = self.context_embeds
        if (ctx.is_cuda and ctx.dtype == torch.float32 and self.rank_embeds.dtype == torch.float32
                and self.sentence_embeds.dtype == torch.float32 and not getattr(self, "_torch_ops_only", False)):
            return _SentenceFn.apply(ctx, self.rank_embeds, self)        # one HIP launch forward, one backward
        if ctx.dim() == 2:
            ctx = ctx[None].expand(self.num_ranks, *ctx.shape)
        src = torch.cat([ctx, self._rank_rows()], dim=1)                                        # [R, C + T, dim]
        picked = src.gather(1, self._order[..., None].expand(-1, -1, src.shape[-1]))
        n = picked.shape[1]
        out = self.sentence_embeds.clone()
        out[:, 1:1 + n] = torch.where(self._valid[..., None], picked, out[:, 1:1 + n])
        return out

    def load_pretrained_parameters(self, ckpt_path):
        ckpt = torch.load(ckpt_path, map_location="cpu", weights_only=True)
        for name in ("context_embeds", "rank_embeds"):
            t = ckpt["model"]["prompt_learner." + name]
            assert t.shape == getattr(self, name).shape
            setattr(self, name, nn.Parameter(t.to(getattr(self, name).device)))


class RankPromptLearner(PlainPromptLearner):
    """``num_base_ranks`` learnable rank embeddings interpolated to ``num_ranks`` sentences (rank_prompt_learner.py:14-156)."""

    def __init__(self, text_config, tokenizer, token_embedding, num_base_ranks: int, num_ranks: int,
                 num_tokens_per_rank: Union[int, List[int]], num_context_tokens: int, rank_tokens_position: str = "tail",
                 init_prompt_path: Optional[str] = None, init_prompt_context_idx: int = 0, init_prompt_rank_idx: int = 0,
                 rank_specific_context: bool = False, interpolation_type: str = "linear",
                 init_context: Optional[str] = None, init_rank_names: Optional[Sequence[str]] = None, **kwargs):
        nn.Module.__init__(self)
        self._setup(text_config, tokenizer, token_embedding, num_ranks, num_base_ranks, num_tokens_per_rank, num_context_tokens,
                    rank_tokens_position, init_prompt_path, init_prompt_context_idx, init_prompt_rank_idx, rank_specific_context,
                    init_context, init_rank_names, uniform_rank_length=True)
        self.num_base_ranks = num_base_ranks
        self.register_buffer("interpolation_weights",
                             self.create_interpolation_weights(num_base_ranks, num_ranks, interpolation_type).to(self.rank_embeds.device),
                             persistent=False)

    def create_interpolation_weights(self, num_base_ranks, num_ranks, interpolation_type):
        """[num_ranks, num_base_ranks], rows sum to 1: closeness of rank r to the base ranks spread evenly over 0..K-1."""
        dt = self.cfg_embedding_dtype
        pos = torch.arange(num_ranks)[:, None].repeat(1, num_base_ranks).to(dt)
        anchors = (torch.linspace(0, num_ranks - 1, 3)[1:2] if num_base_ranks == 1 else torch.linspace(0, num_ranks - 1, num_base_ranks)).to(dt)
        dist = (pos - anchors[None]).abs()
        if interpolation_type == "linear":
            w = 1.0 - dist / (num_ranks - 1)
        elif interpolation_type == "inv_prop":
            w = 1.0 / (dist + 1e-5)
        elif interpolation_type == "normal":
            w = torch.exp(-dist * dist)
        else:
            raise ValueError(f"Got an invalide interpolation_type: {interpolation_type}.")
        return w / w.sum(dim=-1, keepdim=True)

    def _rank_rows(self) -> torch.Tensor:
        return torch.einsum("rb,btd->rtd", self.interpolation_weights, self.rank_embeds)

    def _interp(self):
        return self.interpolation_weights


def load_prompt_learner(method: str, cfg: dict):
    """model/prompt_learners/__init__.py: 'plain' | 'rank'."""
    if method == "rank":
        return RankPromptLearner(**cfg)
    if method == "plain":
        return PlainPromptLearner(**cfg)
    raise ValueError(f"{method} is not a valid prompt learner.")
