"""Seeded CoCa-text-tower weights and prompt-learner inputs for the text-side fixtures.
The reference's gated CONCH weights are not available, so the fixtures use a RANDOM tower of the same architecture whose
every tensor is regenerated from a seed here (identical torch CPU generator on the build container and the GPU box).
Names follow the reference's state dict (model/conch/transformer.py:326-372)."""
from __future__ import annotations

import torch

# name -> (width, heads, layers, vocab, context_length, out_dim)
TOWERS = {
    "conch": dict(width=768, heads=12, layers=12, vocab=32007, ctx=128, out_dim=512),   # conch_ViT-B-16.json text_cfg
    "small": dict(width=128, heads=2, layers=2, vocab=64, ctx=128, out_dim=64),
    "mid": dict(width=256, heads=4, layers=3, vocab=64, ctx=128, out_dim=128),
    "train": dict(width=128, heads=2, layers=2, vocab=64, ctx=128, out_dim=512),    # text features in the CONCH space [K, 512]
}


# Round 6 ("hot" cases): weight seeds whose in_proj / c_fc weights are rescaled AFTER the draw.  With the init stds the attention logits of a
# random tower stay within |.| ~ 3 and the GELU inputs within ~ 2.5 -- a trained CONCH tower does not; the factors below drive the logits to
# |.| ~ 30-60 (softmax rows close to one-hot, large exp arguments) and the c_fc outputs to |.| ~ 6-10 (GELU tails, erf saturating) -- ranges
# the 1e-4 gate had never seen (VERDICT r5 "What's weak" 1).  seed -> (in_proj factor, c_fc factor).
WEIGHT_SCALES = {9021: (3.5, 3.5), 9022: (4.0, 3.0), 9023: (3.0, 4.0)}


def make_tower_weights(name: str, seed: int):
    """Every tensor ~ N(0, std) with the stds of TextTransformer.init_parameters (transformer.py:376-392) -- except that
    biases and LayerNorm affine parameters are made non-trivial so that they are exercised; seeds listed in WEIGHT_SCALES get
    their attention input projection and c_fc weights multiplied by the factors there."""
    c = TOWERS[name]
    d, L = c["width"], c["layers"]
    g = torch.Generator().manual_seed(seed)
    n = lambda *s, std: torch.randn(*s, generator=g) * std   # noqa: E731
    proj_std, attn_std, fc_std = (d ** -0.5) * ((2 * L) ** -0.5), d ** -0.5, (2 * d) ** -0.5
    W = {"token_embedding.weight": n(c["vocab"], d, std=0.02), "positional_embedding": n(c["ctx"], d, std=0.01),
         "cls_emb": n(d, std=0.01), "text_projection": n(d, c["out_dim"], std=d ** -0.5),
         "ln_final.weight": 1 + n(d, std=0.1), "ln_final.bias": n(d, std=0.05)}
    for i in range(L):
        p = f"transformer.resblocks.{i}."
        W[p + "ln_1.weight"], W[p + "ln_1.bias"] = 1 + n(d, std=0.1), n(d, std=0.05)
        W[p + "attn.in_proj_weight"], W[p + "attn.in_proj_bias"] = n(3 * d, d, std=attn_std), n(3 * d, std=0.02)
        W[p + "attn.out_proj.weight"], W[p + "attn.out_proj.bias"] = n(d, d, std=proj_std), n(d, std=0.02)
        W[p + "ln_2.weight"], W[p + "ln_2.bias"] = 1 + n(d, std=0.1), n(d, std=0.05)
        W[p + "mlp.c_fc.weight"], W[p + "mlp.c_fc.bias"] = n(4 * d, d, std=fc_std), n(4 * d, std=0.02)
        W[p + "mlp.c_proj.weight"], W[p + "mlp.c_proj.bias"] = n(d, 4 * d, std=proj_std), n(d, std=0.02)
    if seed in WEIGHT_SCALES:
        s_in, s_fc = WEIGHT_SCALES[seed]
        for i in range(L):
            p = f"transformer.resblocks.{i}."
            W[p + "attn.in_proj_weight"] = W[p + "attn.in_proj_weight"] * s_in
            W[p + "mlp.c_fc.weight"] = W[p + "mlp.c_fc.weight"] * s_fc
    return W


# ---- fixture cases -------------------------------------------------------------------------------------------
# rank prompt learner through the tower: name, tower, weight seed, num_ranks (K), num_base_ranks, rank_tokens_position
RANK_CASES = [
    ("rank_conch_k12", "conch", 9001, 12, 4, "tail"),
    ("rank_conch_k4", "conch", 9002, 4, 4, "tail"),
    ("rank_small_k8_front", "small", 9003, 8, 4, "front"),
    ("rank_mid_k5_middle", "mid", 9004, 5, 3, "middle"),
    ("rank_conch_k12_hot", "conch", 9021, 12, 4, "tail"),        # round 6: rescaled weights (WEIGHT_SCALES)
    ("rank_mid_k6_hot", "mid", 9022, 6, 4, "tail"),
]
# tokenised texts (prompts_text path: the PromptAdapter's prototype prompts): name, tower, seed, sentence lengths (tokens
# between <sot> and <eot>)
TEXT_CASES = [
    ("text_conch", "conch", 9011, [9, 14, 3, 20, 9, 1]),
    ("text_small", "small", 9012, [5, 125, 1, 30]),
    # round 6: rescaled weights AND a pad pattern unlike the shipped prompts' (lengths from 1 to 100 tokens in one call: every prompt's
    # CLS mask differs; 225 compact rows = 15 row tiles)
    ("text_conch_hot_padmix", "conch", 9023, [2, 60, 17, 1, 33, 100]),
]


class ReplayTokenizer:
    """Stand-in for the reference's tokenizer wrapper (model/utils_vl.py:19-75) driven by a fixed text -> ids table, so
    that prompt learners can be built identically on the build container (where the ids of the real CONCH cases were
    captured from the reference's tokenizer) and on the GPU box (no tokenizer files there).  Same call contract:
    ``tok(text | [texts], return_raw_tokens, return_num_tokens)``; a full row is <bos> ids.. <eos> <pad>.. of length 128;
    'raw' strips <bos> and cuts after the longest sentence; the count excludes <bos>/<eos>."""

    def __init__(self, table, bos=1, eos=2, pad=0, length=128):
        self.table = {k: list(v) for k, v in table.items()}
        self.bos_token_id, self.eos_token_id, self.pad_token_id, self.length = bos, eos, pad, length

    def __call__(self, text, return_raw_tokens=True, return_num_tokens=True):
        texts = [text] if isinstance(text, str) else list(text)
        rows = torch.full((len(texts), self.length), self.pad_token_id, dtype=torch.long)
        cnt = torch.zeros(len(texts), dtype=torch.long)
        for i, t in enumerate(texts):
            ids = self.table[t]
            rows[i, 0] = self.bos_token_id
            rows[i, 1:1 + len(ids)] = torch.tensor(ids, dtype=torch.long)
            rows[i, 1 + len(ids)] = self.eos_token_id
            cnt[i] = len(ids)
        out = rows[:, 1:int(cnt.max()) + 1] if return_raw_tokens else rows
        if isinstance(text, str):
            out, cnt = out[0], cnt[0]
        return (out, cnt) if return_num_tokens else out


def synthetic_prompt_table(vocab: int, seed: int, n_ctx: int = 4, rank_lens=(4, 3, 3, 4)):
    """A made-up context + rank names with ids < vocab (3.. are ordinary tokens; 0 pad, 1 bos, 2 eos)."""
    g = torch.Generator().manual_seed(seed)
    ids = lambda n: torch.randint(3, vocab, (n,), generator=g).tolist()   # noqa: E731
    table = {"X.": [ids(1)[0], 3 + (seed % (vocab - 3))], "ctx": ids(n_ctx)}
    names = []
    for i, n in enumerate(rank_lens):
        table[f"rank{i}"] = ids(n)
        names.append(f"rank{i}")
    return table, "ctx", names
