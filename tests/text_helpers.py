"""Rebuild the inputs of a text-side golden case (tests/golden/text_*.npz) from seeds + the ids stored in the fixture."""
import os

import numpy as np
import torch

import text_cases as TC

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load(name):
    return dict(np.load(os.path.join(GOLDEN, f"text_{name}.npz")))


def table_of(fx):
    return {k[4:]: [int(v) for v in fx[k]] for k in fx if k.startswith("ids.")}


def rank_case_inputs(case):
    """-> dict(W, heads, layers, fx, table, tok, ctx_key, rank_keys, sizes) for a RANK_CASES entry."""
    (name, tower, seed, K, base, position) = case
    fx = load(name)
    W = TC.make_tower_weights(tower, seed)
    table = table_of(fx)
    bos, eos, pad = [int(v) for v in fx["special_ids"]]
    tok = TC.ReplayTokenizer(table, bos, eos, pad)
    rank_keys = [f"rank{i}" for i in range(int(fx["rank_keys"]))]
    c = TC.TOWERS[tower]
    return dict(W=W, heads=c["heads"], layers=c["layers"], fx=fx, table=table, tok=tok, ctx_key="ctx", rank_keys=rank_keys,
                special=(bos, eos, pad), cfg=c)


def oracle_rank_case(case, requires_grad=False):
    """Text features of a rank case through the CPU oracle.  Returns (features [K, out], leaves dict)."""
    from oracle import text_oracle as TO
    (name, tower, seed, K, base, position) = case
    inp = rank_case_inputs(case)
    fx, W, table = inp["fx"], inp["W"], inp["table"]
    bos, eos, pad = inp["special"]
    E = W["token_embedding.weight"]
    ctx = torch.from_numpy(fx["context_embeds"]).clone().requires_grad_(requires_grad)
    rk = torch.from_numpy(fx["rank_embeds"]).clone().requires_grad_(requires_grad)
    pseudo = TO.pseudo_sentence_tokens(K, ctx.shape[0], rk.shape[1])
    assert torch.equal(pseudo, torch.from_numpy(fx["pseudo"]))
    xdot = table["X."]
    template = TO.sentence_template(E[pad], E[bos], E[eos], E[xdot[1]], pseudo)
    interp = TO.interpolation_weights(base, K)
    assert np.abs(interp.numpy() - fx["interp"]).max() < 1e-6
    sent = TO.rank_prompt_learner_forward(ctx, rk, template, interp, K, position)
    feats = TO.prompt_encoder_forward(W, inp["heads"], sent, pseudo, inp["layers"])
    return feats, dict(context=ctx, rank=rk, sentence=sent, pseudo=pseudo)
