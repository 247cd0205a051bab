"""Host-side logic of the text side on CPU: the prompt learners' sentence assembly vs the reference-generated fixtures,
state-dict compatibility of the text tower holder, and the compact-row plan (which positions can reach the CLS token)."""
import numpy as np
import pytest
import torch
import torch.nn as nn

import text_cases as TC
import text_helpers as TH
from oracle import text_oracle as TO


def build_learner(case, inp):
    from vlsa_amd.prompt_learner import RankPromptLearner
    (name, tower, seed, K, base, position) = case
    E = nn.Embedding.from_pretrained(inp["W"]["token_embedding.weight"].clone(), freeze=True)
    cfg = dict(max_num_tokens=127, embedding_dim=E.embedding_dim, embedding_dtype=torch.float32)
    return RankPromptLearner(text_config=cfg, tokenizer=inp["tok"], token_embedding=E, num_base_ranks=base, num_ranks=K,
                             num_tokens_per_rank=4, num_context_tokens=8, rank_tokens_position=position,
                             init_context=inp["ctx_key"], init_rank_names=inp["rank_keys"])


@pytest.mark.parametrize("case", TC.RANK_CASES, ids=[c[0] for c in TC.RANK_CASES])
def test_rank_prompt_learner_matches_reference(case):
    inp = TH.rank_case_inputs(case)
    fx, table, E = inp["fx"], inp["table"], inp["W"]["token_embedding.weight"]
    pl = build_learner(case, inp)
    # initial state = embeddings of the tokenised init prompt (rank names right-padded with <eot>/<pad> ids, as the raw rows are)
    assert torch.equal(pl.context_embeds.detach(), E[torch.tensor(table["ctx"])])
    bos, eos, pad = inp["special"]
    tmax = max(len(table[k]) for k in inp["rank_keys"])
    for i, k in enumerate(inp["rank_keys"]):
        ids = (table[k] + [eos] + [pad] * tmax)[:tmax]
        assert torch.equal(pl.rank_embeds[i].detach(), E[torch.tensor(ids)])
    assert set(dict(pl.named_parameters())) == {"context_embeds", "rank_embeds"}
    assert len(pl.state_dict()) == 2                                    # buffers are non-persistent, as in the reference
    assert torch.equal(pl.pseudo_sentence_tokens, torch.from_numpy(fx["pseudo"]))
    assert np.abs(pl.interpolation_weights.numpy() - fx["interp"]).max() < 1e-6
    with torch.no_grad():
        pl.context_embeds.copy_(torch.from_numpy(fx["context_embeds"]))
        pl.rank_embeds.copy_(torch.from_numpy(fx["rank_embeds"]))
    sent = pl()
    s = sent.detach().double()
    assert np.allclose([float(s.sum()), float((s ** 2).sum())], fx["sentence_checksum"], rtol=1e-6)
    _, leaves = TH.oracle_rank_case(case)
    # same assembly as the loop form of the reference (the interpolation is an einsum here: fp32 summation order differs)
    assert (sent.detach() - leaves["sentence"]).abs().max().item() < 1e-7
    # differentiable w.r.t. both parameters
    sent.sum().backward()
    assert pl.context_embeds.grad is not None and pl.rank_embeds.grad is not None


def test_plain_prompt_learner_ragged_rank_lengths():
    from vlsa_amd.prompt_learner import PlainPromptLearner
    table, ctx_key, names = TC.synthetic_prompt_table(64, 77, n_ctx=3, rank_lens=(2, 4, 1))
    tok = TC.ReplayTokenizer(table)
    E = nn.Embedding(64, 32)
    cfg = dict(max_num_tokens=127, embedding_dim=32, embedding_dtype=torch.float32)
    for position in ("tail", "front", "middle"):
        pl = PlainPromptLearner(cfg, tok, E, num_ranks=3, num_tokens_per_rank=4, num_context_tokens=8,
                                rank_tokens_position=position, init_context=ctx_key, init_rank_names=names)
        assert pl.num_tokens_per_rank == [2, 4, 1] and pl.num_context_tokens == 3
        assert [int((p > 0).sum()) for p in pl.pseudo_sentence_tokens] == [1 + 3 + n + 2 for n in (2, 4, 1)]
        sent = pl()
        for i, n in enumerate((2, 4, 1)):
            ctx, rk = pl.context_embeds.detach(), pl.rank_embeds[i, :n].detach()
            want = {"tail": torch.cat([ctx, rk]), "front": torch.cat([rk, ctx]), "middle": torch.cat([ctx[:1], rk, ctx[1:]])}[position]
            assert torch.equal(sent[i, 1:1 + 3 + n].detach(), want)
            assert torch.equal(sent[i, 1 + 3 + n].detach(), E.weight[table["X."][1]].detach())      # "."
            assert torch.equal(sent[i, 2 + 3 + n].detach(), E.weight[2].detach())                   # <eot>
            assert torch.equal(sent[i, 3 + 3 + n].detach(), E.weight[0].detach())                   # <pad>


def test_text_tower_holder_state_dict_keys():
    from vlsa_amd.prompt_encoder import CONCHPromptEncoder
    c = TC.TOWERS["small"]
    enc = CONCHPromptEncoder(width=c["width"], heads=c["heads"], layers=c["layers"], vocab_size=c["vocab"], output_dim=c["out_dim"])
    W = TC.make_tower_weights("small", 5)
    assert set(enc.state_dict()) == set(W)                # the reference's names (model/conch/transformer.py:326-372)
    enc.load_state_dict(W)
    assert enc.text_config == {"max_num_tokens": 127, "embedding_dim": c["width"], "embedding_dtype": torch.float32}
    with pytest.raises(Exception):
        enc(prompts_embedding=torch.zeros(2, 127, c["width"]), prompts_pseudo_tokens=torch.ones(2, 127))   # CPU tensors: no fallback


def test_compact_rows_follow_the_shifted_cls_mask():
    from vlsa_amd.prompt_encoder import compact_rows
    L = 127
    pt = torch.zeros(4, L, dtype=torch.long)
    pt[0, :11] = torch.arange(1, 12)          # ordinary sentence of 11 tokens
    pt[1, :] = torch.arange(1, L + 1)         # fills every slot
    # row 2: all zeros (what generate_pseudo_tokens yields for a sentence without any pad)
    pt[3, :5] = 1; pt[3, 8] = 1               # a hole in the pattern
    rp = compact_rows(pt, 128)
    lens = [rp["seq_row0"][i + 1] - rp["seq_row0"][i] for i in range(4)]
    assert lens == [13, 128, 2, 11]           # n + 2 | 127 tokens + CLS | position 0 + CLS | positions 0..9 + CLS
    r0 = rp["seq_row0"]
    assert rp["cls_keep"][r0[0]:r0[1]] == [1] * 12 + [0]                    # sees 0..11, not itself
    assert rp["cls_keep"][r0[1]:r0[2]] == [1] * 128                         # full sentence: also itself
    assert rp["cls_keep"][r0[2]:r0[3]] == [1, 0]
    assert rp["cls_keep"][r0[3]:r0[4]] == [1] * 6 + [0, 0, 0, 1, 0]         # column 0, 1..5 (tokens 0..4), 9 (token 8)
    # cross-check against the dense mask of the oracle
    import torch.nn.functional as F
    keep = F.pad((pt != 0).unsqueeze(1), (1, 0, L, 0), value=True)[:, -1]   # last row of the [128, 128] mask
    for s in range(4):
        cols = [c for c in range(128) if keep[s, c]]
        got = [rp["row_pos"][r0[s] + j] for j in range(lens[s]) if rp["cls_keep"][r0[s] + j]]
        assert got == cols


def test_compact_rows_with_a_shared_prefix():
    """K rank prompts `<sot> ctx x 4 | rank x 4 . <eot>`: with prefix_len = 5 the five leading rows are planned once and every prompt
    keeps its 7 own positions + CLS: 5 + K * 8 rows instead of K * 13; the plan falls back to no sharing when a prompt does not
    reach past the prefix or one of its prefix positions is padding."""
    from vlsa_amd.prompt_encoder import compact_rows
    K, L = 12, 127
    pt = torch.zeros(K, L, dtype=torch.long)
    pt[:, :11] = torch.arange(1, 12)
    full = compact_rows(pt, 128)
    rp = compact_rows(pt, 128, prefix_len=5)
    assert full["M"] == K * 13 and rp["M"] == 5 + K * 8 and rp["prefix_len"] == 5
    assert rp["seq_row0"][0] == 5 and rp["seq_row0"][-1] == rp["M"] and rp["max_len"] == full["max_len"] == 13
    assert rp["row_pos"][:5] == [0, 1, 2, 3, 4] and rp["row_seq"][:5] == [0] * 5 and rp["row_src"][:5] == [0, 1, 2, 3, 4]
    for s in range(K):
        a, b = rp["seq_row0"][s], rp["seq_row0"][s + 1]
        assert rp["row_pos"][a:b] == list(range(5, 12)) + [127] and rp["row_seq"][a:b] == [s] * 8
        assert rp["row_src"][a:b] == list(range(5, 12)) + [-1] and rp["cls_keep"][a:b] == [1] * 7 + [0]
    short = pt.clone(); short[3, 3:] = 0                      # prompt 3 ends inside the would-be prefix
    assert compact_rows(short, 128, prefix_len=5)["prefix_len"] == 0
    assert compact_rows(pt[:1], 128, prefix_len=5)["prefix_len"] == 0        # a single prompt has nothing to share


def test_learners_report_their_shared_prefix():
    import text_cases as TC
    for case, want in ((TC.RANK_CASES[0], 1 + 4), (TC.RANK_CASES[2], 1), (TC.RANK_CASES[3], 1 + 3 // 2)):   # tail / front / middle
        pl = build_learner(case, __import__("text_helpers").rank_case_inputs(case))
        assert pl.shared_prefix_len == want, (case[0], pl.shared_prefix_len)
