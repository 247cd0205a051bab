"""Text-side oracle (oracle/text_oracle.py) vs the fixtures the reference produced (tests/golden/make_golden_text.py):
rank prompt learner + CoCa text tower forward, gradients w.r.t. the learnable context / rank embeddings, and the
tokenised-text path.  CPU only; the full-size ('conch') tower takes a few seconds per case."""
import numpy as np
import pytest
import torch

import text_cases as TC
import text_helpers as TH
from oracle import text_oracle as TO

torch.set_num_threads(max(4, torch.get_num_threads()))


@pytest.mark.parametrize("case", TC.RANK_CASES, ids=[c[0] for c in TC.RANK_CASES])
def test_rank_prompt_text_features_and_gradients(case):
    fx = TH.load(case[0])
    feats, leaves = TH.oracle_rank_case(case, requires_grad=True)
    s = leaves["sentence"].detach().double()
    assert np.allclose([float(s.sum()), float((s ** 2).sum())], fx["sentence_checksum"], rtol=1e-6)
    # (raw text features, |.| up to ~4: two fp32 evaluation orders of the same 12 blocks -- the reference's fused attention kernels vs the
    # oracle's explicit ones -- differ by a few 1e-6 RELATIVE; the rescaled-weight cases reach 2.4e-5 absolute on a 4.2 entry)
    assert np.abs(feats.detach().numpy() - fx["text_features"]).max() < 2e-5 * max(1.0, 0.5 * np.abs(fx["text_features"]).max())
    (feats * torch.from_numpy(fx["G"])).sum().backward()
    for key, leaf in (("grad_context", "context"), ("grad_rank", "rank")):
        ref = fx[key]
        assert np.abs(leaves[leaf].grad.numpy() - ref).max() < 2e-5 + 2e-4 * np.abs(ref).max(), key


@pytest.mark.parametrize("case", TC.TEXT_CASES, ids=[c[0] for c in TC.TEXT_CASES])
def test_tokenised_text_path(case):
    (name, tower, seed, lens) = case
    fx = TH.load(name)
    W = TC.make_tower_weights(tower, seed)
    c = TC.TOWERS[tower]
    ids = torch.from_numpy(fx["token_ids"])[:, :-1]                       # the last slot is the CLS placeholder
    pseudo = TO.generate_pseudo_tokens(ids)
    # a sentence that fills all 127 slots has no pad to find: the reference's generate_pseudo_tokens then yields all zeros
    # (prompt_encoder.py:257-265) and the CLS token only sees position 0 -- reproduced, not "fixed"
    assert [int((p > 0).sum()) for p in pseudo] == [n + 2 if n + 2 < 127 else 0 for n in lens]
    feats = TO.prompt_encoder_forward(W, c["heads"], W["token_embedding.weight"][ids], pseudo, c["layers"])
    assert np.abs(feats.numpy() - fx["text_features"]).max() < 2e-5 * max(1.0, 0.5 * np.abs(fx["text_features"]).max())


def test_rows_behind_the_sentence_do_not_reach_the_cls_token():
    """What the product's compaction relies on (causal mask + CLS pad mask, model/prompt_encoder.py:245-252,299-303): the
    pooled CLS feature depends on the n sentence positions AND on the first pad position n -- build_cls_mask pads its mask
    on the LEFT, so the CLS row sees columns {0} + {j + 1 : token j is not pad} = 0..n and not itself -- and on nothing
    behind that: garbage in slots n+1.. changes nothing, garbage in slot n does."""
    case = [c for c in TC.RANK_CASES if c[1] == "small"][0]
    feats, leaves = TH.oracle_rank_case(case)
    (name, tower, seed, K, base, position) = case
    W = TC.make_tower_weights(tower, seed)
    c = TC.TOWERS[tower]
    sent = leaves["sentence"].clone()
    n_real = int((leaves["pseudo"][0] > 0).sum())
    sent[:, n_real + 1:] = 37.0 * torch.randn(sent[:, n_real + 1:].shape, generator=torch.Generator().manual_seed(1))
    feats2 = TO.prompt_encoder_forward(W, c["heads"], sent, leaves["pseudo"], c["layers"])
    assert (feats2 - feats).abs().max().item() < 1e-6
    sent[:, n_real] += 0.5 * torch.randn(sent[:, n_real].shape, generator=torch.Generator().manual_seed(2))   # (a constant shift would be removed by the LayerNorms)
    feats3 = TO.prompt_encoder_forward(W, c["heads"], sent, leaves["pseudo"], c["layers"])
    assert (feats3 - feats).abs().max().item() > 1e-3


@pytest.mark.parametrize("name", ["rank_conch_k12_hot", "rank_mid_k6_hot"])
def test_hot_cases_reach_the_ranges_a_trained_tower_reaches(name):
    """Round 6: the rescaled-weight fixtures must really drive the softmax / GELU where the ordinary seeded towers
    never go -- attention logits |.| >= 25 on live (row, key) pairs, c_fc outputs |.| >= 6 -- and the ordinary case must not (so the
    new fixtures add coverage instead of repeating it)."""
    case = next(c for c in TC.RANK_CASES if c[0] == name)
    inp = TH.rank_case_inputs(case)
    _, leaves = TH.oracle_rank_case(case)
    stats = {}
    with torch.no_grad():
        TO.prompt_encoder_forward(inp["W"], inp["heads"], leaves["sentence"], leaves["pseudo"], inp["layers"], stats=stats)
    assert stats["attn_logit_absmax"] >= 25.0 and stats["gelu_input_absmax"] >= 6.0, stats
    base = next(c for c in TC.RANK_CASES if c[1] == case[1] and c[2] not in TC.WEIGHT_SCALES)
    binp = TH.rank_case_inputs(base)
    _, bl = TH.oracle_rank_case(base)
    bstats = {}
    with torch.no_grad():
        TO.prompt_encoder_forward(binp["W"], binp["heads"], bl["sentence"], bl["pseudo"], binp["layers"], stats=bstats)
    assert bstats["attn_logit_absmax"] < 0.5 * stats["attn_logit_absmax"] and bstats["gelu_input_absmax"] < stats["gelu_input_absmax"]
    print(name, stats, "ordinary:", bstats)
