"""Text side on the GPU: rank prompt learner + CoCa text tower in HIP vs the fixtures the REFERENCE
produced around a seeded random-weight tower (tests/golden/make_golden_text.py) and vs the CPU oracle: text features
[K, out] and the gradients w.r.t. the learnable context / rank embeddings within 1e-4; the tokenised-text path; the
compact-row evaluation against garbage in unreachable slots; VLSA end to end (text tower -> cached features -> bag logits)."""
import numpy as np
import pytest
import torch
import torch.nn as nn

import cases
import text_cases as TC
import text_helpers as TH
from test_text_modules_cpu import build_learner

pytestmark = pytest.mark.gpu
TOL = 1e-4


def build_encoder(tower, seed):
    from vlsa_amd.prompt_encoder import CONCHPromptEncoder
    c = TC.TOWERS[tower]
    enc = CONCHPromptEncoder(width=c["width"], heads=c["heads"], layers=c["layers"], vocab_size=c["vocab"], output_dim=c["out_dim"])
    enc.load_state_dict(TC.make_tower_weights(tower, seed))
    for p in enc.parameters():
        p.requires_grad_(False)
    return enc.cuda().eval()


@pytest.mark.parametrize("case", TC.RANK_CASES, ids=[c[0] for c in TC.RANK_CASES])
def test_rank_prompts_through_the_tower_forward_and_backward(case):
    (name, tower, seed, K, base, position) = case
    inp = TH.rank_case_inputs(case)
    fx = inp["fx"]
    enc = build_encoder(tower, seed)
    pl = build_learner(case, inp).cuda()
    with torch.no_grad():
        pl.context_embeds.copy_(torch.from_numpy(fx["context_embeds"]))
        pl.rank_embeds.copy_(torch.from_numpy(fx["rank_embeds"]))
        f0 = enc(prompts_embedding=pl(), prompts_pseudo_tokens=pl.pseudo_sentence_tokens)       # inference route (nothing saved)
    assert np.abs(f0.cpu().numpy() - fx["text_features"]).max() < TOL
    feats = enc(prompts_embedding=pl(), prompts_pseudo_tokens=pl.pseudo_sentence_tokens)         # training route
    assert np.abs(feats.detach().cpu().numpy() - fx["text_features"]).max() < TOL
    (feats * torch.from_numpy(fx["G"]).cuda()).sum().backward()
    for key, p in (("grad_context", pl.context_embeds), ("grad_rank", pl.rank_embeds)):
        ref = fx[key]
        err = np.abs(p.grad.cpu().numpy() - ref).max()
        assert err < TOL * max(1.0, np.abs(ref).max()), (key, err, np.abs(ref).max())
    # the same with the sentences' shared prefix (<sot> + the context tokens in front of the rank tokens) evaluated ONCE for all
    # prompts (compact_rows(prefix_len)): exact under the causal mask -- same features, same gradients
    pl.zero_grad()
    L = pl.shared_prefix_len
    with torch.no_grad():
        fs0 = enc(prompts_embedding=pl(), prompts_pseudo_tokens=pl.pseudo_sentence_tokens, shared_prefix_len=L)
    assert enc._plan(pl.pseudo_sentence_tokens, fs0.device, L).prefix_len == L and enc._plan(pl.pseudo_sentence_tokens, fs0.device, L).M < K * (
        int((pl.pseudo_sentence_tokens[0] > 0).sum()) + 2) or K == 1
    assert np.abs(fs0.cpu().numpy() - fx["text_features"]).max() < TOL
    fs = enc(prompts_embedding=pl(), prompts_pseudo_tokens=pl.pseudo_sentence_tokens, shared_prefix_len=L)
    assert np.abs(fs.detach().cpu().numpy() - fx["text_features"]).max() < TOL
    (fs * torch.from_numpy(fx["G"]).cuda()).sum().backward()
    for key, p in (("grad_context", pl.context_embeds), ("grad_rank", pl.rank_embeds)):
        ref = fx[key]
        err = np.abs(p.grad.cpu().numpy() - ref).max()
        assert err < TOL * max(1.0, np.abs(ref).max()), ("shared prefix", key, err, np.abs(ref).max())
    pl.zero_grad()
    # a second forward / backward on the same plan (workspaces are per call) gives the same numbers
    pl.zero_grad()
    feats2 = enc(prompts_embedding=pl(), prompts_pseudo_tokens=pl.pseudo_sentence_tokens)
    assert torch.equal(feats2.detach(), feats.detach())
    (feats2 * torch.from_numpy(fx["G"]).cuda()).sum().backward()
    assert np.abs(pl.rank_embeds.grad.cpu().numpy() - fx["grad_rank"]).max() < TOL * max(1.0, np.abs(fx["grad_rank"]).max())


def test_shared_prefix_of_many_prompts_takes_the_ticketed_fold():
    """More than 128 compact rows with a shared prefix: the attention backward cannot hold all query rows in one workgroup's LDS and folds
    the prompts' shares of the prefix keys' dK / dV through the per-head ticket (the route every shared-prefix case took before round 6).
    30 rank prompts on the CONCH-size tower: features and gradients with the prefix evaluated once = the row-per-position route."""
    base_case = next(c for c in TC.RANK_CASES if c[0] == "rank_conch_k4")
    case = (base_case[0], base_case[1], base_case[2], 30, base_case[4], base_case[5])
    inp = TH.rank_case_inputs(base_case)
    enc = build_encoder(case[1], case[2])
    pl = build_learner(case, inp).cuda()
    L = pl.shared_prefix_len
    tok = pl.pseudo_sentence_tokens
    assert L > 0 and enc._plan(tok, torch.device("cuda", 0), L).M > 128
    G = torch.randn(30, enc.output_dim, generator=torch.Generator().manual_seed(77)).cuda()
    out = {}
    for name, pref in (("rows", 0), ("prefix", L)):
        pl.zero_grad(set_to_none=True)
        f = enc(prompts_embedding=pl(), prompts_pseudo_tokens=tok, shared_prefix_len=pref)
        (f * G).sum().backward()
        out[name] = (f.detach().clone(), pl.context_embeds.grad.clone(), pl.rank_embeds.grad.clone())
    for a, b, what in zip(out["rows"], out["prefix"], ("features", "d context", "d rank")):
        err, scale = (a - b).abs().max().item(), max(1.0, a.abs().max().item())
        assert err < TOL * scale, (what, err, scale)


@pytest.mark.parametrize("case", TC.TEXT_CASES, ids=[c[0] for c in TC.TEXT_CASES])
def test_tokenised_text_path(case):
    (name, tower, seed, lens) = case
    fx = TH.load(name)
    enc = build_encoder(tower, seed)
    with torch.no_grad():
        feats = enc(prompts_text=torch.from_numpy(fx["token_ids"]).cuda())
    assert np.abs(feats.cpu().numpy() - fx["text_features"]).max() < TOL


def test_unreachable_slots_are_ignored_and_reachable_ones_are_not():
    case = [c for c in TC.RANK_CASES if c[1] == "mid"][0]
    (name, tower, seed, K, base, position) = case
    inp = TH.rank_case_inputs(case)
    enc = build_encoder(tower, seed)
    pl = build_learner(case, inp).cuda()
    with torch.no_grad():
        sent = pl()
        n = int((pl.pseudo_sentence_tokens[0] > 0).sum())
        f0 = enc(prompts_embedding=sent, prompts_pseudo_tokens=pl.pseudo_sentence_tokens)
        junk = sent.clone()
        junk[:, n + 1:] = float("nan")                           # never read: not even NaNs get through
        f1 = enc(prompts_embedding=junk, prompts_pseudo_tokens=pl.pseudo_sentence_tokens)
        assert torch.equal(f0, f1)
        junk = sent.clone()
        junk[:, n] += 0.3 * torch.randn_like(junk[:, n])         # the first pad slot IS seen by the CLS token (shifted mask)
        f2 = enc(prompts_embedding=junk, prompts_pseudo_tokens=pl.pseudo_sentence_tokens)
        assert (f2 - f0).abs().max().item() > 1e-3


def test_cpu_inputs_fail_loudly():
    from vlsa_amd._native import VlsaNativeError
    enc = build_encoder("small", 1)
    x = torch.zeros(2, 127, 128, device="cuda", requires_grad=True)
    pt = torch.ones(2, 127, dtype=torch.long, device="cuda")
    with torch.no_grad():
        enc(prompts_embedding=x, prompts_pseudo_tokens=pt)
    with pytest.raises(VlsaNativeError):
        enc(prompts_embedding=x.detach().cpu(), prompts_pseudo_tokens=pt)


@pytest.mark.parametrize("case", [c for c in TC.RANK_CASES if c[0] in ("rank_conch_k4", "rank_small_k8_front", "rank_mid_k5_middle")],
                         ids=lambda c: c[0])
@pytest.mark.parametrize("prefix", [False, True], ids=["rows", "shared_prefix"])
def test_trainable_tower_features_and_weight_gradients(case, prefix):
    """``vlsa_txt_encoder_frozen: False`` (runner/vlsa_handler.py:131): tower parameters that require grad get their gradients from
    the HIP weight-gradient products (``vlsa_tt_backward_train``; a torch route until round 4).  Text features, d prompts (context /
    rank embeds) and the gradients of EVERY tower parameter vs the CPU oracle's autograd over the full 128 positions
    (model/prompt_encoder.py:267-322 restated in oracle/text_oracle.py)."""
    from oracle import text_oracle as TO
    (name, tower, seed, K, base, position) = case
    inp = TH.rank_case_inputs(case)
    fx = inp["fx"]
    enc = build_encoder(tower, seed)
    for p in enc.parameters():
        p.requires_grad_(True)
    enc.token_embedding.weight.requires_grad_(False)                 # (the learner reads it at construction only)
    pl = build_learner(case, inp).cuda()
    with torch.no_grad():
        pl.context_embeds.copy_(torch.from_numpy(fx["context_embeds"]))
        pl.rank_embeds.copy_(torch.from_numpy(fx["rank_embeds"]))
    L = pl.shared_prefix_len if prefix else 0
    feats = enc(prompts_embedding=pl(), prompts_pseudo_tokens=pl.pseudo_sentence_tokens, shared_prefix_len=L)
    assert type(feats.grad_fn).__name__ == "_TextTowerTrainFnBackward"              # the native route, not torch ops
    assert np.abs(feats.detach().cpu().numpy() - fx["text_features"]).max() < TOL
    G = torch.from_numpy(fx["G"])
    (feats * G.cuda()).sum().backward()
    # the oracle with every weight as a leaf
    W = {k: v.clone().requires_grad_(k != "token_embedding.weight") for k, v in inp["W"].items()}
    ctx = torch.from_numpy(fx["context_embeds"]).clone().requires_grad_(True)
    rk = torch.from_numpy(fx["rank_embeds"]).clone().requires_grad_(True)
    E = inp["W"]["token_embedding.weight"]
    bos, eos, pad = inp["special"]
    pseudo = TO.pseudo_sentence_tokens(K, ctx.shape[0], rk.shape[1])
    template = TO.sentence_template(E[pad], E[bos], E[eos], E[inp["table"]["X."][1]], pseudo)
    sent = TO.rank_prompt_learner_forward(ctx, rk, template, TO.interpolation_weights(base, K), K, position)
    ref = TO.prompt_encoder_forward(W, inp["heads"], sent, pseudo, inp["layers"])
    (ref * G).sum().backward()

    def close(got, want, what):
        got, want = got.detach().cpu().numpy(), want.detach().numpy()
        cases.record_grad_error("trainable tower: " + what, np.abs(got - want).max(), np.abs(want).max(), 1e-4 * np.abs(want).max() + 1e-7)
        assert np.abs(got - want).max() <= 1e-4 * np.abs(want).max() + 1e-7, (what, np.abs(got - want).max(), np.abs(want).max())
    close(pl.context_embeds.grad, ctx.grad, "context")
    close(pl.rank_embeds.grad, rk.grad, "rank")
    sd = dict(enc.named_parameters())
    checked = 0
    for k, w in W.items():
        if k == "token_embedding.weight":
            continue
        g = sd[k].grad
        assert g is not None, k
        if k == "positional_embedding":
            # positions no compact row uses get no gradient here and an exactly-zero one in the reference ... except position
            # rows that only feed masked-out attention: compare where the reference is non-zero, and require zeros elsewhere
            assert float(g.abs().max()) > 0
        close(g, w.grad, k)
        checked += 1
    assert checked == 5 + 12 * inp["layers"]


def test_trainable_tower_two_sgd_steps_and_partial_freezing():
    """the packed weight copies follow the optimizer (re-packed when a weight's version changes): two SGD steps on every tower
    parameter vs the same two steps through the CPU oracle; then only ``text_projection`` / ``ln_final`` trainable: the same
    gradients for those, none for the rest."""
    from oracle import text_oracle as TO
    case = [c for c in TC.RANK_CASES if c[0] == "rank_small_k8_front"][0]
    (name, tower, seed, K, base, position) = case
    inp = TH.rank_case_inputs(case)
    fx = inp["fx"]
    enc = build_encoder(tower, seed)
    names = [k for k, _ in enc.named_parameters() if k != "token_embedding.weight"]
    for k, p in enc.named_parameters():
        p.requires_grad_(k != "token_embedding.weight")
    pl = build_learner(case, inp).cuda()
    with torch.no_grad():
        pl.context_embeds.copy_(torch.from_numpy(fx["context_embeds"]))
        pl.rank_embeds.copy_(torch.from_numpy(fx["rank_embeds"]))
    G = torch.from_numpy(fx["G"])
    W = {k: v.clone().requires_grad_(k != "token_embedding.weight") for k, v in inp["W"].items()}
    E = inp["W"]["token_embedding.weight"]
    bos, eos, pad = inp["special"]
    ctx, rk = torch.from_numpy(fx["context_embeds"]), torch.from_numpy(fx["rank_embeds"])
    pseudo = TO.pseudo_sentence_tokens(K, ctx.shape[0], rk.shape[1])
    template = TO.sentence_template(E[pad], E[bos], E[eos], E[inp["table"]["X."][1]], pseudo)
    sent = TO.rank_prompt_learner_forward(ctx, rk, template, TO.interpolation_weights(base, K), K, position)
    sd = dict(enc.named_parameters())
    opt = torch.optim.SGD([sd[k] for k in names], lr=0.05)
    opt_ref = torch.optim.SGD([W[k] for k in names], lr=0.05)
    for step in range(3):
        feats = enc(prompts_embedding=pl().detach(), prompts_pseudo_tokens=pl.pseudo_sentence_tokens, shared_prefix_len=pl.shared_prefix_len)
        ref = TO.prompt_encoder_forward(W, inp["heads"], sent, pseudo, inp["layers"])
        assert (feats.detach().cpu() - ref.detach()).abs().max().item() < TOL, step
        if step == 2:
            break
        opt.zero_grad(); opt_ref.zero_grad()
        (feats * G.cuda()).sum().backward()
        (ref * G).sum().backward()
        opt.step(); opt_ref.step()
    full = {k: sd[k].grad.clone() for k in names}
    # partial freezing: the frozen parameters get no gradient, the others the same one
    for k in names:
        sd[k].grad = None
        sd[k].requires_grad_(k in ("text_projection", "ln_final.weight", "ln_final.bias"))
    with torch.no_grad():
        for k in names:
            sd[k].copy_(inp["W"][k])
    for k, v in W.items():
        v.grad = None
        v.data.copy_(inp["W"][k])
    feats = enc(prompts_embedding=pl().detach(), prompts_pseudo_tokens=pl.pseudo_sentence_tokens)
    (feats * G.cuda()).sum().backward()
    ref = TO.prompt_encoder_forward(W, inp["heads"], sent, pseudo, inp["layers"])
    (ref * G).sum().backward()
    for k in names:
        if sd[k].requires_grad:
            assert (sd[k].grad.cpu() - W[k].grad).abs().max().item() <= 1e-4 * W[k].grad.abs().max().item() + 1e-7, k
        else:
            assert sd[k].grad is None, k
    assert len(full) == 5 + 12 * inp["layers"]


def test_vlsa_end_to_end_with_gpu_text_side():
    """VLSA(prompt_learner=RankPromptLearner, prompt_encoder=CONCHPromptEncoder): rank prompts -> text tower -> cached text
    features -> bag logits, forward and one backward, vs the CPU oracles (model/vlsa.py:149-156,181-198)."""
    from oracle import text_oracle as TO, vlsa_oracle as O
    from vlsa_amd.prompt_adapter import PromptAdapter
    from vlsa_amd.vlsa import VLSA
    case = [c for c in TC.RANK_CASES if c[0] == "rank_conch_k4"][0]
    (name, tower, seed, K, base, position) = case
    inp = TH.rank_case_inputs(case)
    enc = build_encoder(tower, seed)
    pl = build_learner(case, inp)
    P = 12
    params = cases.make_params(P, K, 4242)
    qnet = PromptAdapter(method="TaskRes", num_prompts=P, pretrained_prompt_features=params["prompt"], res_ratio=0.5)
    cfg = dict(name="VLFAN", dim_in=512, dim_hid=256, use_feat_proj=False, num_query=P, query="Text", gated_query=False,
               query_pooling="mean", pred_head="default")
    model = VLSA.from_modules(cfg, prompt_learner=pl, prompt_encoder=enc, query_network=qnet, logit_scale_init=cases.LOGIT_SCALE)
    with torch.no_grad():
        qnet.residual_features.copy_(params["resid"])
        model.mil_encoder.visual_adapter.weight.copy_(params["W"]); model.mil_encoder.visual_adapter.bias.copy_(params["b"])
    model = model.cuda()
    bags = [cases.make_bag(n, 4250 + i) for i, n in enumerate((700, 1300, 64))]
    # CPU oracle: text features, then the bag path
    feats_ref, leaves = TH.oracle_rank_case(case, requires_grad=True)
    with torch.no_grad():   # the oracle case uses the fixture's (perturbed) embeddings: give the model the same
        pl.context_embeds.copy_(leaves["context"].detach()); pl.rank_embeds.copy_(leaves["rank"].detach())
    Q = 0.5 * params["resid"] + params["prompt"]
    ref_logits = torch.cat([O.vlsa_vlfan_forward(x, Q, feats_ref, torch.tensor(cases.LOGIT_SCALE), head_weight=params["W"],
                                                 head_bias=params["b"])["logits"] for x in bags])
    model.eval()
    with torch.no_grad():
        logits, _, That = model.forward_bags([x.cuda() for x in bags])
        assert model.forward_text_only() is model.forward_text_only()          # cached: the tower ran once
    assert (logits.cpu() - ref_logits.detach()).abs().max().item() < TOL
    model.train()
    logits2, _, _ = model.forward_bags([x.cuda() for x in bags])
    G = torch.randn(logits2.shape, generator=torch.Generator().manual_seed(5))
    (logits2 * G.cuda()).sum().backward()
    (ref_logits * G).sum().backward()
    for p, leaf in ((pl.context_embeds, leaves["context"]), (pl.rank_embeds, leaves["rank"])):
        ref = leaf.grad
        cases.record_grad_error("end to end: prompt embeds", (p.grad.cpu() - ref).abs().max().item(), ref.abs().max().item())
        assert (p.grad.cpu() - ref).abs().max().item() < 2e-3 * max(1e-3, ref.abs().max().item()) + 1e-5
    # after an optimizer-like update the cached text features are recomputed
    with torch.no_grad():
        t0 = model.forward_text_only().clone()
        pl.rank_embeds.add_(0.01)
        assert not torch.equal(model.forward_text_only(), t0)


@pytest.mark.parametrize("kind", ["rank_tail", "rank_front", "rank_middle", "rank_own_context", "plain_ragged"])
def test_sentence_assembly_kernels_match_the_torch_ops(kind):
    """vlsa_prompt_sentences / _backward (one launch each way) vs the learner's torch-op assembly on the same device -- which
    tests/test_text_modules_cpu.py pins to the reference's sentences (fixtures) on the CPU."""
    from vlsa_amd.prompt_learner import PlainPromptLearner, RankPromptLearner
    torch.manual_seed(5)
    vocab, dim = 500, 128
    E = torch.nn.Embedding(vocab, dim).cuda()
    table, ctx_key, names = TC.synthetic_prompt_table(vocab, 3, n_ctx=6, rank_lens=(4, 2, 3, 4) if kind == "plain_ragged" else (3, 3, 3, 3))
    tok = TC.ReplayTokenizer(table)
    cfg = dict(max_num_tokens=127, embedding_dim=dim, embedding_dtype=torch.float32)
    if kind == "plain_ragged":
        pl = PlainPromptLearner(cfg, tok, E, num_ranks=4, num_tokens_per_rank=[4, 2, 3, 4], num_context_tokens=6,
                                init_context=ctx_key, init_rank_names=names)
    else:
        pl = RankPromptLearner(cfg, tok, E, num_base_ranks=4, num_ranks=7, num_tokens_per_rank=3, num_context_tokens=6,
                               rank_tokens_position={"rank_front": "front", "rank_middle": "middle"}.get(kind, "tail"),
                               rank_specific_context=kind == "rank_own_context", init_context=ctx_key, init_rank_names=names)
    with torch.no_grad():
        pl.context_embeds.add_(torch.randn_like(pl.context_embeds) * 0.1)
        pl.rank_embeds.add_(torch.randn_like(pl.rank_embeds) * 0.1)
    G = torch.randn_like(pl.sentence_embeds)
    out = pl()
    (out * G).sum().backward()
    g_hip = (pl.context_embeds.grad.clone(), pl.rank_embeds.grad.clone())
    pl.zero_grad(set_to_none=True)
    pl._torch_ops_only = True
    ref = pl()
    (ref * G).sum().backward()
    assert out.shape == ref.shape and (out - ref).abs().max().item() < 1e-6
    assert (g_hip[0] - pl.context_embeds.grad).abs().max().item() < 1e-5 * max(1.0, pl.context_embeds.grad.abs().max().item())
    assert (g_hip[1] - pl.rank_embeds.grad).abs().max().item() < 1e-5 * max(1.0, pl.rank_embeds.grad.abs().max().item())


# ---- the persistent forward (k_tt_forward_persistent): one launch for the 12 blocks, stages ordered by in-kernel counters --------
def _tower_kernels(fn):
    """names of the HIP kernels `fn` launches (torch profiler, device activity only)"""
    from torch.profiler import ProfilerActivity, profile
    with profile(activities=[ProfilerActivity.CUDA]) as prof:
        fn()
        torch.cuda.synchronize()
    return [e.key for e in prof.key_averages() for _ in range(e.count)]


@pytest.mark.parametrize("name", ["rank_conch_k12", "rank_conch_k4", "text_conch"])
def test_persistent_forward_equals_the_launch_per_stage_path(name, monkeypatch):
    """CONCH-size tower, <= 112 compact rows, VLSA_TT_PERSIST=1: the 12 blocks as ONE persistent launch (opt-in: measured slower than
    the default, profiles/r04_bench_text_persist.txt).  Same features as the launch-per-stage path up to the summation order of the K split (8 instead of 4 waves), same fixtures, with
    and without saved activations (the backward pass reads what the persistent launch wrote), repeated calls bit-identical, and
    no in-kernel wait timed out."""
    if name.startswith("rank"):
        case = next(c for c in TC.RANK_CASES if c[0] == name)
        inp = TH.rank_case_inputs(case)
        fx = inp["fx"]
        enc = build_encoder(case[1], case[2])
        pl = build_learner(case, inp).cuda()
        with torch.no_grad():
            pl.context_embeds.copy_(torch.from_numpy(fx["context_embeds"]))
            pl.rank_embeds.copy_(torch.from_numpy(fx["rank_embeds"]))
        L = pl.shared_prefix_len
        kw = dict(prompts_pseudo_tokens=pl.pseudo_sentence_tokens, shared_prefix_len=L)
        run = lambda: enc(prompts_embedding=pl(), **kw)                                     # noqa: E731
        plan = enc._plan(pl.pseudo_sentence_tokens, torch.device("cuda", 0), L)
        grads = lambda: (pl.context_embeds.grad.clone(), pl.rank_embeds.grad.clone())      # noqa: E731
        want = fx["text_features"]
    else:
        case = next(c for c in TC.TEXT_CASES if c[0] == name)
        fx = TH.load(name)
        enc = build_encoder(case[1], case[2])
        ids = torch.from_numpy(fx["token_ids"]).cuda()
        run = lambda: enc(prompts_text=ids)                                                 # noqa: E731      (68 rows, prompts straddle row tiles)
        plan, grads, want = None, None, fx["text_features"]
    assert plan is None or plan.M <= 112
    monkeypatch.setenv("VLSA_TT_PERSIST", "1")
    with torch.no_grad():
        a = run()
        names = _tower_kernels(run)
        a2 = run()
    # (the blocks' own launches are gone: their attention kernel and their LayerNorm-prologue products; the text projection behind the
    #  blocks is a k_tt_gemm<1, ...> launch on both paths)
    assert any("k_tt_forward_persistent" in n for n in names) and not any("k_tt_attn_fwd" in n or "k_tt_gemm<1, 4, 1" in n for n in names), names
    assert torch.equal(a, a2)
    monkeypatch.setenv("VLSA_TT_PERSIST", "0")
    with torch.no_grad():
        b = run()
        names = _tower_kernels(run)
    assert not any("k_tt_forward_persistent" in n for n in names)
    assert np.abs(a.cpu().numpy() - want).max() < TOL and np.abs(b.cpu().numpy() - want).max() < TOL
    assert (a - b).abs().max().item() < 2e-5
    if grads is not None:      # training route: persistent forward with saved activations -> the launch-per-stage backward
        G = torch.from_numpy(fx["G"]).cuda()
        got = {}
        for mode in ("1", "0"):
            monkeypatch.setenv("VLSA_TT_PERSIST", mode)
            pl.zero_grad(set_to_none=True)
            f = run()
            (f * G).sum().backward()
            got[mode] = (f.detach().clone(),) + grads()
        for x, y in zip(got["1"], got["0"]):
            assert (x - y).abs().max().item() < 2e-5 * max(1.0, y.abs().max().item())
        for key, g in (("grad_context", got["1"][1]), ("grad_rank", got["1"][2])):
            assert np.abs(g.cpu().numpy() - fx[key]).max() < TOL * max(1.0, np.abs(fx[key]).max()), key
    assert len(enc._plans) >= 1
    for pln in enc._plans.values():
        pln.check_status(wait=True)
