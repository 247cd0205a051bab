"""Fused attention-score kernel (vlsa_gated_scores) for the ABMIL-style pooling over N patches vs the CPU oracle
(oracle.gated_attention_pooling / attention_pooling on the bf16-rounded bag), and through the DeepMIL module; the module
on bf16 bags is additionally pinned to reference-generated fixtures in tests/test_gpu_modules_r2.py."""
import pytest
import torch

import cases

pytestmark = pytest.mark.gpu
TOL = 1e-4


def _weights(seed, gated, scale=1.0):
    g = cases.gen(seed)
    u = lambda *s, b: (torch.rand(*s, generator=g) * 2 - 1) * b  # noqa: E731
    Wa, ba = u(256, 512, b=scale / 512 ** 0.5), u(256, b=0.05)
    Wg, bg = (u(256, 512, b=scale / 512 ** 0.5), u(256, b=0.05)) if gated else (None, None)
    w2, c = u(1, 256, b=1 / 16), u(1, b=0.06)
    return Wa, ba, Wg, bg, w2, c


def _ref(X, Wa, ba, Wg, bg, w2, c):
    """raw scores from the CPU oracle's restatement of model/layers.py:103-122 / 137-153 (pinned to the reference by the
    deepmil_dm_* / deepmil_dmb_* fixtures), on the bf16-rounded values in fp32"""
    from oracle import vlsa_oracle as O
    Xf = X.float()
    if Wg is not None:
        return O.gated_attention_pooling(Xf, Wa, ba, Wg, bg, w2, c)[1]
    return O.attention_pooling(Xf, Wa, ba, w2, c)[1]


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
@pytest.mark.parametrize("gated", [True, False])
@pytest.mark.parametrize("N", [1, 16, 127, 128, 129, 1000, 5001, 16384, 16385, 20000, 32768, 32769, 50001, 65536, 65537, 100003])
def test_fused_scores_vs_torch(N, gated, dtype):
    from vlsa_amd import functional as F
    dev = torch.device("cuda", 0)
    X = cases.make_bag(N, 3000 + N, "clustered" if N % 2 else "iid").to(dtype)     # fp32: NOT bf16-representable values
    W = _weights(3100 + N, gated, scale=3.0)          # pre-activations of a few units: tanh / sigmoid well exercised
    ref = _ref(X, *W)
    fs = F.FusedAttnScores()
    Wd = [None if t is None else t.to(dev) for t in W]
    got = fs(X.to(dev), *Wd)
    torch.cuda.synchronize()
    assert got.shape == (N,)
    assert (got.cpu() - ref).abs().max().item() < TOL
    # strided rows (a view into a wider matrix) and a second call re-using the packed weights
    wide = torch.zeros(N, 640, dtype=dtype, device=dev)
    wide[:, :512] = X.to(dev)
    got2 = fs(wide[:, :512], *Wd)
    assert torch.equal(got2, got)
    # parameter update -> weights are re-packed
    Wd[0] = Wd[0] * 0.5
    W2 = list(W); W2[0] = W[0] * 0.5
    got3 = fs(X.to(dev), *Wd)
    assert (got3.cpu() - _ref(X, *W2)).abs().max().item() < TOL


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
def test_saturating_activations(dtype):
    from vlsa_amd import functional as F
    dev = torch.device("cuda", 0)
    X = (cases.make_bag(300, 3200) * 40).to(dtype)       # |pre-activation| up to ~100: tanh -> +-1, sigmoid -> 0 / 1
    W = _weights(3201, True, scale=3.0)
    got = F.FusedAttnScores()(X.to(dev), *[t.to(dev) for t in W])
    assert torch.isfinite(got).all()
    assert (got.cpu() - _ref(X, *W)).abs().max().item() < TOL


@pytest.mark.parametrize("pooling", ["attention", "gated_attention"])
def test_deepmil_bf16_bag_uses_fused_scores(pooling):
    from vlsa_amd.deepmil import DeepMIL
    dev = torch.device("cuda", 0)
    torch.manual_seed(11)
    m = DeepMIL(dim_in=512, dim_hid=256, use_feat_proj=False, pooling=pooling, pred_head="Adapter").to(dev).eval()
    X = cases.make_bag(3000, 3300, "clustered").to(torch.bfloat16).to(dev)
    with torch.no_grad():
        out_b, attn_b = m(X[None], ret_with_attn=True)               # bf16 bag: fused MFMA scores
        assert hasattr(m, "_fused_scores")
        out_f, attn_f = m(X.float()[None], ret_with_attn=True)       # same values as fp32: the fp32 variant of the fused kernel
    assert (out_b - out_f).abs().max().item() < TOL * max(1.0, out_f.abs().max().item())
    assert (attn_b - attn_f).abs().max().item() < TOL * max(1.0, attn_f.abs().max().item())


@pytest.mark.parametrize("pooling", ["gated_attention", "attention"])
@pytest.mark.parametrize("N", [16384, 20001, 70000])
def test_deepmil_large_bf16_bag_scores_and_pooling_in_one_launch(N, pooling):
    """A large bf16 bag through DeepMIL(gated_attention) in eval mode takes vlsa_gated_scores_pool_batch (scores + pooling partials in
    ONE launch of the persistent LDS-DMA kernel + the per-bag fold): same logits / attention weights as the fp32 bag's route (score
    kernel, then pooling kernel) and as the CPU oracle's pooling (model/layers.py:103-122)."""
    from vlsa_amd import functional as F
    from vlsa_amd.deepmil import DeepMIL
    from oracle import vlsa_oracle as O
    dev = torch.device("cuda", 0)
    torch.manual_seed(13)
    gated = pooling == "gated_attention"
    m = DeepMIL(dim_in=512, dim_hid=256, use_feat_proj=False, pooling=pooling, pred_head="Adapter").to(dev).eval()
    X = cases.make_bag(N, 3400 + N, "clustered").to(torch.bfloat16)
    Xd = X.to(dev)
    fs = F.FusedAttnScores()
    sg = m.sigma
    w = ((sg.fc1[0].weight, sg.fc1[0].bias, sg.score[0].weight, sg.score[0].bias, sg.fc2.weight, sg.fc2.bias) if gated else
         (sg.attention[0].weight, sg.attention[0].bias, None, None, sg.attention[2].weight, sg.attention[2].bias))
    if not F._score_big_tile(False, gated)[0] or F._NO_FUSED_POOL:
        pytest.skip("the persistent LDS-DMA kernel / its one-launch route is switched off (VLSA_GS_TILE=0 / VLSA_GS_NO_FUSED_POOL=1)")
    with torch.no_grad():
        got = fs.scores_and_pool(Xd, *w)
        assert got is not None                                     # the one-launch route applies
        pooled, a = got
        wc = [None if t is None else t.detach().cpu() for t in w]
        ref_pooled, ref_a = (O.gated_attention_pooling(X.float(), *wc) if gated else
                             O.attention_pooling(X.float(), wc[0], wc[1], wc[4], wc[5]))[:2]
        assert (a.cpu() - ref_a.reshape(-1)).abs().max().item() < TOL
        assert (pooled.cpu().reshape(-1) - ref_pooled.reshape(-1)).abs().max().item() < TOL
        two = F.scored_pool(Xd, fs(Xd, *w))                        # score kernel, then pooling kernel
        assert (pooled.reshape(-1) - two.reshape(-1)).abs().max().item() < 2e-5
        out_b, attn_b = m(Xd[None], ret_with_attn=True)
        out_f, attn_f = m(Xd.float()[None], ret_with_attn=True)
    assert (out_b - out_f).abs().max().item() < TOL * max(1.0, out_f.abs().max().item())
    assert (attn_b - attn_f).abs().max().item() < TOL * max(1.0, attn_f.abs().max().item())


@pytest.mark.parametrize("gated", [True, False])
@pytest.mark.parametrize("N", [1, 300, 2798, 20001])
def test_fp32_bag_scores_and_pooling_from_one_host_call(N, gated):
    """fp32 bags (the reference's own format): vlsa_gated_scores_pool chains the score kernel, the pooling partials and their merge inside
    the library -- same scores as the plain score launch (bit for bit), pooled row within 1e-4 of the oracle and of the two-call route."""
    from vlsa_amd import functional as F
    dev = torch.device("cuda", 0)
    X = cases.make_bag(N, 3500 + N, "clustered" if N % 2 else "iid")
    W = _weights(3600 + N, gated, scale=3.0)
    Wd = [None if t is None else t.to(dev) for t in W]
    fs = F.FusedAttnScores()
    if F._NO_FUSED_POOL:
        pytest.skip("VLSA_GS_NO_FUSED_POOL=1")
    Xd = X.to(dev)
    got = fs.scores_and_pool(Xd, *Wd)
    assert got is not None
    pooled, a = got
    assert torch.equal(a, fs(Xd, *Wd))
    ref_a = _ref(X, *W)
    assert (a.cpu() - ref_a).abs().max().item() < TOL
    want = torch.softmax(ref_a.double(), 0)[None] @ X.double()
    assert (pooled.cpu().double() - want).abs().max().item() < TOL
    assert (pooled.reshape(-1) - F.scored_pool(Xd, a).reshape(-1)).abs().max().item() < 2e-5


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
@pytest.mark.parametrize("gated", [True, False])
@pytest.mark.parametrize("N", [1, 257, 2799, 19001])
def test_scores_pooling_and_adapter_head_from_one_host_call(N, gated, dtype):
    """vlsa_gated_scores_pool_adapter: the Adapter head (model/deepmil.py:283-286) behind scores + pooling in the same host call --
    same scores and pooled row as the call without it, logit identical to the separate vlsa_adapter_head launch on that row and within
    1e-4 of keep * f + (1 - keep) * relu(W2 relu(W1 f)) in float64."""
    from vlsa_amd import functional as F
    if F._NO_FUSED_POOL:
        pytest.skip("VLSA_GS_NO_FUSED_POOL=1")
    dev = torch.device("cuda", 0)
    X = cases.make_bag(N, 7100 + N, "clustered" if N % 2 else "iid").to(dtype)
    W = _weights(7200 + N, gated, scale=3.0)
    Wd = [None if t is None else t.to(dev) for t in W]
    g = torch.Generator().manual_seed(7300 + N)
    W1 = torch.randn(128, 512, generator=g) / 512 ** 0.5
    W2 = torch.randn(512, 128, generator=g) / 128 ** 0.5
    fs = F.FusedAttnScores()
    Xd = X.to(dev)
    pooled0, a0 = fs.scores_and_pool(Xd, *Wd)
    pooled, a, logit = fs.scores_and_pool(Xd, *Wd, adapter=(W1.to(dev), W2.to(dev), 0.8))
    assert torch.equal(a, a0) and torch.equal(pooled, pooled0)
    assert logit.shape == (1, 512)
    assert torch.equal(logit.reshape(-1), F.adapter_head(pooled, W1.to(dev), W2.to(dev), 0.8))
    f = pooled.cpu().double().reshape(-1)
    want = 0.8 * f + 0.2 * torch.relu(W2.double() @ torch.relu(W1.double() @ f))
    assert (logit.cpu().double().reshape(-1) - want).abs().max().item() < TOL


@pytest.mark.parametrize("R", [4, 36, 64, 128, 192, 320, 512, 640])
def test_adapter_head_alone(R):
    """vlsa_adapter_head: keep f + (1 - keep) relu(W2 relu(W1 f)) (model/deepmil.py:283-286, model/layers.py:50-62) -- two launches of
    k_rows_dot_relu (one wave per output row), any R % 4 == 0."""
    from vlsa_amd import functional as F
    dev = torch.device("cuda", 0)
    g = torch.Generator().manual_seed(8100 + R)
    f = torch.randn(512, generator=g)
    W1 = torch.randn(R, 512, generator=g) / 512 ** 0.5
    W2 = torch.randn(512, R, generator=g) / R ** 0.5
    got = F.adapter_head(f.to(dev), W1.to(dev), W2.to(dev), 0.3).cpu().double()
    want = 0.3 * f.double() + 0.7 * torch.relu(W2.double() @ torch.relu(W1.double() @ f.double()))
    assert (got - want).abs().max().item() < 2e-5


@pytest.mark.parametrize("R", [36, 256])
def test_scores_pooling_and_adapter_head_other_widths_and_large_bags(R):
    """R != 128, and a bag beyond 65 536 rows (more than one round of tiles per walker)."""
    from vlsa_amd import functional as F
    if F._NO_FUSED_POOL:
        pytest.skip("VLSA_GS_NO_FUSED_POOL=1")
    dev = torch.device("cuda", 0)
    for N in (5000, 70001):
        X = cases.make_bag(N, 8200 + N, "iid").to(torch.bfloat16).to(dev)
        Wd = [None if t is None else t.to(dev) for t in _weights(8300 + N, True, scale=3.0)]
        g = torch.Generator().manual_seed(8400 + R)
        W1 = (torch.randn(R, 512, generator=g) / 512 ** 0.5).to(dev)
        W2 = (torch.randn(512, R, generator=g) / R ** 0.5).to(dev)
        fs = F.FusedAttnScores()
        pooled0, a0 = fs.scores_and_pool(X, *Wd)
        pooled, a, logit = fs.scores_and_pool(X, *Wd, adapter=(W1, W2, 0.5))
        assert torch.equal(a, a0)
        assert torch.equal(pooled, pooled0)
        f = pooled.double().reshape(-1)
        want = 0.5 * f + 0.5 * torch.relu(W2.double() @ torch.relu(W1.double() @ f))
        assert (logit.double().reshape(-1) - want).abs().max().item() < 2e-5


