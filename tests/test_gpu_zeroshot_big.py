"""Zero-shot pieces at full bag size: two-stage per-class top-k mean and the many-rows normalise vs torch."""
import pytest
import torch

import cases

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("N", [4096, 4097, 50_000, 200_000, 700_001])
@pytest.mark.parametrize("k", [1, 10, 32, 10 ** 9])
def test_topk_mean_two_stage(N, k):
    from vlsa_amd import functional as F
    dev = torch.device("cuda", 0)
    g = cases.gen(6000 + N % 1000 + k % 100)
    S = torch.randn(5, N, generator=g)
    S[1, : N // 3] = 0.25                       # ties
    S[2] = -S[2].abs()                          # all negative
    S[3, -7:] = 10.0                            # winners in the last chunk
    S[4, :3] = torch.tensor([9.0, 8.0, 7.0])    # winners in the first chunk
    kk = min(k, N)
    ref = S.double().topk(kk, dim=1).values.mean(dim=1)
    got = F.topk_mean(S.to(dev), k, out_scale=1.0).cpu().double()
    assert (got - ref).abs().max().item() < 2e-6 * max(1.0, ref.abs().max().item()) + (1e-5 if kk == N else 0.0)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
@pytest.mark.parametrize("N", [1, 5, 2798, 50_000])
def test_normalize_many(N, dtype):
    from vlsa_amd import functional as F
    dev = torch.device("cuda", 0)
    X = cases.make_bag(N, 6100 + N % 97).to(dtype)
    if N >= 5:
        X[3] = 0                                # zero row: the 1e-12 clamp
    ref = torch.nn.functional.normalize(X.float(), dim=-1)
    got = F.normalize_many(X.to(dev)).cpu()
    assert got.dtype == torch.float32 and got.shape == (N, 512)
    assert (got - ref).abs().max().item() < 2e-7 * 4


def test_zeroshot_full_size_matches_oracle_and_can_skip_features():
    from oracle import vlsa_oracle as O
    from vlsa_amd.vlsa import VLSA
    dev = torch.device("cuda", 0)
    K, N = 4, 20_000
    params = cases.make_params(8, K, 6200)
    X = cases.make_bag(N, 6201, "clustered")
    for pooling in ("logit_mean", "logit_max", "logit_top10"):
        net = VLSA.from_modules(dict(name="FeatMIL", dim_in=512, pooling=pooling), pretrained_text_features=params["T"].clone(),
                   logit_scale_init=cases.LOGIT_SCALE).to(dev).eval()
        with torch.no_grad():
            logits, feats, That = net(X[None].to(dev))
            ref_logits, ref_v, ref_T = O.vlsa_logits(X, params["T"], torch.tensor(cases.LOGIT_SCALE))
            _, ref = O.logit_pooling(ref_logits, pooling)
            assert (logits.cpu() - ref).abs().max().item() < 1e-4
            assert (feats.cpu() - ref_v).abs().max().item() < 1e-6 and (That.cpu() - ref_T).abs().max().item() < 1e-6
            net.return_patch_features = False
            logits2, none_feats, _ = net(X[None].to(dev))
            assert none_feats is None and torch.equal(logits2, logits)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
@pytest.mark.parametrize("N,D", [(1, 512), (3, 512), (2798, 512), (50_000, 512), (5000, 1024), (700, 264), (700, 504)])
def test_streaming_reductions_vs_torch(N, D, dtype):
    """colmax / rowdot / scored pooling: the 16-byte-load kernels (D % 8 == 0 is the library's contract; 264 and 504 have a
    partly filled last chunk)."""
    from vlsa_amd import functional as F
    dev = torch.device("cuda", 0)
    g = cases.gen(6300 + N % 89 + D)
    X = torch.randn(N, D, generator=g).to(dtype)
    Xd = X.to(dev)
    Xf = X.float()
    assert torch.equal(F.colmax(Xd).cpu(), Xf.max(dim=0).values)
    v = torch.randn(D, generator=g)
    rd = F.rowdot(Xd, v.to(dev)).cpu()
    assert (rd - Xf @ v).abs().max().item() < 2e-5 * max(1.0, (Xf @ v).abs().max().item())
    a = torch.randn(N, generator=g) * 3
    sp = F.scored_pool(Xd, a.to(dev)).cpu()
    ref = torch.softmax(a, dim=0) @ Xf
    assert (sp - ref).abs().max().item() < 1e-5 * max(1.0, ref.abs().max().item())
    mean = F.scored_pool(Xd, None).cpu()
    assert (mean - Xf.mean(dim=0)).abs().max().item() < 1e-5
