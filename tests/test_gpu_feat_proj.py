"""Fused Feat_Projecter kernel (vlsa_feat_project: Linear(512, 512) + LayerNorm over all N patch rows in one launch) vs the
CPU oracle's restatement of model/layers.py:65-82 (pinned to the reference by the featproj_* fixtures, which
tests/test_gpu_modules_r2.py replays through the modules -- and thereby through this kernel in no-grad mode)."""
import pytest
import torch

import cases

pytestmark = pytest.mark.gpu
TOL = 1e-4


def _params(seed, scale=1.0):
    g = cases.gen(seed)
    u = lambda *s, b: (torch.rand(*s, generator=g) * 2 - 1) * b  # noqa: E731
    return u(512, 512, b=scale / 512 ** 0.5), u(512, b=0.1), 1.0 + u(512, b=0.3), u(512, b=0.2)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
@pytest.mark.parametrize("N", [1, 16, 63, 64, 65, 127, 128, 129, 1000, 5001, 20000])
def test_fused_feat_projecter_vs_oracle(N, dtype):
    from oracle import vlsa_oracle as O
    from vlsa_amd import functional as F
    dev = torch.device("cuda", 0)
    X = cases.make_bag(N, 5000 + N, "clustered" if N % 2 else "iid").to(dtype)     # fp32: NOT bf16-representable values
    W, b, gm, bt = _params(5100 + N, scale=2.0)
    ref = O.feat_projecter_forward(X.float(), W, b, gm, bt)
    fp = F.FusedFeatProjecter()
    Wd = [t.to(dev) for t in (W, b, gm, bt)]
    got = fp(X.to(dev), *Wd, 1e-5)
    torch.cuda.synchronize()
    assert got.shape == (N, 512) and got.dtype == torch.float32
    assert (got.cpu() - ref).abs().max().item() < TOL
    # strided rows (a view into a wider matrix) and a second call re-using the packed weights
    wide = torch.zeros(N, 640, dtype=dtype, device=dev)
    wide[:, :512] = X.to(dev)
    assert torch.equal(fp(wide[:, :512], *Wd, 1e-5), got)
    # parameter update -> weights are re-packed
    Wd[0] = Wd[0] * 0.5
    got2 = fp(X.to(dev), *Wd, 1e-5)
    assert (got2.cpu() - O.feat_projecter_forward(X.float(), W * 0.5, b, gm, bt)).abs().max().item() < TOL


def test_constant_and_large_rows():
    """rows with (almost) no variance after the projection exercise the eps inside the root; |x| >> 1 rows the range"""
    from oracle import vlsa_oracle as O
    from vlsa_amd import functional as F
    dev = torch.device("cuda", 0)
    X = cases.make_bag(300, 5200)
    X[7] = 0.0
    X[8] = X[8] * 1e3
    X[9] = X[9] * 1e-6
    W, b, gm, bt = _params(5201)
    b = torch.zeros_like(b)                      # row 7 projects to exactly 0: variance 0, output = beta
    ref = O.feat_projecter_forward(X, W, b, gm, bt)
    got = F.FusedFeatProjecter()(X.to(dev), W.to(dev), b.to(dev), gm.to(dev), bt.to(dev), 1e-5)
    assert torch.isfinite(got).all()
    assert (got.cpu() - ref).abs().max().item() < 2e-4          # row 9: rstd ~ 300 amplifies the 2^-17 operand split
    assert (got[7].cpu() - bt).abs().max().item() < 1e-6


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
def test_module_routes_inference_through_the_kernel(dtype):
    from vlsa_amd.layers import Feat_Projecter
    dev = torch.device("cuda", 0)
    torch.manual_seed(3)
    m = Feat_Projecter(512, 512).to(dev)
    with torch.no_grad():
        m.projecter[1].weight.uniform_(0.5, 1.5)
        m.projecter[1].bias.uniform_(-0.2, 0.2)
    X = cases.make_bag(777, 5300, "clustered").to(dtype).to(dev)
    y_train = m(X[None])                                           # parameters require grad: the fused kernel under autograd
    assert y_train.requires_grad and hasattr(m, "_fused")          # (vlsa_feat_project_train + vlsa_feat_project_backward)
    with torch.no_grad():
        y_eval = m(X[None])                                        # no grad: the same kernel without the statistics output
    assert y_eval.shape == (1, 777, 512) and y_eval.dtype == torch.float32
    assert (y_eval - y_train.detach().float()).abs().max().item() == 0
    y_torch = m.projecter(X.float())                               # the two torch modules on the same parameters
    assert (y_eval[0] - y_torch.detach()).abs().max().item() < TOL
    G = torch.randn(777, 512, device=dev)
    gw = torch.autograd.grad((y_torch * G).sum(), m.projecter[0].weight)[0]
    (y_train[0] * G).sum().backward()
    cases.record_grad_error("projecter weight", (m.projecter[0].weight.grad - gw).abs().max().item(), gw.abs().max().item())
    assert (m.projecter[0].weight.grad - gw).abs().max().item() < 2e-3 * gw.abs().max().item()
