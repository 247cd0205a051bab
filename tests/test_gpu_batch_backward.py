"""Batched aggregation backward (B bags sharing the queries, one persistent launch) vs torch.autograd through the CPU
oracle and vs the per-bag HIP backward; module-level: VLSA.forward_bags with gradients vs the per-bag loop the
reference's training step runs (runner/vlsa_handler.py:260-289)."""
import pytest
import torch

import cases
from oracle import vlsa_oracle as O

pytestmark = pytest.mark.gpu

GRAD_RTOL = 1e-4  # relative to the largest gradient entry, as in test_gpu_modules.py (observed <= 2.6e-5: profiles/r04_grad_errors.txt)


def _oracle_grads(bags, Q, Gs, gated):
    Q = Q.clone().requires_grad_(True)
    outs = []
    total = 0.0
    for X, G in zip(bags, Gs):
        out = O.vlfan_forward(X.float(), Q, gated_query=gated)["out"]
        outs.append(out.detach())
        total = total + (out * G).sum()
    total.backward()
    return torch.stack(outs), Q.grad


@pytest.mark.parametrize("dtype,P,gated", [(torch.bfloat16, 12, False), (torch.bfloat16, 7, True), (torch.bfloat16, 1, False),
                                           (torch.float32, 12, False), (torch.bfloat16, 16, False), (torch.float32, 16, True),
                                           (torch.float32, 5, False)])
@pytest.mark.parametrize("sizes", [[3000, 1, 33, 700], [64, 65, 31, 32, 4100, 17, 200, 1000, 5]])
def test_batched_backward_vs_oracle_autograd(sizes, dtype, P, gated):
    from vlsa_amd import functional as F
    dev = torch.device("cuda", 0)
    bags = [cases.make_bag(n, 900 + i).to(dtype) for i, n in enumerate(sizes)]
    params = cases.make_params(P, 4, 910, gated=gated)
    Q = 0.5 * params["resid"] + params["prompt"]
    g = cases.gen(911)
    Gs = [torch.randn(P, 512, generator=g) for _ in sizes]
    ref_out, ref_grad = _oracle_grads(bags, Q, Gs, gated)
    Qd = Q.to(dev).requires_grad_(True)
    out = F.vlfan_cross_attention_bags([x.to(dev) for x in bags], Qd, gated=gated)
    (out * torch.stack(Gs).to(dev)).sum().backward()
    torch.cuda.synchronize()
    assert (out.detach().cpu() - ref_out).abs().max().item() < 1e-4 * max(1.0, ref_out.abs().max().item())
    scale = ref_grad.abs().max().item()
    cases.record_grad_error("dQ", (Qd.grad.cpu() - ref_grad).abs().max().item(), scale, GRAD_RTOL * scale)
    assert (Qd.grad.cpu() - ref_grad).abs().max().item() < GRAD_RTOL * scale


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
def test_batched_backward_equals_per_bag_backward_full_size(dtype):
    """32 bags of 2k..50k bf16 / fp32 rows: the persistent batch backward vs the sum of the per-bag HIP backward."""
    from vlsa_amd import functional as F
    dev = torch.device("cuda", 0)
    sizes = [50_000, 10_000, 2798, 20_001] + [2000 + 977 * i for i in range(28)]
    base = cases.make_bag(50_000, 920).to(dtype).to(dev)
    bags = [base[:n] if i % 2 == 0 else base[50_000 - n:] for i, n in enumerate(sizes)]
    params = cases.make_params(12, 4, 921)
    Q = (0.5 * params["resid"] + params["prompt"]).to(dev)
    G = torch.randn(len(sizes), 12, 512, generator=cases.gen(922)).to(dev)
    Qa = Q.clone().requires_grad_(True)
    out = F.vlfan_cross_attention_bags(bags, Qa)
    (out * G).sum().backward()
    Qb = Q.clone().requires_grad_(True)
    total = 0.0
    for i, x in enumerate(bags):
        o, _ = F.vlfan_cross_attention(x, Qb)
        assert (o - out[i]).abs().max().item() < 2e-5 * max(1.0, o.abs().max().item())
        total = total + (o * G[i]).sum()
    total.backward()
    torch.cuda.synchronize()
    scale = Qb.grad.abs().max().item()
    assert (Qa.grad - Qb.grad).abs().max().item() < 2e-4 * scale


@pytest.mark.parametrize("pooling", ["mean", "weight", "attention"])
def test_forward_bags_training_matches_per_bag_loop(pooling):
    from vlsa_amd.vlsa import VLSA
    dev = torch.device("cuda", 0)
    P, K = 12, 4
    params = cases.make_params(P, K, 930)
    sizes = [1500, 40, 2798, 333, 64, 1000]
    bags = [cases.make_bag(n, 940 + i).to(torch.bfloat16).to(dev)[None] for i, n in enumerate(sizes)]

    def build():
        torch.manual_seed(5)
        cfg = dict(name="VLFAN", dim_in=512, dim_hid=64, use_feat_proj=False, query="Parameter", num_query=P,
                   gated_query=False, query_pooling=pooling, pred_head="default")
        m = VLSA.from_modules(cfg, pretrained_text_features=params["T"].clone()).to(dev)
        with torch.no_grad():
            m.mil_encoder.Q.copy_((0.5 * params["resid"] + params["prompt"]).to(dev))
            m.mil_encoder.visual_adapter.weight.copy_(params["W"].to(dev))
            m.mil_encoder.visual_adapter.bias.copy_(params["b"].to(dev))
        return m.train()

    w = torch.randn(len(sizes), K, generator=cases.gen(931)).to(dev)
    a, b = build(), build()
    logits_a, feat_a, _ = a.forward_bags(bags)
    (logits_a * w).sum().backward()
    logits_b = torch.cat([b(x)[0] for x in bags])
    (logits_b * w).sum().backward()
    torch.cuda.synchronize()
    assert (logits_a - logits_b).abs().max().item() < 1e-4
    for (na, pa), (nb, pb) in zip(a.named_parameters(), b.named_parameters()):
        assert na == nb
        if pb.grad is None:
            assert pa.grad is None
            continue
        scale = max(pb.grad.abs().max().item(), 1e-6)
        # + 2e-6: the attention-pooling output bias is softmax-invariant, its gradient is rounding noise around 0
        cases.record_grad_error("batched vs per-bag: " + na, (pa.grad - pb.grad).abs().max().item(), scale, GRAD_RTOL * scale + 2e-6)
        assert (pa.grad - pb.grad).abs().max().item() < GRAD_RTOL * scale + 2e-6, na


@pytest.mark.parametrize("B,P,K,linear", [(32, 12, 12, True), (5, 7, 4, True), (1, 1, 1, True), (70, 12, 8, False), (3, 16, 64, True)])
def test_fused_training_head_matches_torch_autograd(B, P, K, linear):
    """VF.head_train (mean pooling + adapter + normalise + cosine logits: two launches each way) vs the same math as torch ops
    under autograd (model/deepmil.py:203-204, model/vlsa.py:188-192), incl. gradients flowing into the returned unit features."""
    import torch.nn.functional as TF
    from vlsa_amd import functional as F
    dev = torch.device("cuda", 0)
    g = cases.gen(1200 + B)
    mk = lambda *s, sc=1.0: (torch.randn(*s, generator=g) * sc).to(dev).requires_grad_(True)  # noqa: E731
    rows, T, ls = mk(B, P, 512), mk(K, 512), torch.tensor(cases.LOGIT_SCALE, device=dev, requires_grad=True)
    W, b = (mk(512, 512, sc=512 ** -0.5), mk(512, sc=0.1)) if linear else (None, None)
    Gl, Gv, Gt = (torch.randn(B, K, generator=g).to(dev), torch.randn(B, 512, generator=g).to(dev) * 0.3,
                  torch.randn(K, 512, generator=g).to(dev) * 0.3)

    def ref():
        pooled = rows.mean(dim=1)
        v = pooled @ W.t() + b if linear else pooled
        vh, th = TF.normalize(v, dim=-1), TF.normalize(T, dim=-1)
        return ls.exp() * vh @ th.t(), vh, th
    leaves = [t for t in (rows, W, b, T, ls) if t is not None]
    for use_feats in (False, True):
        lo, vh, th = F.head_train(rows, W, b, T, ls)
        loss = (lo * Gl).sum() + ((vh * Gv).sum() + (th * Gt).sum() if use_feats else 0.0)
        got = torch.autograd.grad(loss, leaves)
        lo2, vh2, th2 = ref()
        loss2 = (lo2 * Gl).sum() + ((vh2 * Gv).sum() + (th2 * Gt).sum() if use_feats else 0.0)
        want = torch.autograd.grad(loss2, leaves)
        assert (lo - lo2).abs().max().item() < 1e-4 and (vh - vh2).abs().max().item() < 1e-6 and (th - th2).abs().max().item() < 1e-6
        for a, w_ in zip(got, want):
            assert a.shape == w_.shape
            assert (a - w_).abs().max().item() < 2e-5 * max(1.0, w_.abs().max().item()), (use_feats, a.shape)
