"""Batched persistent forward (B bags per launch) vs the single-bag path and the CPU oracle."""
import pytest
import torch

import cases
from oracle import vlsa_oracle as O

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
@pytest.mark.parametrize("sizes", [[50_000, 10_000, 2798, 33], [1, 16, 17, 32, 31, 64, 8192, 8200], [5000] * 8,
                                   [40, 0, 100_000, 7], [300]])
def test_batch_equals_single_bag_and_oracle(sizes, dtype):
    from vlsa_amd import functional as F
    dev = torch.device("cuda", 0)
    P, K, D = 12, 4, 512
    bags = [cases.make_bag(n, 300 + i).to(dtype).to(dev) if n > 0 else torch.empty(0, D, dtype=dtype, device=dev)
            for i, n in enumerate(sizes)]
    params = cases.make_params(P, K, 310)
    Q = (0.5 * params["resid"] + params["prompt"]).to(dev)
    T, W, b = params["T"].to(dev), params["W"].to(dev), params["b"].to(dev)
    ls = torch.tensor(cases.LOGIT_SCALE, device=dev)
    plan = F.VlfanBatchPlan(len(sizes), P, K, dev)
    plan.set_bags(bags)
    for rep in range(2):  # second run re-uses the workspace / ticket counters
        plan.run(Q, T, ls, W, b)
    torch.cuda.synchronize()
    for i, n in enumerate(sizes):
        if n == 0:
            continue  # an empty bag has no defined softmax (the reference would produce NaN too)
        single = F.VlfanInferencePlan(n, D, P, K, dev)
        ref_logits = single.run(bags[i], Q, T, ls, W, b)
        torch.cuda.synchronize()
        scale = max(1.0, single.out.abs().max().item())
        # fp32 bags: the batch kernel is exact f32 MFMA, the single-bag kernel a 2-term bf16 split of X (~2e-5 off)
        tol = 2e-5 if dtype == torch.bfloat16 else 1e-4
        assert (plan.out[i] - single.out).abs().max().item() < tol * scale, (i, n)
        assert (plan.logits[i] - ref_logits).abs().max().item() < tol, (i, n)
        assert (plan.vhat[i] - single.vhat).abs().max().item() < 1e-5
        if n <= 10_000:
            cpu = O.vlsa_vlfan_forward(bags[i].float().cpu(), Q.cpu(), T.cpu(), ls.cpu(), head_weight=W.cpu(), head_bias=b.cpu())
            assert (plan.logits[i].cpu() - cpu["logits"][0]).abs().max().item() < 1e-4, (i, n)
            assert (plan.incidence[i].cpu() - cpu["incidence"][0]).abs().max().item() < 1e-4


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
@pytest.mark.parametrize("reserved,groups", [(8, 0), (13, 0), (250, 0), (1000, 0), (0, 1), (0, 2), (32, 4), (0, 8), (32, 16), (0, 64)])
def test_partial_batch_with_reserved_cus_and_groups(reserved, groups, dtype):
    """Fewer persistent workgroups (CUs left free for tail / communication kernels) and other numbers of bags in flight:
    same partial merge result."""
    import ctypes
    from vlsa_amd import functional as F, _native as nat
    dev = torch.device("cuda", 0)
    P, K = 12, 4
    sizes = [5000, 64, 3333, 70, 9000, 1, 640, 2798, 100]
    bags = [cases.make_bag(n, 500 + i).to(dtype).to(dev) for i, n in enumerate(sizes)]
    params = cases.make_params(P, K, 510)
    Q = (0.5 * params["resid"] + params["prompt"]).to(dev)
    plan = F.VlfanBatchPlan(len(sizes), P, K, dev)
    plan.set_bags(bags)
    plan.run(Q, params["T"].to(dev), torch.tensor(cases.LOGIT_SCALE, device=dev), params["W"].to(dev), params["b"].to(dev))
    ref_out = plan.out.clone()
    lib, B, D = nat.load(), len(sizes), 512
    G = lib.vlsa_batch_partials_per_bag_ex(B, reserved, groups)
    S = {0: 8, 1: 1, 2: 2, 4: 4, 8: 8, 16: 8, 64: 8}[groups]      # power of two <= B = 9
    assert G == (256 - min((reserved + S - 1) // S * S, 256 - S)) // S
    plan.ws.zero_()
    nat.check(lib.vlsa_vlfan_partial_batch_ex(F._p(plan.desc), B, plan.dt, D, F._p(plan.qprep), P, F._p(plan.ws), reserved, groups,
                                              F._stream()), "partial_batch_ex")
    wf = plan.ws.view(torch.float32)
    n_ml = B * G * nat.P_STRIDE
    st = (ctypes.c_int64 * 9)(nat.P_STRIDE, nat.P_STRIDE, P * D, G * nat.P_STRIDE, G * nat.P_STRIDE, G * P * D,
                              nat.P_STRIDE, nat.P_STRIDE, P * D)
    m2, l, out = torch.empty(B, 16, device=dev), torch.empty(B, 16, device=dev), torch.empty(B, P, D, device=dev)
    nat.check(lib.vlsa_vlfan_merge_batch_strided(F._p(wf), F._p(wf[n_ml:]), F._p(wf[2 * n_ml:]), B, G, P, D, 1, st, F._p(m2),
                                                 F._p(l), F._p(out), F._stream()), "merge")
    torch.cuda.synchronize()
    assert (out - ref_out).abs().max().item() < 2e-5 * max(1.0, ref_out.abs().max().item())


def test_batch_plan_hipgraph_replay():
    """One captured run of a batch (hipGraph): replays reproduce the eager result and follow in-place parameter updates."""
    from vlsa_amd import functional as F
    dev = torch.device("cuda", 0)
    P, K = 12, 4
    sizes = [2798, 64, 1000, 5000, 333, 17, 2048, 4097]
    bags = [cases.make_bag(n, 700 + i).to(torch.bfloat16).to(dev) for i, n in enumerate(sizes)]
    params = cases.make_params(P, K, 710)
    Q = (0.5 * params["resid"] + params["prompt"]).to(dev)
    T, W, b = params["T"].to(dev), params["W"].to(dev), params["b"].to(dev)
    ls = torch.tensor(cases.LOGIT_SCALE, device=dev)
    plan = F.VlfanBatchPlan(len(sizes), P, K, dev)
    plan.set_bags(bags)
    eager = plan.run(Q, T, ls, W, b).clone()
    g = plan.capture(Q, T, ls, W, b)
    plan.logits.zero_()
    g.replay()
    torch.cuda.synchronize()
    assert torch.equal(plan.logits, eager)
    Q.mul_(-1.0)                                  # in-place update: the graph reads Q by address
    g.replay()
    torch.cuda.synchronize()
    ref = plan.logits.clone()
    assert (ref - eager).abs().max().item() > 1e-3
    assert torch.equal(plan.run(Q, T, ls, W, b), ref)
