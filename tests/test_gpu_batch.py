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
