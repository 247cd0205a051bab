"""More than 64 bags per forward launch (round 4: ``vlsa_batch_forward_max_bags`` = 256).  A workgroup of the persistent streaming
kernels keeps only ITS bags (grp, grp + S, ...) in its LDS table, so a launch takes up to S x 64 bags; the tails are grids over B.
Checked against the CPU oracle, against the <= 64-bag launches, for every admissible number of bags in flight, both dtypes,
ragged sizes, with attention weights, and through ``VLSA.forward_bags`` (which picks the launch width from the bag sizes)."""
import pytest
import torch

import cases
from oracle import vlsa_oracle as O

pytestmark = pytest.mark.gpu
TOL = 1e-4
SIZES = [2798, 17, 1, 900, 64, 65, 4100, 333, 127, 129, 31, 32, 1500, 63, 2047, 700]


def _inputs(n_bags, dtype, seed=9100, gated=False, P=12, K=4):
    sizes = [SIZES[(i * 7 + i // 16) % len(SIZES)] for i in range(n_bags)]
    bags = [cases.make_bag(n, seed + i, "clustered" if i % 3 else "iid").to(dtype) for i, n in enumerate(sizes)]
    params = cases.make_params(P, K, seed + 1000, gated)
    Q = 0.5 * params["resid"] + params["prompt"]
    return sizes, bags, params, Q


def _oracle_logits(x, Q, params, gated=False):
    return O.vlsa_vlfan_forward(x.float(), Q, params["T"], torch.tensor(cases.LOGIT_SCALE), head_weight=params["W"], head_bias=params["b"],
                                gated_query=gated)["logits"]


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
@pytest.mark.parametrize("n_bags", [65, 130, 256])
def test_wide_launch_vs_oracle_and_vs_64_bag_launches(n_bags, dtype):
    from vlsa_amd import functional as F
    dev = torch.device("cuda", 0)
    sizes, bags, params, Q = _inputs(n_bags, dtype)
    dbags = [x.to(dev) for x in bags]
    args = [t.to(dev) for t in (Q, params["T"], torch.tensor(cases.LOGIT_SCALE), params["W"], params["b"])]
    plan = F.VlfanBatchPlan(n_bags, 12, 4, dev)
    plan.set_bags(dbags)
    assert plan.groups >= (n_bags + 63) // 64
    logits = plan.run(*args).clone()
    parts = []
    for i in range(0, n_bags, 64):
        p = F.VlfanBatchPlan(len(dbags[i:i + 64]), 12, 4, dev)
        p.set_bags(dbags[i:i + 64])
        parts.append(p.run(*args).clone())
    narrow = torch.cat(parts)
    assert (logits - narrow).abs().max().item() < 5e-5           # (different partial splits: not bit-equal; both within 1e-4 of the oracle)
    for i in list(range(0, n_bags, 11)) + [n_bags - 1]:
        ref = _oracle_logits(bags[i], Q, params)
        assert (logits[i:i + 1].cpu() - ref).abs().max().item() < TOL, (i, sizes[i])


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
def test_every_number_of_bags_in_flight(dtype):
    """S = 4 ... 256 for 256 bags (S = 256: one workgroup per bag, one partial record per bag); asking for fewer than B / 64 is raised"""
    from vlsa_amd import functional as F
    dev = torch.device("cuda", 0)
    sizes, bags, params, Q = _inputs(256, dtype, seed=9400)
    dbags = [x.to(dev) for x in bags]
    args = [t.to(dev) for t in (Q, params["T"], torch.tensor(cases.LOGIT_SCALE), params["W"], params["b"])]
    plan = F.VlfanBatchPlan(256, 12, 4, dev)
    plan.set_bags(dbags)
    ref = {i: _oracle_logits(bags[i], Q, params) for i in (0, 1, 2, 63, 64, 128, 200, 255)}
    seen = []
    for S in (1, 4, 8, 16, 32, 64, 128, 256):
        plan.groups = S
        logits = plan.run(*args).clone()
        seen.append(logits)
        for i, r in ref.items():
            assert (logits[i:i + 1].cpu() - r).abs().max().item() < TOL, (S, i, sizes[i])
    assert torch.equal(seen[0], seen[1])                          # S = 1 is not admissible for 256 bags: runs as S = 4
    assert max((a - seen[0]).abs().max().item() for a in seen) < 5e-5


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
@pytest.mark.parametrize("gated", [False, True])
def test_wide_launch_attention_weights(dtype, gated):
    from vlsa_amd import functional as F
    dev = torch.device("cuda", 0)
    sizes, bags, params, Q = _inputs(150, dtype, seed=9700, gated=gated)
    args = [t.to(dev) for t in (Q, params["T"], torch.tensor(cases.LOGIT_SCALE), params["W"], params["b"])]
    plan = F.VlfanBatchPlan(150, 12, 4, dev, gated=gated, want_attn=True)
    plan.set_bags([x.to(dev) for x in bags])
    logits = plan.run(*args).clone()
    for i in (0, 5, 64, 65, 100, 149):
        r = O.vlfan_forward(bags[i].float(), Q, gated_query=gated)
        A = plan.attn.views[i].cpu()
        assert A.shape == r["A"].shape and (A - r["A"]).abs().max().item() < TOL, (i, sizes[i])
        assert (logits[i:i + 1].cpu() - _oracle_logits(bags[i], Q, params, gated)).abs().max().item() < TOL


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
def test_wide_launch_with_empty_bags_and_one_big_bag(dtype):
    """empty bags (no defined softmax: their rows are skipped, as in the narrow tests), a 70k-patch bag among slide-sized ones (its
    group's workgroups carry it alone), a second run on the same plan (workspace / descriptor reuse)"""
    from vlsa_amd import functional as F
    dev = torch.device("cuda", 0)
    sizes = [SIZES[i % len(SIZES)] for i in range(140)]
    for i in (0, 63, 64, 65, 139):
        sizes[i] = 0
    sizes[77] = 70_000
    bags = [cases.make_bag(n, 9800 + i).to(dtype) if n > 0 else torch.empty(0, 512, dtype=dtype) for i, n in enumerate(sizes)]
    params = cases.make_params(12, 4, 9801)
    Q = 0.5 * params["resid"] + params["prompt"]
    args = [t.to(dev) for t in (Q, params["T"], torch.tensor(cases.LOGIT_SCALE), params["W"], params["b"])]
    plan = F.VlfanBatchPlan(140, 12, 4, dev)
    plan.set_bags([x.to(dev) for x in bags])
    first = plan.run(*args).clone()
    logits = plan.run(*args).clone()
    live = [i for i, n in enumerate(sizes) if n > 0]
    assert torch.equal(first[live], logits[live])
    for i in (1, 62, 66, 77, 100, 138):
        assert (logits[i:i + 1].cpu() - _oracle_logits(bags[i], Q, params)).abs().max().item() < TOL, (i, sizes[i])


def test_module_picks_the_launch_width_from_the_bag_sizes():
    from vlsa_amd import functional as F
    from test_gpu_bagset import _net
    net, params = _net(K=5)
    net.eval()
    sizes, bags, _, _ = _inputs(300, torch.bfloat16, seed=9900)
    dbags = [x.cuda() for x in bags]
    with torch.no_grad():
        wide = net.forward_bags(F.BagSet(dbags))
        narrow = torch.cat([net.forward_bags(dbags[i:i + 64])[0] for i in range(0, 300, 64)])
        big = [cases.make_bag(30_000, 9990 + i).to(torch.bfloat16).cuda() for i in range(66)]        # 30k-patch bags stay at 64 per launch
        net.forward_bags(big)
    assert tuple(wide[0].shape) == (300, 5) and (wide[0] - narrow).abs().max().item() < 5e-5
    widths = sorted(k[1] for k in net._plans if k[0] == "batch")
    assert 256 in widths and 44 in widths and 64 in widths and 2 in widths, widths
    Q = 0.5 * params["resid"] + params["prompt"]
    for i in (0, 255, 256, 299):
        ref = O.vlsa_vlfan_forward(bags[i].float(), Q, params["T"], net.logit_scale.detach().cpu(), head_weight=params["W"],
                                   head_bias=params["b"])["logits"]
        assert (wide[0][i:i + 1].cpu() - ref).abs().max().item() < TOL


def test_limits():
    from vlsa_amd import _native as nat
    from vlsa_amd import functional as F
    lib = nat.load()
    assert lib.vlsa_batch_max_bags() == 64 and lib.vlsa_batch_forward_max_bags() == 256
    assert lib.vlsa_batch_workspace_bytes(256, 12, 512) - 256 * 64 == lib.vlsa_batch_workspace_bytes(64, 12, 512) - 64 * 64
    with pytest.raises(ValueError):
        F.VlfanBatchPlan(257, 12, 4, torch.device("cuda", 0))
    with pytest.raises(ValueError):
        F._BagTable([torch.zeros(4, 512, device="cuda")] * 65)          # the training / score launches stay at 64
