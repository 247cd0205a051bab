"""Multi-GPU driver code on one GPU: (i) records of several shards merged with the strided HIP merge equal the
unsharded result; (ii) ShardedVlfanPlan / sharded_vlfan_forward over a 1-rank RCCL group (exercises the collective,
the side-stream pipeline and the record layout end to end)."""
import os

import numpy as np
import pytest
import torch

import cases
from oracle import vlsa_oracle as O

pytestmark = pytest.mark.gpu


def test_strided_merge_of_shard_records_equals_unsharded():
    import ctypes
    from vlsa_amd import _native as nat, functional as F
    from vlsa_amd.sharded import REC_HDR, record_floats, shard_bounds
    N, P, D, world = 20_001, 12, 512, 8
    X = cases.make_bag(N, 5).to(torch.bfloat16).cuda()
    params = cases.make_params(P, 4, 6)
    Q = (0.5 * params["resid"] + params["prompt"]).cuda()
    qp = F.prepare_queries(Q)
    rf = record_floats(P, D)
    gathered = torch.zeros(world, rf, device="cuda")
    for r in range(world):
        a, b = shard_bounds(N, world, r)
        pm, pl, pacc, _ = F.vlfan_partial(X[a:b], qp)
        m2, l, acc = F.vlfan_merge(pm, pl, pacc, normalise=False)
        gathered[r, :16], gathered[r, 16:32], gathered[r, 32:] = m2, l, acc.reshape(-1)
    lib = nat.load()
    m2g = torch.empty(16, device="cuda"); lg = torch.empty(16, device="cuda"); out = torch.empty(P, D, device="cuda")
    base = gathered.data_ptr()
    nat.check(lib.vlsa_vlfan_merge_strided(ctypes.c_void_p(base), rf, ctypes.c_void_p(base + 64), rf,
                                           ctypes.c_void_p(base + 4 * REC_HDR), rf, world, P, D, 1, F._p(m2g), F._p(lg),
                                           F._p(out), F._stream()), "merge_strided")
    ref, _, _ = F.vlfan_aggregate(X, Q)
    scale = max(1.0, ref.abs().max().item())
    assert (out - ref).abs().max().item() < 1e-4 * scale
    cpu = O.vlfan_forward(X.float().cpu(), Q.cpu())
    assert (out.cpu() - cpu["out"]).abs().max().item() < 1e-4 * scale


def test_sharded_plan_over_one_rank_rccl_group():
    import torch.distributed as dist
    from vlsa_amd import functional as F
    from vlsa_amd.sharded import ShardedVlfanPlan, sharded_vlfan_forward
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29611")
    dev = torch.device("cuda", 0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    try:
        N, P, K, D = 5000, 12, 4, 512
        bags = [cases.make_bag(N, 20 + i).to(torch.bfloat16).cuda() for i in range(3)]
        params = cases.make_params(P, K, 30)
        Q = (0.5 * params["resid"] + params["prompt"]).cuda()
        T, W, b = params["T"].cuda(), params["W"].cuda(), params["b"].cuda()
        ls = torch.tensor(cases.LOGIT_SCALE, device=dev)
        ref_plan = F.VlfanInferencePlan(N, D, P, K, dev)
        refs = [ref_plan.run(x, Q, T, ls, W, b).clone() for x in bags]
        for pipeline in (False, True):
            plan = ShardedVlfanPlan(N, D, P, K, dev, dist, pipeline=pipeline)
            got = []
            for i, x in enumerate(bags):
                plan.run(x, Q, T, ls, W, b)
                if not pipeline:
                    got.append(plan.local.logits.clone())
                elif i > 0:
                    got.append(plan.local.logits.clone())   # logits of bag i-1
            if pipeline:
                got.append(plan.finish().clone())
            torch.cuda.synchronize()
            for g, r in zip(got, refs):
                assert (g - r).abs().max().item() < 1e-5, pipeline
        # batched sharded plan (B bags per launch, one all-gather per batch)
        from vlsa_amd.sharded import ShardedVlfanBatchPlan
        for pipeline, reserved in ((False, None), (True, None), (True, 8), (True, 100)):
            # reserved: CUs left without a persistent workgroup (room for RCCL next to the streaming kernel)
            bp = ShardedVlfanBatchPlan(3, P, K, dev, dist, pipeline=pipeline, reserved_cus=reserved)
            bp.set_bags(bags)
            bp.run(Q, T, ls, W, b)
            bp.run(Q, T, ls, W, b)
            got = bp.finish().clone()
            torch.cuda.synchronize()
            for i in range(3):
                assert (got[i] - refs[i]).abs().max().item() < 2e-5, (pipeline, i)
        out, A = sharded_vlfan_forward(bags[0], Q, want_attn=True)
        ref_out, ref_A, _ = F.vlfan_aggregate(bags[0], Q, want_attn=True)
        assert (out - ref_out).abs().max().item() < 1e-5 and (A - ref_A).abs().max().item() < 1e-6
    finally:
        dist.destroy_process_group()


@pytest.mark.gpu
def test_bag_parallel_training_under_ddp():
    """Training data parallelism has BAGS as the unit: the model wrapped in torch DDP over RCCL, each rank's
    bags through the batched HIP forward + backward, gradients all-reduced by DDP.  One rank here (the multi-rank reduction
    is DDP's own); checks that the custom autograd functions and the list-of-bags forward work under the wrapper."""
    import os
    import torch.distributed as dist
    from torch.nn.parallel import DistributedDataParallel as DDP
    from vlsa_amd.vlsa import VLSA
    from vlsa_amd.losses import SurvObjective
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29713", RANK="0", WORLD_SIZE="1")
    dev = torch.device("cuda", 0)
    dist.init_process_group("nccl", device_id=dev)
    try:
        P, K = 12, 4
        params = cases.make_params(P, K, 5000)
        bags = [cases.make_bag(n, 5010 + i).to(torch.bfloat16).to(dev) for i, n in enumerate([900, 64, 2798, 333])]
        t, e = torch.tensor([0, 3, 1, 2], device=dev), torch.tensor([1.0, 0.0, 1.0, 0.0], device=dev)

        def build():
            cfg = dict(name="VLFAN", dim_in=512, use_feat_proj=False, query="Parameter", num_query=P, query_pooling="mean")
            m = VLSA.from_modules(cfg, pretrained_text_features=params["T"].clone()).to(dev)
            with torch.no_grad():
                m.mil_encoder.Q.copy_((0.5 * params["resid"] + params["prompt"]).to(dev))
                m.mil_encoder.visual_adapter.weight.copy_(params["W"].to(dev))
                m.mil_encoder.visual_adapter.bias.copy_(params["b"].to(dev))
            return m.train()

        plain, wrapped = build(), DDP(build(), device_ids=[0])
        objective = SurvObjective()
        for model in (plain, wrapped):
            logits = model(bags)[0]
            net = model.module if isinstance(model, DDP) else model
            objective(logits, t, e, net.get_logit_scale()).backward()
        torch.cuda.synchronize()
        for (na, pa), (nb, pb) in zip(plain.named_parameters(), wrapped.module.named_parameters()):
            assert na == nb and (pa.grad is None) == (pb.grad is None)
            if pa.grad is not None:
                assert (pa.grad - pb.grad).abs().max().item() <= 1e-6 * max(1.0, pa.grad.abs().max().item()), na
    finally:
        dist.destroy_process_group()
