"""world_size = 2 on ONE GPU: two processes share cuda:0 and exchange their compact records through a gloo group (RCCL does
not allow two ranks on one device; gloo moves CUDA tensors through the host).  Everything above the transport is the real
N > 1 code on the real kernels: shard bounds, per-rank persistent kernel with reserved CUs, record fold, all-gather,
global merge + head, the side-stream pipeline of ShardedVlfanBatchPlan and ShardedVlfanPlan.  Both ranks must reproduce the
unsharded result."""
import os

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import cases

pytestmark = pytest.mark.gpu


def _worker(rank, world, port, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from vlsa_amd import functional as F
        from vlsa_amd.sharded import ShardedVlfanBatchPlan, ShardedVlfanPlan, shard_bounds
        dev = torch.device("cuda", 0)
        torch.cuda.set_device(dev)
        P, K, B = 12, 4, 5
        sizes = [20_000, 333, 4097, 64, 9000]
        params = cases.make_params(P, K, 8200)
        Q = (0.5 * params["resid"] + params["prompt"]).to(dev)
        T, W, b = params["T"].to(dev), params["W"].to(dev), params["b"].to(dev)
        ls = torch.tensor(cases.LOGIT_SCALE, device=dev)
        full = [cases.make_bag(n, 8210 + i).to(torch.bfloat16).to(dev) for i, n in enumerate(sizes)]
        shards = []
        for x in full:
            a, c = shard_bounds(x.shape[0], world, rank)
            shards.append(x[a:c])
        errs = {}
        # unsharded reference on this rank
        ref_plan = F.VlfanBatchPlan(B, P, K, dev)
        ref_plan.set_bags(full)
        ref = ref_plan.run(Q, T, ls, W, b).clone()
        for pipeline in (False, True):
            bp = ShardedVlfanBatchPlan(B, P, K, dev, dist, pipeline=pipeline)
            bp.set_bags(shards)
            for _ in range(3):
                bp.run(Q, T, ls, W, b)
            got = bp.finish().clone()
            torch.cuda.synchronize()
            errs[f"batch pipeline={pipeline}"] = (got - ref).abs().max().item()
        # batched plan with attention weights, pipelined over two DIFFERENT batches of equal sizes: batch 2 streams before
        # batch 1's tail runs, so batch 1's weights must come from its own per-slot score buffer
        from oracle import vlsa_oracle as Orc
        sizes2 = [3000, 517, 64]
        setA = [cases.make_bag(n, 8400 + i).to(torch.bfloat16) for i, n in enumerate(sizes2)]
        setB = [cases.make_bag(n, 8500 + i).to(torch.bfloat16) for i, n in enumerate(sizes2)]
        bnd = [shard_bounds(n, world, rank) for n in sizes2]
        for pipeline in (True, False):
            bp = ShardedVlfanBatchPlan(len(sizes2), P, K, dev, dist, pipeline=pipeline, want_attn=True)
            got = []
            for bags_ in (setA, setB):
                bp.set_bags([x[a:c].to(dev) for x, (a, c) in zip(bags_, bnd)])
                bp.run(Q, T, ls, W, b)
                if not pipeline:
                    got.append([v.clone() for v in bp.A])
                elif len(got) == 0 and bags_ is setB:
                    got.append([v.clone() for v in bp.A])      # batch A's weights, drained by batch B's run()
            if pipeline:
                bp.finish()
                got.append([v.clone() for v in bp.A])
            torch.cuda.synchronize()
            e = 0.0
            for views, bags_ in zip(got, (setA, setB)):
                for v, x, (a, c) in zip(views, bags_, bnd):
                    e = max(e, (v.cpu() - Orc.vlfan_forward(x.float(), Q.cpu())["A"][:, a:c]).abs().max().item())
            errs[f"attn batch pipeline={pipeline}"] = e
        # single-bag sharded plan, pipelined over 3 bags
        a, c = shard_bounds(sizes[0], world, rank)
        sp = ShardedVlfanPlan(c - a, 512, P, K, dev, dist, pipeline=True)
        outs = []
        for i in range(3):
            sp.run(full[0][a:c], Q, T, ls, W, b)
            if i > 0:
                outs.append(sp.local.logits.clone())
        outs.append(sp.finish().clone())
        torch.cuda.synchronize()
        errs["single"] = max((o - ref[0]).abs().max().item() for o in outs)
        # attention weights through the PIPELINED plan (model/deepmil.py:198,206-215): three DIFFERENT bags of one size;
        # the streaming kernel of bag i+1 runs before the tail of bag i, so bag i's A[:, shard] must still come from bag
        # i's scores.  Checked against the CPU oracle on the full bag.
        from oracle import vlsa_oracle as O
        n3 = 6000
        a, c = shard_bounds(n3, world, rank)
        bags3 = [cases.make_bag(n3, 8300 + i).to(torch.bfloat16) for i in range(3)]
        bags3[1][17] = 0.0                                    # an all-zero patch row in the middle bag
        refs3 = [O.vlfan_forward(x.float(), Q.cpu()) for x in bags3]
        for pipeline in (True, False):
            sp = ShardedVlfanPlan(c - a, 512, P, K, dev, dist, want_attn=True, pipeline=pipeline)
            got_A, got_l = [], []
            for i, x in enumerate(bags3):
                sp.run(x[a:c].to(dev), Q, T, ls, W, b)
                if pipeline and i > 0:
                    got_A.append(sp.A.clone()); got_l.append(sp.local.logits.clone())
                elif not pipeline:
                    got_A.append(sp.A.clone()); got_l.append(sp.local.logits.clone())
            if pipeline:
                sp.finish()
                got_A.append(sp.A.clone()); got_l.append(sp.local.logits.clone())
            torch.cuda.synchronize()
            errs[f"attn pipeline={pipeline}"] = max((A.cpu() - r["A"][:, a:c]).abs().max().item() for A, r in zip(got_A, refs3))
            distinct = (refs3[0]["A"][:, a:c] - refs3[1]["A"][:, a:c]).abs().max().item()
            assert distinct > 1e-3            # the three bags really have different weights
        ret[rank] = errs
    finally:
        dist.destroy_process_group()


def test_two_ranks_on_one_gpu_reproduce_the_unsharded_result():
    world = 2
    mgr = mp.Manager()
    ret = mgr.dict()
    port = 29800 + (os.getpid() % 150)
    mp.spawn(_worker, args=(world, port, ret), nprocs=world, join=True)
    assert len(ret) == world
    for r in range(world):
        for what, err in ret[r].items():
            assert err < (1e-4 if what.startswith("attn") else 2e-5), (r, what, err)
