"""The three exchanges of the patch-sharded batch plan (vlsa_amd/sharded.py) with 2 / 4 / 8 PROCESSES sharing ONE GPU: "owner"
(all_to_all_single of the records to their bag owners + all-gather of the packed results, here over gloo: RCCL refuses two ranks on a
device), "ipc" (kernels storing into peer buffers mapped through hipIpc, epoch flags, acknowledgement gates: csrc/xchg.hip -- the real
transport, IPC works between processes on one device) and "allgather" (rounds 1-4).  Every rank must reproduce the unsharded batch
plan: logits, incidence, unit image features, (m2, l), and -- sharded -- the attention weights against the CPU oracle; pipelined and
not, a batch size the ranks do not divide, fewer bags than ranks, several launches through both pipeline slots."""
import os

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import cases

pytestmark = pytest.mark.gpu


def _worker(rank, world, port, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle import vlsa_oracle as O
        from vlsa_amd import functional as F
        from vlsa_amd.sharded import ShardedVlfanBatchPlan, shard_bounds
        dev = torch.device("cuda", 0)
        torch.cuda.set_device(dev)
        P, K = 12, 4
        params = cases.make_params(P, K, 9200)
        Q = (0.5 * params["resid"] + params["prompt"]).to(dev)
        T, W, b = params["T"].to(dev), params["W"].to(dev), params["b"].to(dev)
        ls = torch.tensor(cases.LOGIT_SCALE, device=dev)
        errs = {}
        for B in (2 * world + 1, max(1, world - 1)):                      # not divisible by the ranks; fewer bags than ranks
            sizes = [3000 + 517 * i if i % 3 else 64 + i for i in range(B)]
            full = [cases.make_bag(n, 9300 + i).to(torch.bfloat16).to(dev) for i, n in enumerate(sizes)]
            full2 = [cases.make_bag(n, 9400 + i).to(torch.bfloat16).to(dev) for i, n in enumerate(sizes)]
            bnd = [shard_bounds(n, world, rank) for n in sizes]
            cut = lambda bags: [x[a:c] for x, (a, c) in zip(bags, bnd)]  # noqa: E731
            ref_plan = F.VlfanBatchPlan(B, P, K, dev)
            refs = []
            for bags in (full, full2):
                ref_plan.set_bags(bags)
                ref_plan.run(Q, T, ls, W, b)
                refs.append([t.clone() for t in (ref_plan.logits, ref_plan.incidence, ref_plan.vhat, ref_plan.m2, ref_plan.l)])
            for exchange in ("owner", "ipc", "allgather"):
                for pipeline in (False, True):
                    tag = f"B={B} {exchange} pipeline={pipeline}"
                    bp = ShardedVlfanBatchPlan(B, P, K, dev, dist, pipeline=pipeline, exchange=exchange, timeout_s=20.0)
                    # five launches alternating between two different batches: both slots, the gates of the peer-write protocol
                    got = []
                    for i in range(5):
                        bp.set_bags(cut(full if i % 2 == 0 else full2))
                        bp.run(Q, T, ls, W, b)
                        if pipeline and i > 0:
                            got.append((i - 1, [t.clone() for t in (bp.logits, bp.incidence, bp.vhat, bp.m2, bp.l)]))
                        elif not pipeline:
                            got.append((i, [t.clone() for t in (bp.logits, bp.incidence, bp.vhat, bp.m2, bp.l)]))
                    if pipeline:
                        bp.finish()
                        got.append((4, [t.clone() for t in (bp.logits, bp.incidence, bp.vhat, bp.m2, bp.l)]))
                    torch.cuda.synchronize()
                    e = 0.0
                    for i, outs in got:
                        want = refs[i % 2]
                        for name, g, w_ in zip(("logits", "incidence", "vhat"), outs, want):
                            e = max(e, float((g - w_).abs().max()))
                        # (m2, l): the kernels' running reference maximum is not unique, the log-sum-exp m2 + log2(l) is
                        lse_g = outs[3][:, :P] + torch.log2(outs[4][:, :P])
                        lse_w = want[3][:, :P] + torch.log2(want[4][:, :P])
                        e = max(e, float((lse_g - lse_w).abs().max()) * 0.1)          # 2e-4 in log2 units passes
                    errs[tag] = e
                    errs[tag + " status"] = float(bp.status())
                    xb = bp.exchange_bytes()
                    assert xb["sent"] >= 0 and xb["received"] >= 0
                    bp.close()
            # attention weights, sharded, through the owner and the peer-write exchange (pipelined: batch 2 streams before batch 1's tail)
            for exchange in ("owner", "ipc"):
                bp = ShardedVlfanBatchPlan(B, P, K, dev, dist, pipeline=True, exchange=exchange, want_attn=True, timeout_s=20.0)
                got = []
                for bags in (full, full2):
                    bp.set_bags(cut(bags))
                    bp.run(Q, T, ls, W, b)
                    if bags is full2:
                        got.append([v.clone() for v in bp.A])              # batch 1's weights, drained by batch 2's run()
                bp.finish()
                got.append([v.clone() for v in bp.A])
                torch.cuda.synchronize()
                e = 0.0
                for views, bags in zip(got, (full, full2)):
                    for i in (0, B - 1):
                        a, c = bnd[i]
                        refA = O.vlfan_forward(bags[i].float().cpu(), Q.cpu())["A"][:, a:c]
                        assert tuple(views[i].shape) == (P, c - a)
                        if c > a:
                            e = max(e, float((views[i].cpu() - refA).abs().max()))
                errs[f"B={B} {exchange} attn"] = e
                errs[f"B={B} {exchange} attn status"] = float(bp.status())
                bp.close()
        ret[rank] = errs
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 4, 8])
def test_exchanges_on_one_gpu_reproduce_the_unsharded_result(world):
    mgr = mp.Manager()
    ret = mgr.dict()
    port = 29300 + (os.getpid() % 150) + world
    mp.spawn(_worker, args=(world, port, ret), nprocs=world, join=True)
    assert len(ret) == world
    for r in range(world):
        for what, err in ret[r].items():
            if what.endswith("status"):
                assert err == 0.0, (r, what, err)
            else:
                assert err < (1e-4 if "attn" in what else 2e-5), (r, what, err)


def test_a_peer_that_never_arrives_sets_the_status_bits_instead_of_hanging():
    """one rank, world 1 faked as 2 through the flag tables: a wait on a flag nobody raises returns after the time-out"""
    import ctypes
    from vlsa_amd import _native as nat
    from vlsa_amd import functional as VF
    lib = nat.load()
    dev = torch.device("cuda", 0)
    flags = torch.zeros(4, dtype=torch.int32, device=dev)
    status = torch.zeros(1, dtype=torch.int32, device=dev)
    V = ctypes.c_void_p * 2
    tab = V(flags.data_ptr(), flags.data_ptr() + 4)
    torch.cuda.synchronize()
    import time
    t0 = time.perf_counter()
    nat.check(lib.vlsa_xchg_wait(2, tab, 1, int(0.2 * 100e6), VF._p(status), VF._stream()), "vlsa_xchg_wait")
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    assert int(status.item()) == 2 and 0.15 < dt < 5.0
    flags[:2] = 1
    status.zero_()
    nat.check(lib.vlsa_xchg_wait(2, tab, 1, int(0.2 * 100e6), VF._p(status), VF._stream()), "vlsa_xchg_wait")
    assert int(status.item()) == 0
