"""N > 1 path on CPU with gloo (world_size 2): the shard partition, the one-collective exchange of compact records and
the log-sum-exp merge reproduce the single-process result.  The local partial and the merge arithmetic are provided
here by the CPU oracle (test infrastructure) -- on a GPU box the same driver code calls the HIP kernels."""
import os
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import cases
from oracle import vlsa_oracle as O


def test_shard_bounds_partition_rows():
    from vlsa_amd.sharded import shard_bounds
    for N in (0, 1, 15, 16, 17, 1000, 50_000, 200_000, 199_999):
        for world in (1, 2, 3, 8):
            edges = [shard_bounds(N, world, r) for r in range(world)]
            assert edges[0][0] == 0 and edges[-1][1] == N
            for (a, b), (c, d) in zip(edges[:-1], edges[1:]):
                assert b == c and a <= b
            sizes = [b - a for a, b in edges]
            assert max(sizes) - min(sizes) <= 32
            assert all(a % 16 == 0 or a == N for a, _ in edges)


def _worker(rank, world, port, N, P, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from vlsa_amd import sharded
        X = cases.make_bag(N, 11)
        params = cases.make_params(P, 4, 12)
        Q = 0.5 * params["resid"] + params["prompt"]
        a, b = sharded.shard_bounds(N, world, rank)
        Qh = O.l2_normalize(Q)
        m, l, acc, s = O.vlfan_partial(X[a:b], Qh)              # natural-log-domain partial of this shard
        LOG2E = 1.4426950408889634
        rec = torch.zeros(sharded.record_floats(P, 512))
        rec[:P] = m * LOG2E                                       # ABI records carry the log2-domain max
        rec[16:16 + P] = l
        rec[32:] = acc.reshape(-1)
        gathered = torch.empty(world, rec.numel())
        sharded.all_gather_records(rec, gathered)
        ms = [gathered[r, :P] / LOG2E for r in range(world)]
        ls = [gathered[r, 16:16 + P] for r in range(world)]
        accs = [gathered[r, 32:].reshape(P, 512) for r in range(world)]
        mg, lg, out = O.merge_partials(ms, ls, accs)
        A_local = torch.exp(s - mg[:, None]) / lg[:, None]
        ref = O.vlfan_forward(X, Q)
        err_out = (out - ref["out"]).abs().max().item()
        err_A = (A_local - ref["A"][:, a:b]).abs().max().item()
        ret[rank] = (err_out, err_A, a, b)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("N", [1000, 33])
def test_two_rank_gloo_exchange_matches_single_process(N):
    world, P = 2, 12
    mgr = mp.Manager()
    ret = mgr.dict()
    port = 29500 + (os.getpid() % 500) + N % 7
    mp.spawn(_worker, args=(world, port, N, P, ret), nprocs=world, join=True)
    assert len(ret) == world
    for r in range(world):
        err_out, err_A, a, b = ret[r]
        assert err_out < 1e-4 and err_A < 1e-5, (r, err_out, err_A)
    assert ret[0][3] == ret[1][2]


# ---- zero-shot top-k and attention pooling over N, patch-sharded: exchange + re-selection logic on CPU (gloo) -------------
def _worker_other(rank, world, port, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from vlsa_amd import sharded
        N, K, k = 1000, 4, 10
        X = cases.make_bag(N, 21)
        T = cases.make_params(1, K, 22)["T"]
        a, b = sharded.shard_bounds(N, world, rank)
        cos = (O.l2_normalize(X[a:b]) @ O.l2_normalize(T).t()).t().contiguous()          # [K, n_loc]
        vals = torch.full((K, k), float("-inf"))
        kk = min(k, cos.shape[1])
        vals[:, :kk] = cos.topk(kk, dim=1).values                                        # what vlsa_topk_values leaves per rank
        cand = sharded.merge_topk_candidates(sharded.gather_rows(vals))
        pooled = cand.topk(k, dim=1).values.mean(dim=1)
        ref_logits = O.vlsa_zeroshot_forward(X, T, torch.tensor(cases.LOGIT_SCALE), "logit_top10")[0]
        err_zs = (pooled * torch.tensor(cases.LOGIT_SCALE).exp() - ref_logits[0]).abs().max().item()
        # attention pooling over N: per-rank online-softmax record, one gather, log-sum-exp merge
        pp = cases.make_pool_params("gated_attention", 23)
        full = O.gated_attention_pooling(X, pp["wa"], pp["ba"], pp["wg"], pp["bg"], pp["w2"], pp["b2"])
        raw = full[1][a:b]
        m = raw.max()
        w = torch.exp(raw - m)
        rec = torch.cat([m.reshape(1), w.sum().reshape(1), w @ X[a:b]])
        g = sharded.gather_rows(rec)
        mg = g[:, 0].max()
        f = torch.exp(g[:, 0] - mg)
        pooled_attn = (f[:, None] * g[:, 2:]).sum(dim=0) / (f * g[:, 1]).sum()
        ret[rank] = (err_zs, (pooled_attn - full[0]).abs().max().item())
    finally:
        dist.destroy_process_group()


def test_two_rank_zeroshot_topk_and_attention_pool_exchange():
    world = 2
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker_other, args=(world, 29300 + (os.getpid() % 500), ret), nprocs=world, join=True)
    assert len(ret) == world
    for r in range(world):
        assert ret[r][0] < 1e-4 and ret[r][1] < 1e-5, ret[r]


# ---- bag-owner exchange (round 5): records to their owners (all_to_all_single with split sizes), owner-side fold, packed results to
# everyone, assembly in the caller's bag order -- the partition / split / ordering logic of ShardedVlfanBatchPlan(exchange="owner")
def test_owner_partition_covers_every_bag_once():
    from vlsa_amd.sharded import owner_partition
    for B in (1, 3, 5, 8, 64, 65):
        for world in (1, 2, 3, 4, 8):
            perm, counts, starts = owner_partition(B, world)
            assert sorted(perm) == list(range(B)) and sum(counts) == B and len(counts) == world
            for o in range(world):
                mine = perm[starts[o]:starts[o] + counts[o]]
                assert mine == list(range(o, B, world))                     # bag b is owned by rank b % world, in rising order
            assert max(counts) - min(counts) <= 1 and max(counts) == counts[0]


def _worker_owner(rank, world, port, B, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from vlsa_amd import sharded
        P, K = 4, 3
        LOG2E = 1.4426950408889634
        rf = sharded.record_floats(P, 512)
        params = cases.make_params(P, K, 31)
        Q = 0.5 * params["resid"] + params["prompt"]
        Qh = O.l2_normalize(Q)
        sizes = [40 + 37 * i for i in range(B)]
        bags = [cases.make_bag(n, 700 + i) for i, n in enumerate(sizes)]
        perm, counts, starts = sharded.owner_partition(B, world)
        n_me = counts[rank]
        # local records in owner-major order (what the streaming kernel + local fold leave in `rec`)
        rec = torch.zeros(B, rf)
        for t, bidx in enumerate(perm):
            a, b = sharded.shard_bounds(sizes[bidx], world, rank)
            if b > a:
                m, l, acc, _ = O.vlfan_partial(bags[bidx][a:b], Qh)
                rec[t, :P], rec[t, 16:16 + P], rec[t, 32:] = m * LOG2E, l, acc.reshape(-1)
            else:
                rec[t, :P] = float("-inf")                                   # an empty shard: the neutral record
        recv = torch.empty(world, max(n_me, 1), rf)
        dist.all_to_all_single(recv.view(-1)[:world * n_me * rf], rec.view(-1), output_split_sizes=[n_me * rf] * world,
                               input_split_sizes=[c * rf for c in counts])
        # owner-side fold + "head" (here: the pooled mean row's first K entries stand in for the logits)
        nmax = max(counts)
        res = torch.zeros(nmax, K + 16)
        for j in range(n_me):
            ms = [recv[r, j, :P] / LOG2E for r in range(world)]
            ls = [recv[r, j, 16:16 + P] for r in range(world)]
            accs = [recv[r, j, 32:].reshape(P, 512) for r in range(world)]
            mg, lg, out = O.merge_partials(ms, ls, accs)
            res[j, :K] = out.mean(dim=0)[:K]
            res[j, K:K + P] = lg
        resg = torch.empty(world, nmax, K + 16)
        dist.all_gather_into_tensor(resg.view(-1), res.view(-1))
        # assembly in the caller's order: bag b = j * world + o  (what vlsa_xchg_collect does on the device)
        got = torch.stack([resg[b % world, b // world] for b in range(B)])
        err = 0.0
        for bidx in range(B):
            ref = O.vlfan_forward(bags[bidx], Q)["out"].mean(dim=0)[:K]
            err = max(err, (got[bidx, :K] - ref).abs().max().item())
        ret[rank] = (err, got[:, :K].clone())
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,B", [(2, 5), (3, 4), (3, 2)])
def test_owner_exchange_over_gloo_matches_the_unsharded_result(world, B):
    mgr = mp.Manager()
    ret = mgr.dict()
    port = 29100 + (os.getpid() % 300) + 7 * world + B
    mp.spawn(_worker_owner, args=(world, port, B, ret), nprocs=world, join=True)
    assert len(ret) == world
    for r in range(world):
        assert ret[r][0] < 1e-4, (r, ret[r][0])
        assert torch.equal(ret[r][1], ret[0][1])                              # identical on every rank
