import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..")); sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "tests")); sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "tests", "golden"))
import torch, torch.distributed as dist, torch.multiprocessing as mp
import cases

def worker(rank, world, port):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from vlsa_amd import functional as F
    from vlsa_amd.sharded import ShardedVlfanBatchPlan, shard_bounds
    dev = torch.device("cuda", 0); torch.cuda.set_device(dev)
    P, K = 12, 4
    params = cases.make_params(P, K, 9200)
    Q = (0.5 * params["resid"] + params["prompt"]).to(dev)
    T, W, b = params["T"].to(dev), params["W"].to(dev), params["b"].to(dev)
    ls = torch.tensor(cases.LOGIT_SCALE, device=dev)
    B = 2 * world + 1
    sizes = [3000 + 517 * i if i % 3 else 64 + i for i in range(B)]
    full = [cases.make_bag(n, 9300 + i).to(torch.bfloat16).to(dev) for i, n in enumerate(sizes)]
    bnd = [shard_bounds(n, world, rank) for n in sizes]
    cut = [x[a:c] for x, (a, c) in zip(full, bnd)]
    rp = F.VlfanBatchPlan(B, P, K, dev); rp.set_bags(full); rp.run(Q, T, ls, W, b)
    want = [t.clone() for t in (rp.logits, rp.incidence, rp.vhat, rp.m2, rp.l)]
    for ex in ("allgather", "owner", "ipc"):
        bp = ShardedVlfanBatchPlan(B, P, K, dev, dist, pipeline=False, exchange=ex, timeout_s=5)
        bp.set_bags(cut)
        for it in range(2):
            bp.run(Q, T, ls, W, b)
            torch.cuda.synchronize()
            got = (bp.logits, bp.incidence, bp.vhat, bp.m2, bp.l)
            if rank == 0:
                print(ex, it, [float((g[:, :P] - w[:, :P]).abs().max()) if n in ("m2", "l") else float((g - w).abs().max())
                               for n, g, w in zip(("logits", "inc", "vhat", "m2", "l"), got, want)], "status", bp.status(), flush=True)
                if it == 1 and ex != "allgather":
                    print(" per-bag logit err", [round(float((bp.logits[i] - want[0][i]).abs().max()), 5) for i in range(B)], bp.counts, bp.perm, flush=True)
        bp.close()
    dist.destroy_process_group()

if __name__ == "__main__":
    w = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    mp.spawn(worker, args=(w, 29333), nprocs=w, join=True)
