#!/usr/bin/env python
"""Contract benchmark: patches/s of the per-slide VLSA forward (language-guided patch aggregation).

    python bench.py --gpus N --steps K --warmup W          (N > 1: launched by torch.distributed.run)

Workload (BASELINE.json): synthetic 50k x 512 bf16 CONCH bag, P = 12 text-prototype queries, K = 4 ordinal
rank prompts, mean query pooling, Linear(512,512) visual adapter -- `configs[2]`, the configuration the
metric is quoted on.  A step = one bag through query/text normalisation, the streaming aggregation kernel,
the partial merge and the incidence head (= VLSA.forward in eval mode with cached text features,
reference model/vlsa.py:181-198).  Bags are resident in HBM before the timed region; 8 distinct bags are
rotated (410 MB > the 256 MiB Infinity Cache) so the stream really comes from HBM.

N > 1: the bag is N x 50k patches, patch-sharded across the ranks (weak scaling: 50k rows per GPU); each
rank streams its shard, the ranks all-gather their compact (m, l, acc[P,512]) partials over RCCL, every
rank merges and runs the replicated head.  value = whole-job patches/s.
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

N_PER_GPU = 50_000
D, P, K = 512, 12, 4
N_BAGS = 8
HBM_PEAK_GBPS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.3 TB/s achievable)


def synth(device, seed):
    g = torch.Generator(device=device).manual_seed(seed)
    bags = [torch.randn(N_PER_GPU, D, device=device, generator=g).to(torch.bfloat16) for _ in range(N_BAGS)]
    gq = torch.Generator(device=device).manual_seed(1234)  # parameters identical on every rank
    Q = 0.5 * torch.randn(P, D, device=device, generator=gq) + torch.randn(P, D, device=device, generator=gq)
    T = torch.randn(K, D, device=device, generator=gq)
    W = (torch.rand(D, D, device=device, generator=gq) * 2 - 1) / D ** 0.5
    b = (torch.rand(D, device=device, generator=gq) * 2 - 1) / D ** 0.5
    ls = torch.tensor(4.0309, device=device)
    return bags, Q, T, W, b, ls


def cpu_baseline(seconds=10.0):
    """The CPU oracle (restatement of the reference's torch op sequence, pinned to the reference by
    tests/golden) timed on this host's cores on the same workload: kind = "port"."""
    from oracle import vlsa_oracle as O
    ncpu = os.cpu_count() or 1
    g = torch.Generator().manual_seed(7)
    X = torch.randn(N_PER_GPU, D, generator=g).to(torch.bfloat16).float()
    Q = torch.randn(P, D, generator=g)
    T = torch.randn(K, D, generator=g)
    W = torch.randn(D, D, generator=g) / D ** 0.5
    b = torch.randn(D, generator=g) / D ** 0.5
    ls = torch.tensor(4.0309)
    with torch.no_grad():
        # torch CPU kernels stop scaling (and then regress) well below the core count of a GPU host: pick the
        # fastest thread count from a short calibration and report THAT many cores.
        best = (float("inf"), 1)
        for th in sorted({1, 8, 16, 32, 64, ncpu} & set(range(1, ncpu + 1))):
            torch.set_num_threads(th)
            O.vlsa_vlfan_forward(X, Q, T, ls, head_weight=W, head_bias=b)
            t0 = time.perf_counter()
            O.vlsa_vlfan_forward(X, Q, T, ls, head_weight=W, head_bias=b)
            best = min(best, (time.perf_counter() - t0, th))
        cores = best[1]
        torch.set_num_threads(cores)
        n, t0 = 0, time.perf_counter()
        while time.perf_counter() - t0 < seconds:
            O.vlsa_vlfan_forward(X, Q, T, ls, head_weight=W, head_bias=b)
            n += 1
        dt = time.perf_counter() - t0
    return {"value": N_PER_GPU * n / dt, "unit": "patches/s", "cores": cores, "kind": "port",
            "sample": f"{n} bags of 50000x512 (fp32 math on bf16-rounded values) in {dt:.1f} s, torch {torch.__version__} CPU, "
                      f"best of 1/8/16/32/64/{ncpu} threads"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=400)
    ap.add_argument("--warmup", type=int, default=40)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-graph", action="store_true", help="eager launches instead of hipGraph replay")
    a = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        if world == 1 and a.gpus > 1:
            sys.exit("launch with: python -m torch.distributed.run --nnodes=1 --nproc-per-node N bench.py --gpus N ...")
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    from vlsa_amd import functional as F

    dist = None
    force_sharded = os.environ.get("VLSA_BENCH_FORCE_SHARDED") == "1"  # exercise the N > 1 code path on one GPU
    if world > 1 or force_sharded:
        import torch.distributed as dist
        if force_sharded and "RANK" not in os.environ:
            os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29677", RANK="0", WORLD_SIZE="1")
        dist.init_process_group("nccl", device_id=device)

    bags, Q, T, W, b, ls = synth(device, 100 + rank)
    if world == 1 and not force_sharded:
        plan = F.VlfanInferencePlan(N_PER_GPU, D, P, K, device)
        step = lambda i: plan.run(bags[i % N_BAGS], Q, T, ls, W, b)  # noqa: E731
    else:
        from vlsa_amd.sharded import ShardedVlfanPlan
        plan = ShardedVlfanPlan(N_PER_GPU, D, P, K, device, dist)
        step = lambda i: plan.run(bags[i % N_BAGS], Q, T, ls, W, b)  # noqa: E731

    def sync():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    # ---- warmup (also JIT-free: everything is precompiled) ------------------------------------------------
    for i in range(a.warmup):
        step(i)
    sync()

    use_graph = (not a.no_graph) and world == 1 and not force_sharded
    if use_graph:
        chunk = min(a.steps, 64)
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        graphs = {}
        with torch.cuda.stream(s):
            for n in {chunk, a.steps % chunk} - {0}:
                gr = torch.cuda.CUDAGraph()
                with torch.cuda.graph(gr, stream=s):
                    for i in range(n):
                        step(i)
                graphs[n] = gr
        torch.cuda.current_stream().wait_stream(s)
        for gr in graphs.values():  # one untimed replay each
            gr.replay()
        sync()

    sync()
    t0 = time.perf_counter()
    if use_graph:
        for _ in range(a.steps // chunk):
            graphs[chunk].replay()
        if a.steps % chunk:
            graphs[a.steps % chunk].replay()
    else:
        for i in range(a.steps):
            step(i)
    if hasattr(plan, "finish"):
        plan.finish()
    sync()
    dt = time.perf_counter() - t0
    if dist is not None:
        tt = torch.tensor([dt], device=device, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())

    # ---- roofline of the dominant kernel: HIP events around each launch on the launching stream ----------
    roof = None
    if rank == 0:
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(100)]
        base = plan.local if hasattr(plan, "local") else plan
        for i in range(10):
            base.run_partial_only(bags[i % N_BAGS])
        torch.cuda.synchronize()
        for i, (e0, e1) in enumerate(ev):
            e0.record()
            base.run_partial_only(bags[i % N_BAGS])
            e1.record()
        torch.cuda.synchronize()
        ts = sorted(e0.elapsed_time(e1) for e0, e1 in ev)
        # an event pair around NOTHING measures the pair's own cost on this stream; subtract it so the figure is the
        # kernel's duration (what rocprofv3 --kernel-trace reports), not duration + event overhead
        for e0, e1 in ev:
            e0.record()
            e1.record()
        torch.cuda.synchronize()
        null_ms = sorted(e0.elapsed_time(e1) for e0, e1 in ev)[len(ev) // 2]
        ts = [max(t - null_ms, 0.0) for t in ts]
        avg_ms = sum(ts) / len(ts)
        algo_bytes = N_PER_GPU * D * 2  # 1024 B per bf16 patch row
        ach = algo_bytes / (avg_ms * 1e-3) / 1e9
        roof = {"bound": "hbm", "kernel": "k_vlfan_partial_dma<false> (bf16 rows, D=512)", "achieved": round(ach, 1),
                "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBPS, 4), "traffic": None,
                "avg_us": round(avg_ms * 1e3, 2), "min_us": round(ts[0] * 1e3, 2), "event_pair_us": round(null_ms * 1e3, 2),
                "bytes_per_launch": algo_bytes}
    if dist is not None:
        dist.barrier()

    if rank == 0:
        total_patches = N_PER_GPU * world * a.steps
        out = {
            "metric": "patches/sec per slide (50k x 512 CONCH bag)", "value": total_patches / dt, "unit": "patches/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": dt / a.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
            "config": {"workload": "configs[2]: synthetic 50k x 512 bf16 bag per GPU, P=12 queries, K=4 rank prompts, "
                                   "mean pooling + Linear(512,512) head; N GPUs = one N*50k-patch bag patch-sharded",
                       "rows_per_gpu": N_PER_GPU, "D": D, "P": P, "K": K, "bags_rotated": N_BAGS,
                       "launch": "hipGraph replay" if use_graph else "eager"},
            "roofline": roof,
        }
        if not a.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline()
        try:  # flush anything native libraries (RCCL banner) left in the C stdio buffer, so the JSON is the last line
            import ctypes
            ctypes.CDLL(None).fflush(None)
        except Exception:
            pass
        sys.stdout.flush()
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
