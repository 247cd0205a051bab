"""Per-kernel timing of the VLFAN forward pieces (development tool; bench.py is the contract benchmark)."""
import argparse
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from vlsa_amd import functional as F


def timeit(fn, iters, warm=5):
    for _ in range(warm):
        fn(0)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(iters):
        fn(i)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters  # us


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=50000)
    ap.add_argument("--dtype", default="bf16")
    ap.add_argument("--p", type=int, default=12)
    ap.add_argument("--k", type=int, default=4)
    ap.add_argument("--bags", type=int, default=8)
    ap.add_argument("--iters", type=int, default=200)
    a = ap.parse_args()
    dt = torch.bfloat16 if a.dtype == "bf16" else torch.float32
    dev = "cuda"
    g = torch.Generator(device=dev).manual_seed(0)
    bags = [torch.randn(a.n, 512, device=dev, generator=g).to(dt) for _ in range(a.bags)]
    Q = torch.randn(a.p, 512, device=dev, generator=g)
    T = torch.randn(a.k, 512, device=dev, generator=g)
    W = torch.randn(512, 512, device=dev, generator=g) / 22.6
    b = torch.randn(512, device=dev, generator=g) / 22.6
    ls = torch.tensor(4.0309, device=dev)
    qp = F.prepare_queries(Q)
    That, _ = F.normalize_rows(T)
    nbytes = a.n * 512 * bags[0].element_size()
    print(f"N={a.n} dtype={a.dtype} P={a.p} bag={nbytes/1e6:.1f} MB partials={F.num_partials(a.n)}")
    for name, kern in (("dma", 3), ("mfma", 2)):
        us = timeit(lambda i: F.vlfan_partial(bags[i % a.bags], qp, kernel=kern), a.iters)
        print(f"partial[{name}] (rotating {a.bags} bags, incl. python+alloc): {us:8.2f} us  {nbytes/us/1e3:8.1f} GB/s")
        us = timeit(lambda i: F.vlfan_partial(bags[0], qp, kernel=kern), a.iters)
        print(f"partial[{name}] (same bag, L3-resident):               {us:8.2f} us  {nbytes/us/1e3:8.1f} GB/s")
    pm, pl, pacc, _ = F.vlfan_partial(bags[0], qp)
    us = timeit(lambda i: F.vlfan_merge(pm, pl, pacc), a.iters)
    print(f"merge: {us:8.2f} us")
    m2, l, out = F.vlfan_merge(pm, pl, pacc)
    us = timeit(lambda i: F.head_forward(out, "mean", None, W, b, That, ls), a.iters)
    print(f"head:  {us:8.2f} us")
    us = timeit(lambda i: F.prepare_queries(Q), a.iters)
    print(f"prepare_queries: {us:8.2f} us")

    def full(i):
        pm, pl, pacc, _ = F.vlfan_partial(bags[i % a.bags], qp)
        m2, l, out = F.vlfan_merge(pm, pl, pacc)
        return F.head_forward(out, "mean", None, W, b, That, ls)
    us = timeit(full, a.iters)
    print(f"full forward (eager python): {us:8.2f} us  -> {a.n/us:8.1f} M patches/s")
    # graph-captured
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for i in range(3):
            full(i)
    torch.cuda.current_stream().wait_stream(s)
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        for i in range(a.bags):
            full(i)
    us = timeit(lambda i: gr.replay(), max(10, a.iters // a.bags)) / a.bags
    print(f"full forward (hipGraph of {a.bags} bags): {us:8.2f} us/bag -> {a.n/us:8.1f} M patches/s  {nbytes/us/1e3:8.1f} GB/s")


if __name__ == "__main__":
    main()
