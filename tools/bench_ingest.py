"""PCIe-inclusive ingest rate: host fp32 bags -> resident bf16 arena (both conversion modes) vs the reference's blocking
per-bag fp32 .to(device).  The GPU boxes are shared hosts: every configuration is timed 5 times, best and median shown."""
import sys, os, time, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vlsa_amd.ingest import DeviceBagArena
dev = torch.device("cuda", 0)
n, B = 50_000, 8
host = [torch.randn(n, 512) for _ in range(B)]
def trials(f, k=5):
    f()
    ts = []
    for _ in range(k):
        torch.cuda.synchronize(); t0 = time.perf_counter(); f(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    return min(ts), statistics.median(ts)
def ref():
    for x in host:
        y = x.to(dev)          # what runner/vlsa_handler.py:205 does every step (pageable memory, blocking)
best, med = trials(ref)
print(f"reference-style blocking fp32 .to(device): best {best / B * 1e3:6.2f} ms/bag ({B * n / best / 1e6:6.1f} M patches/s, "
      f"{B * n * 2048 / best / 1e9:5.1f} GB/s)  median {med / B * 1e3:6.2f} ms/bag")
for threads in (16, 4):
    for convert in ("device", "host"):
        for chunk in (8192, 32768):
            arena = DeviceBagArena((B + 1) * 50_048, dev, chunk_rows=chunk, convert=convert, host_threads=threads)
            def up():
                for i, x in enumerate(host):
                    arena.add(i, x)
                arena.reset()
            best, med = trials(up)
            per_el = 4 if convert == "device" else 2
            print(f"arena threads={threads:3d} convert={convert:6s} chunk={chunk:6d}: best {best / B * 1e3:6.2f} ms/bag "
                  f"({B * n / best / 1e6:6.1f} M patches/s, PCIe {B * n * 512 * per_el / best / 1e9:5.1f} GB/s)  median {med / B * 1e3:6.2f} ms/bag")
            del arena
