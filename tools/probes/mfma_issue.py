"""Shader cycles per MFMA on one SIMD (tools/probes/mfma_issue.hip).  Build here:
hipcc --offload-arch=gfx950 -O3 -shared -fPIC tools/probes/mfma_issue.hip -o tools/probes/libmfma_issue.so"""
import ctypes, os, torch
here = os.path.dirname(os.path.abspath(__file__))
lib = ctypes.CDLL(os.path.join(here, "libmfma_issue.so"))
lib.mfma_issue_launch.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_float, ctypes.c_void_p]
out = torch.zeros(1024, dtype=torch.int64, device="cuda"); sink = torch.zeros(1024 * 512, device="cuda")
names = {0: "plain", 1: "s_setprio around groups of 16", 2: "s_barrier per 64", 3: "16 ds_read_b128 per 64"}
for seed, what in ((0.0, "zero operands"), (0.37, "non-zero operands")):
    for blocks in (1, 256):
        for threads in (256, 512):
            for mode in (0, 1, 2, 3):
                iters = 2000
                for _ in range(2):
                    lib.mfma_issue_launch(out.data_ptr(), sink.data_ptr(), blocks, threads, iters, mode, seed, None)
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(); lib.mfma_issue_launch(out.data_ptr(), sink.data_ptr(), blocks, threads, iters, mode, seed, None); e1.record()
                torch.cuda.synchronize()
                cyc = out[:blocks].float().mean().item()
                per = cyc / (iters * 64 * (threads // 256))
                us = e0.elapsed_time(e1) * 1e3
                print(f"{what:18s} blocks {blocks:3d} waves/SIMD {threads // 256}: {names[mode]:32s} {per:6.2f} cycles per MFMA and SIMD; kernel {us:8.1f} us -> {cyc / us / 1e3:5.2f} GHz")
