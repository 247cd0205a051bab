import os, ctypes, torch
lib = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "probes/libmfma_rate.so"))
out = torch.zeros(256 * 512, device="cuda")
iters = 2000
for threads in (256, 512):
    for ldsr in (0, 1):
        for _ in range(3): lib.mfma_rate_launch(ctypes.c_void_p(out.data_ptr()), 256, threads, iters, ldsr, None)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5): lib.mfma_rate_launch(ctypes.c_void_p(out.data_ptr()), 256, threads, iters, ldsr, None)
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / 5
        n_mfma_per_simd = iters * 32 * (threads // 256)       # 32 MFMAs per iteration per wave (4 q x 2 t x 4 r x ... see kernel)
        waves = threads // 64
        total = 256 * waves * iters * 64
        print(f"threads={threads} ldsreads={ldsr}: {us:8.1f} us  {total * 16384 / us / 1e6:8.1f} TFLOP/s  ({us * 1e-6 * 2.4e9 / (iters * 64 * (threads // 256)):5.1f} cycles@2.4GHz per MFMA per SIMD)")
