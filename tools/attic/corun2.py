import sys, os, time, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from vlsa_amd import functional as F
lib = ctypes.CDLL(os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tools/probes/libcorun_probe.so"))
dev = "cuda"
B, n = 32, 50000
bags = [torch.randn(n, 512, device=dev).to(torch.bfloat16) for _ in range(B)]
Q = torch.randn(12, 512, device=dev); T = torch.randn(4, 512, device=dev)
W = torch.randn(512, 512, device=dev) / 22; b = torch.randn(512, device=dev); ls = torch.tensor(4.03, device=dev)
plan = F.VlfanBatchPlan(B, 12, 4, dev); plan.set_bags(bags); plan.run(Q, T, ls, W, b)
out = torch.empty(4096 * 256, device=dev)
sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
torch.cuda.synchronize()
for lds in (1024,):
    for hireg in (0, 24, 32, 48, 64, 80, 96, 128):
        for blocks in (384,):
            ea, eb, e0 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            with torch.cuda.stream(sa):
                e0.record(sa)
                plan.run_partial_only()
                ea.record(sa)
            time.sleep(0.00008)
            with torch.cuda.stream(sb):
                lib.probe_launch(ctypes.c_void_p(out.data_ptr()), blocks, lds, hireg, 2000, ctypes.c_void_p(sb.cuda_stream))
                eb.record(sb)
            torch.cuda.synchronize()
            print(f"lds={lds:6d} hireg={hireg} blocks={blocks}: persistent done at {e0.elapsed_time(ea)*1e3:7.1f} us, probe done at {e0.elapsed_time(eb)*1e3:7.1f} us")
