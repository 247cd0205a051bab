"""Shader-cycle stamps of workgroup 0 / thread 0 of the single-bag streaming kernel k_vlfan_partial_dma (-DVLSA_TIMING): where the
~18 us of one 50k-patch slide go.  Build (CPU container): python tools/dma_stamps.py build; run on the GPU box: python tools/dma_stamps.py"""
import ctypes, glob, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIBD = os.path.join(ROOT, "vlsa_amd", "_lib")
LIB = os.path.join(LIBD, "variants", "libvlsa_dmatiming.so")
if len(sys.argv) > 1 and sys.argv[1] == "build":
    os.makedirs(os.path.join(LIBD, "variants"), exist_ok=True)
    objs = [o for o in glob.glob(os.path.join(LIBD, "obj", "*.o")) if not o.endswith("vlfan_partial_dma.o")]
    subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-DVLSA_TIMING", "-c",
                           os.path.join(ROOT, "vlsa_amd", "csrc", "vlfan_partial_dma.hip"), "-o", "/tmp/dma_timing.o"])
    subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", *objs, "/tmp/dma_timing.o", "-o", LIB])
    sys.exit(0)
os.environ["VLSA_HIP_LIB"] = LIB
sys.path.insert(0, ROOT)
import torch
from vlsa_amd import _native as nat
from vlsa_amd.vlsa import VLSA
dev = "cuda"
N = int(sys.argv[1]) if len(sys.argv) > 1 else 50000
cfg = dict(name="VLFAN", dim_in=512, use_feat_proj=False, query="Parameter", num_query=12, query_pooling="mean")
net = VLSA.from_modules(cfg, pretrained_text_features=torch.randn(4, 512)).to(dev).eval()
bags = [torch.randn(N, 512, device=dev).to(torch.bfloat16)[None] for _ in range(8)]
lib = nat.load()
lib.vlsa_debug_read_cycles.argtypes = [ctypes.c_void_p]
buf = (ctypes.c_longlong * 32)()
names = ["kernel start", "setup done, first DMA about to be issued", "query fragments in registers", "first tile landed", "scores of the first tile",
         "exchange done", "first tile's weighted sum", "loop done", "row groups merged", "partial stored", "stores drained"]
with torch.no_grad():
    for i in range(40):
        net(bags[i % 8])
    torch.cuda.synchronize()
    rows = []
    for i in range(8):
        net(bags[i % 8]); torch.cuda.synchronize()
        assert lib.vlsa_debug_read_cycles(buf) == 0
        rows.append([buf[k] - buf[0] for k in range(11)])
for k in range(11):
    v = sorted(r[k] for r in rows)
    print(f"{names[k]:45s}: +{v[len(v) // 2]:7d} cycles (median of 8 calls; min {v[0]}, max {v[-1]})")
