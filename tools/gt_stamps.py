"""Cycle stamps of k_scores_tile_p (wave 0 of workgroup 0, its first tiles): where a tile's time goes.
Build (CPU container): python tools/gt_stamps.py build  -> vlsa_amd/_lib/variants/libvlsa_gtstamp.so (-DVLSA_GT_STAMP)
Run (GPU box):          python tools/gt_stamps.py [N] [gated|ungated]"""
import ctypes, glob, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIBD = os.path.join(ROOT, "vlsa_amd", "_lib")
LIB = os.path.join(LIBD, "variants", "libvlsa_gtstamp.so")
if len(sys.argv) > 1 and sys.argv[1] == "build":
    os.makedirs(os.path.join(LIBD, "variants"), exist_ok=True)
    objs = [o for o in glob.glob(os.path.join(LIBD, "obj", "*.o")) if not o.endswith("gated_scores_tile.o")]
    subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-DVLSA_GT_STAMP", *os.environ.get("GT_EXTRA", "").split(), "-c",
                           os.path.join(ROOT, "vlsa_amd", "csrc", "gated_scores_tile.hip"), "-o", "/tmp/gt_stamp.o"])
    subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", *objs, "/tmp/gt_stamp.o", "-o", LIB])
    sys.exit(0)
os.environ["VLSA_HIP_LIB"] = LIB
sys.path.insert(0, ROOT)
import torch
from vlsa_amd import functional as F, _native as nat
n = int(sys.argv[1]) if len(sys.argv) > 1 else 393216
gated = not (len(sys.argv) > 2 and sys.argv[2] == "ungated")
pool = len(sys.argv) > 3 and sys.argv[3] == "pool"            # scores + pooling in one launch (k_scores_tile_p<.., POOL>)
dev = "cuda"
Wa = torch.randn(256, 512, device=dev) / 22; ba = torch.randn(256, device=dev) * 0.05
Wg = torch.randn(256, 512, device=dev) / 22 if gated else None; bg = torch.randn(256, device=dev) * 0.05 if gated else None
w2 = torch.randn(1, 256, device=dev) / 16; c = torch.randn(1, device=dev)
X = torch.randn(n, 512, device=dev).to(torch.bfloat16)
fs = F.FusedAttnScores()
for _ in range(20):
    if pool: fs.scores_and_pool(X, Wa, ba, Wg, bg, w2, c)
    else: fs(X, Wa, ba, Wg, bg, w2, c)
torch.cuda.synchronize()
lib = nat.load()
buf = (ctypes.c_longlong * 256)()
lib.vlsa_debug_gt_stamps.argtypes = [ctypes.c_void_p]
assert lib.vlsa_debug_gt_stamps(buf) == 0
st = list(buf)
t0 = st[0]
print("prologue: kernel start -> tile 0 start: %d cycles" % (st[1] - st[0]))
k = 1
for tile in range(4):
    if st[k] == 0: break
    base = st[k]
    print(f"tile {tile}: start +{base - t0}: wait for steps 0-2 {st[k+1]-st[k]}, barrier {st[k+2]-st[k+1]}")
    k += 3
    prev = st[k - 1]
    row = []
    for s in range(16):
        row.append(f"s{s}: half {st[k]-prev} wait {st[k+1]-st[k]} bar {st[k+2]-st[k+1]}")
        prev = st[k + 2]; k += 3
    print("   " + " | ".join(row))
    if pool and (gated is False or True):
        # (gated + pool: two passes per tile; the stamps below are those of ONE pass, the pooling stamps exist behind the second)
        print(f"   second half of step 15: {st[k]-prev}; stamps behind the K loop (deltas): {[st[k+i+1]-st[k+i] for i in range(3)]}")
        k += 4 if not gated else 2
        continue
    print(f"   second half of step 15: {st[k]-prev}; epilogue {st[k+1]-st[k]}; tile total {st[k+1]-base}")
    k += 2

