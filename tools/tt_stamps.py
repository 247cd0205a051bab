"""Cycle stamps of the LayerNorm-fused text GEMM (build with VLSA_EXTRA_HIPCC_FLAGS=-DVLSA_TT_DEBUG): the last PRO_LN launch of a
forward pass leaves, for workgroup 5 / thread 0: 0 start, 1 loads issued, 2 partial sums, 3 mean, 4 rstd, 5 slab normalised,
6 MFMA loop done, 7 reduction written + barrier, 8 end (100 MHz-independent s_memtime-class counter: shader clocks)."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import torch
import text_cases as TC
import text_helpers as TH
from test_text_modules_cpu import build_learner
from test_gpu_text_tower import build_encoder
from vlsa_amd import _native as nat
case = TC.RANK_CASES[0]
inp = TH.rank_case_inputs(case)
enc = build_encoder(case[1], case[2])
pl = build_learner(case, inp).cuda()
PREFIX = 0 if "--no-prefix" in sys.argv else pl.shared_prefix_len
lib = nat.load()
fn = lib._lib.vlsa_tt_debug_stamps if hasattr(lib, "_lib") else None
import glob
raw = ctypes.CDLL(glob.glob(os.path.join(ROOT, "vlsa_amd", "_lib", "libvlsa_hip.so"))[0])
for it in range(6):
    with torch.no_grad():
        enc(prompts_embedding=pl(), prompts_pseudo_tokens=pl.pseudo_sentence_tokens, shared_prefix_len=PREFIX)
    torch.cuda.synchronize()
    buf = (ctypes.c_longlong * 16)()
    raw.vlsa_tt_debug_stamps(buf)
    v = list(buf)[:9]
    print("stamps (cycles since start):", [x - v[0] for x in v])
