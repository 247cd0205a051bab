"""Where a task of the persistent text-tower forward (k_tt_forward_persistent) spends its time: shader-clock stamps of ONE workgroup
(VLSA_TT_STAMPS=<workgroup>) at the six phases of each of its tasks -- start | dependency met | operands landed | arithmetic done |
stores issued | drained + counted in -- averaged over blocks 1..11 of a CONCH-size forward over the K = 12 rank prompts."""
import os
import sys
import time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import ctypes
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
L = pl.shared_prefix_len
with torch.no_grad():
    sent = pl()
run = lambda: enc(prompts_embedding=sent, prompts_pseudo_tokens=pl.pseudo_sentence_tokens, shared_prefix_len=L)   # noqa: E731
os.environ["VLSA_TT_PERSIST"] = "1"
names = ["QKV (ln_1)", "attention", "out-proj", "c_fc (ln_2)", "c_proj"]
phases = ["wait", "operands", "arithmetic", "reduce+stores", "drain+count"]
with torch.no_grad():
    for _ in range(20):
        run()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(50):
        run()
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / 50 * 1e6
    plan = enc._plan(pl.pseudo_sentence_tokens, sent.device, L)
    lib = nat.load()
    off = lib.vlsa_tt_status_offset(ctypes.byref(enc._c_model(sent.device)), ctypes.byref(plan.c), 0)
    assert off >= 0
    print(f"persistent forward: {wall:.0f} us per call (stamps off)")
    for wg in (0, 37, 100, 155, 167, 200, 223, 251):
        os.environ["VLSA_TT_STAMPS"] = str(wg)
        for _ in range(3):
            run()
        torch.cuda.synchronize()
        base = plan.ws.data_ptr() + off + 32
        pad = (-base) % 8
        st = plan.ws[off + 32 + pad: off + 32 + pad + 12 * 5 * 6 * 8].view(torch.int64).cpu().view(12, 5, 6).double()
        span = float(st[11].max() - st[0][0][0])
        print(f"workgroup {wg}: first stamp -> last stamp {span:.0f} ticks")
        for s_ in range(5):
            t = st[1:, s_, :]
            if float(t.abs().max()) == 0:
                continue
            d = (t[:, 1:] - t[:, :-1]).mean(0)
            tot = float((t[:, 5] - t[:, 0]).mean())
            print(f"   {names[s_]:12s} task {tot:8.0f} ticks: " + "  ".join(f"{p} {float(x):7.0f}" for p, x in zip(phases, d)))
        # idle between this workgroup's tasks does not exist (a task starts when the one before has counted in): block period
        per = float((st[1:, 0, 0] - st[:-1, 0, 0]).mean())
        print(f"   block period {per:.0f} ticks  (x 12 = {12 * per:.0f})")
    os.environ.pop("VLSA_TT_STAMPS")
