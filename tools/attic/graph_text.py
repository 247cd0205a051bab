"""Does a hipGraph replay of the text tower's ~65 forward launches beat the eager launch sequence?  (tower alone, no grad)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import torch
import text_cases as TC
import text_helpers as TH
from test_text_modules_cpu import build_learner
from test_gpu_text_tower import build_encoder
case = TC.RANK_CASES[0]
inp = TH.rank_case_inputs(case)
enc = build_encoder(case[1], case[2])
pl = build_learner(case, inp).cuda()
with torch.no_grad():
    sent = pl()
    fn = lambda: enc(prompts_embedding=sent, prompts_pseudo_tokens=pl.pseudo_sentence_tokens)
    for _ in range(10): ref = fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(50): fn()
    torch.cuda.synchronize(); te = (time.perf_counter() - t0) / 50
    side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(3): fn()
    torch.cuda.current_stream().wait_stream(side)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        out = fn()
    for _ in range(10): g.replay()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(50): g.replay()
    torch.cuda.synchronize(); tg = (time.perf_counter() - t0) / 50
    print(f"tower forward eager {te * 1e6:.0f} us, hipGraph replay {tg * 1e6:.0f} us, max |diff| {(out - ref).abs().max().item():.2e}")
