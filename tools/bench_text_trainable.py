import os, sys, time
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import torch
import text_cases as TC, text_helpers as TH
from test_text_modules_cpu import build_learner
from test_gpu_text_tower import build_encoder
case = TC.RANK_CASES[0]
inp = TH.rank_case_inputs(case)
enc = build_encoder(case[1], case[2])
for p in enc.parameters(): p.requires_grad_(True)
enc.token_embedding.weight.requires_grad_(False)
pl = build_learner(case, inp).cuda()
def fb():
    enc.zero_grad(set_to_none=True); pl.zero_grad(set_to_none=True)
    f = enc(prompts_embedding=pl(), prompts_pseudo_tokens=pl.pseudo_sentence_tokens, shared_prefix_len=pl.shared_prefix_len)
    f.sum().backward()
for _ in range(5): fb()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(20): fb()
torch.cuda.synchronize()
print(f"trainable tower (K = {case[3]} prompts, compact rows + shared prefix; HIP forward + backward incl. ALL weight gradients, vlsa_tt_backward_train): {(time.perf_counter()-t0)/20*1e3:.2f} ms   (torch route until round 4: 14.4 ms)")
