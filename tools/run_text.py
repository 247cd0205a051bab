"""N forward passes (no grad) of the CONCH-size text tower over the K = 12 rank prompts -- the command the `--pmc FETCH_SIZE` pass of
tools/collect_profiles.sh profiles; `python tools/run_text.py summarise <dir>` then prints the traffic per pass: FETCH_SIZE / WRITE_SIZE
summed over every k_tt_* launch of a pass (units and the gfx950 x 2 on FETCH_SIZE as /opt/skills/guides/MI355X_MICROARCH.md prescribes
and profiles/r0*_pmc_batch_kernel*.json use) against the 340 MB weight stream a pass cannot avoid."""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PASSES, WARM = 10, 3
if len(sys.argv) > 2 and sys.argv[1] == "summarise":
    import collections, csv, glob, json
    import re
    tot, n = collections.defaultdict(float), collections.defaultdict(int)
    per = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(sys.argv[2] + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            name = r["Kernel_Name"]
            if "k_tt_" in name and "k_tt_pack" not in name:       # (the one-time packing of the weights is not part of a pass)
                tot[r["Counter_Name"]] += float(r["Counter_Value"])
                n[r["Counter_Name"]] += 1
                short = re.sub(r"\(.*", "", name).replace("void ", "").replace("vlsa::tt::", "")
                per[short][r["Counter_Name"]].append(float(r["Counter_Value"]))
    out = {"passes": PASSES + WARM, "launches_per_pass": {k: v / (PASSES + WARM) for k, v in n.items()}}
    out["per_kernel_MB_per_launch"] = {k: {"launches_per_pass": len(next(iter(v.values()))) / (PASSES + WARM),
                                           **{c: round(sum(x) / len(x) * 1024 * (2 if c == "FETCH_SIZE" else 1) / 1e6, 2) for c, x in v.items()}}
                                       for k, v in sorted(per.items())}
    if "FETCH_SIZE" in tot:
        out["fetch_MB_per_pass"] = round(tot["FETCH_SIZE"] * 1024 * 2 / (PASSES + WARM) / 1e6, 1)       # KB, x 2 on gfx950
    if "WRITE_SIZE" in tot:
        out["write_MB_per_pass"] = round(tot["WRITE_SIZE"] * 1024 / (PASSES + WARM) / 1e6, 1)
    out["weights_MB"] = round((12 * 12 * 768 * 768 + 768 * 512) * 4 / 1e6, 1)
    print(json.dumps(out))
    sys.exit(0)
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
if "--bwd" in sys.argv:              # forward + backward passes (d prompts; the tower frozen)
    for _ in range(PASSES + WARM):
        pl.zero_grad(set_to_none=True)
        enc(prompts_embedding=pl(), prompts_pseudo_tokens=pl.pseudo_sentence_tokens, shared_prefix_len=pl.shared_prefix_len).sum().backward()
else:
    with torch.no_grad():
        for _ in range(PASSES + WARM):
            enc(prompts_embedding=pl(), prompts_pseudo_tokens=pl.pseudo_sentence_tokens, shared_prefix_len=pl.shared_prefix_len)
torch.cuda.synchronize()
