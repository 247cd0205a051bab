"""Randomised cross-check of the round-5 tails: the batched plan (k_vlfan_merge_pool_small / _batch, k_head_linear_mfma / k_head_linear,
k_head_finish) and the single-slide plan (k_vlfan_merge_wpart + k_head_finish_parts, or merge + ticketed head) against a float64 torch
tail computed from the single-bag aggregation kernels' rows: random B (1 .. 256), P, K, pooling (mean / max / weight), Linear or identity
adapter, gated queries, both dtypes, ragged bag sizes.  python tools/fuzz_tails.py [rounds] [seed]"""
import os, sys, random
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vlsa_amd import functional as F

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 60
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
rng = random.Random(seed)
dev = torch.device("cuda", 0)
g = torch.Generator().manual_seed(seed)
pool = torch.randn(80_000, 512, generator=g)
pool[::5] *= 2.5
pool_d = {torch.bfloat16: pool.to(torch.bfloat16).to(dev), torch.float32: pool.to(dev)}
worst = {"batch logits": 0.0, "batch vhat": 0.0, "batch incidence": 0.0, "single logits": 0.0, "single vhat": 0.0}


def tail64(rows, mode, pw, W, b, T, ls):
    rows = rows.double()
    if mode == "mean":
        pooled = rows.mean(dim=-2)
    elif mode == "max":
        pooled = rows.max(dim=-2).values
    else:
        pooled = (torch.softmax(pw.double(), 0)[:, None] * rows).sum(dim=-2)
    v = pooled if W is None else pooled @ W.double().t() + b.double()
    vh = v / v.norm(dim=-1, keepdim=True).clamp_min(1e-12)
    Th = T.double() / T.double().norm(dim=-1, keepdim=True).clamp_min(1e-12)
    logits = ls.double().exp() * vh @ Th.t()
    return logits, vh, torch.softmax(logits, dim=-1)


for it in range(rounds):
    dt = rng.choice([torch.bfloat16, torch.float32])
    B = rng.choice([1, 2, 7, 15, 16, 17, 31, 33, 64, 65, 100, 128, 200, 256])
    P = rng.choice([1, 4, 7, 12, 12, 13, 16])
    K = rng.choice([1, 4, 5, 8, 12, 33, 64])
    mode = rng.choice(["mean", "mean", "max", "weight"])
    ident = rng.random() < 0.2
    gated = rng.random() < 0.3
    big = B <= 33 and rng.random() < 0.3
    sizes = [rng.randint(20_000, 50_000) if (big and rng.random() < 0.3) else rng.choice([1, 16, 17, 63, 64, 65, 300, 700, 1023, 2798, 4100]) for _ in range(B)]
    bags = []
    for n in sizes:
        o = rng.randint(0, 80_000 - n)
        bags.append(pool_d[dt][o:o + n])
    Q = torch.randn(P + (1 if gated else 0), 512, generator=g).to(dev)
    T = torch.randn(K, 512, generator=g).to(dev)
    W = None if ident else (torch.randn(512, 512, generator=g) / 22).to(dev)
    b = None if ident else (torch.randn(512, generator=g) * 0.1).to(dev)
    pw = torch.randn(P, generator=g).to(dev) if mode == "weight" else None
    ls = torch.tensor(rng.uniform(2.0, 4.6), device=dev)
    rows = torch.stack([F.vlfan_aggregate(x, Q, gated)[0] for x in bags])              # [B, P, 512] from the single-bag kernels
    rl, rv, ri = tail64(rows, mode, pw, W, b, T, ls)
    plan = F.VlfanBatchPlan(B, P, K, dev, gated=gated, pool=mode, identity_head=ident)
    plan.set_bags(bags)
    for key in (None, 0, 0):                                                            # fresh preparation, then the cached one twice
        plan.run(Q, T, ls, W, b, pw, params_key=key)
    torch.cuda.synchronize()
    e = {"batch logits": (plan.logits.double() - rl).abs().max().item(), "batch vhat": (plan.vhat.double() - rv).abs().max().item(),
         "batch incidence": (plan.incidence.double() - ri).abs().max().item()}
    i = rng.randrange(B)
    sp = F.VlfanInferencePlan(sizes[i], 512, P, K, dev, gated=gated, pool=mode, identity_head=ident)
    sp.run(bags[i], Q, T, ls, W, b, pw)
    torch.cuda.synchronize()
    e["single logits"] = (sp.logits.double() - rl[i]).abs().max().item()
    e["single vhat"] = (sp.vhat.double() - rv[i]).abs().max().item()
    for k, v in e.items():
        worst[k] = max(worst[k], v)
    bad = e["batch logits"] > 1e-4 or e["single logits"] > 1e-4 or e["batch vhat"] > 1e-5 or e["single vhat"] > 1e-5 or e["batch incidence"] > 1e-4
    if bad or it % 10 == 0:
        print(f"{'FAIL ' if bad else ''}round {it}: {str(dt)[6:]} B={B} P={P} K={K} {mode} ident={ident} gated={gated} G={plan.groups} " + " ".join(f"{k}={v:.2e}" for k, v in e.items()), flush=True)
    if bad:
        sys.exit(1)
print("worst:", " ".join(f"{k}={v:.2e}" for k, v in worst.items()))
