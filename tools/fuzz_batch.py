"""Randomised cross-check of the persistent multi-bag kernels against the single-bag kernels: random bag counts / sizes (tile and
unit boundaries, empty workgroups, one huge bag among tiny ones), both dtypes, with and without attention weights, forward and
backward.  python tools/fuzz_batch.py [rounds] [seed]"""
import os, sys, random
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vlsa_amd import functional as F

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 60
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
rng = random.Random(seed)
dev = torch.device("cuda", 0)
g = torch.Generator().manual_seed(seed)
pool = torch.randn(70_000, 512, generator=g)
pool[::7] *= 3.0
pool_d = {torch.bfloat16: pool.to(torch.bfloat16).to(dev), torch.float32: pool.to(dev)}
worst = {"out": 0.0, "A": 0.0, "grad": 0.0}
special = [1, 15, 16, 17, 31, 32, 33, 63, 64, 65, 127, 128, 129, 255, 256, 257, 511, 512, 513, 1023, 1024, 1025, 4095, 4096, 4097]
for it in range(rounds):
    dt = rng.choice([torch.bfloat16, torch.float32])
    B = rng.choice([1, 2, 3, 5, 8, 13, 32, 33, 64])
    P = rng.choice([1, 4, 7, 12, 12, 12, 13, 16])
    gated = rng.random() < 0.3
    sizes = []
    for _ in range(B):
        r = rng.random()
        sizes.append(rng.choice(special) if r < 0.5 else (rng.randint(1, 3000) if r < 0.9 else rng.randint(20_000, 60_000)))
    bags = []
    for n in sizes:
        o = rng.randint(0, 70_000 - n)
        bags.append(pool_d[dt][o:o + n])
    Q = torch.randn(P + (1 if gated else 0), 512, generator=g).to(dev)
    Qa = Q.clone().requires_grad_(True)
    want = rng.random() < 0.5
    r = F.vlfan_cross_attention_bags(bags, Qa, gated=gated, want_attn=want)
    out, attn = (r if want else (r, None))
    G = torch.randn(B, P, 512, generator=g).to(dev)
    (out * G).sum().backward()
    Qb = Q.clone().requires_grad_(True)
    tot = 0.0
    for i, x in enumerate(bags):
        o, A = F.vlfan_cross_attention(x, Qb, gated=gated, want_attn=want)
        e = (o - out[i]).abs().max().item() / max(1.0, o.abs().max().item())
        worst["out"] = max(worst["out"], e)
        assert e < 6e-5, ("out", it, i, sizes[i], dt, P, gated, e)      # two different kernels, each within 1e-4 of the oracle
        if want:
            ea = (A - attn[i]).abs().max().item()
            worst["A"] = max(worst["A"], ea)
            assert ea < 2e-5, ("A", it, i, sizes[i], dt, P, gated, ea)
        tot = tot + (o * G[i]).sum()
    tot.backward()
    eg = (Qa.grad - Qb.grad).abs().max().item() / max(1.0, Qb.grad.abs().max().item())   # (a one-patch bag has a zero gradient: only rounding noise ~1e-4 of O(100) terms)
    worst["grad"] = max(worst["grad"], eg)
    assert eg < 1e-3, ("grad", it, sizes, dt, P, gated, eg)
torch.cuda.synchronize()
print("fuzz ok:", rounds, "rounds; worst relative differences", worst)

# ---- the other batched launches: zero-shot (per-class cosines + top-k mean) and the DeepMIL scores + pooling -----------------
worst2 = {"zeroshot": 0.0, "deepmil_a": 0.0, "deepmil_pool": 0.0}
fs = {True: F.FusedAttnScores(), False: F.FusedAttnScores()}
W = {gt: [t.to(dev) if t is not None else None for t in
          (torch.randn(256, 512, generator=g) / 8, torch.randn(256, generator=g) * 0.05,
           torch.randn(256, 512, generator=g) / 8 if gt else None, torch.randn(256, generator=g) * 0.05 if gt else None,
           torch.randn(1, 256, generator=g) / 16, torch.randn(1, generator=g) * 0.05)] for gt in (True, False)}
for it in range(rounds // 2):
    dt = rng.choice([torch.bfloat16, torch.float32])
    B = rng.choice([1, 2, 5, 13, 32, 64])
    sizes = [rng.choice(special) if rng.random() < 0.5 else (rng.randint(1, 3000) if rng.random() < 0.9 else rng.randint(20_000, 60_000))
             for _ in range(B)]
    bags = []
    for n in sizes:
        o = rng.randint(0, 70_000 - n)
        bags.append(pool_d[dt][o:o + n])
    K = rng.choice([1, 4, 8, 12, 16, 20])
    T = torch.randn(K, 512, generator=g).to(dev)
    ls = torch.tensor(4.03, device=dev)
    k = rng.choice([None, 1, 3, 10, 32])
    zs = F.zeroshot_pool_bags(bags, T, ls, k)
    for i, x in enumerate(bags):
        cos = F.class_cosines(x, T)
        ref = ls.exp() * F.topk_mean(cos, x.shape[0] if k is None else min(k, x.shape[0]))
        e = (zs[i] - ref).abs().max().item()
        worst2["zeroshot"] = max(worst2["zeroshot"], e)
        assert e < 2e-4, ("zeroshot", it, i, sizes[i], dt, K, k, e)
    gated = rng.random() < 0.5
    pooled, a, offs = fs[gated].pool_bags(bags, *W[gated])
    for i, x in enumerate(bags):
        ai = fs[gated](x, *W[gated])
        ea = (ai - a[offs[i]:offs[i + 1]]).abs().max().item()
        ep = (F.scored_pool(x, ai) - pooled[i]).abs().max().item()
        worst2["deepmil_a"], worst2["deepmil_pool"] = max(worst2["deepmil_a"], ea), max(worst2["deepmil_pool"], ep)
        assert ea < 5e-6 and ep < 2e-5, ("deepmil", it, i, sizes[i], dt, gated, ea, ep)
torch.cuda.synchronize()
print("fuzz ok (other encoders):", rounds // 2, "rounds; worst absolute differences", worst2)

# ---- wide forward launches (round 4: up to 256 bags per launch, a workgroup's table holds its own bags only): random bag counts,
# bags in flight (auto and forced, incl. one workgroup per bag), both dtypes, with and without attention weights
worst3 = {"out": 0.0, "A": 0.0}
for it in range(max(4, rounds // 6)):
    dt = rng.choice([torch.bfloat16, torch.float32])
    B = rng.choice([65, 97, 128, 129, 200, 255, 256])
    P = rng.choice([1, 4, 12, 12, 13, 16])
    gated = rng.random() < 0.3
    sizes = [rng.choice(special) if rng.random() < 0.5 else (rng.randint(1, 3000) if rng.random() < 0.97 else rng.randint(20_000, 60_000))
             for _ in range(B)]
    bags = []
    for n in sizes:
        o = rng.randint(0, 70_000 - n)
        bags.append(pool_d[dt][o:o + n])
    Q = torch.randn(P + (1 if gated else 0), 512, generator=g).to(dev)
    T = torch.randn(4, 512, generator=g).to(dev)
    Wh = (torch.randn(512, 512, generator=g) / 22).to(dev)
    bh = torch.randn(512, generator=g).to(dev)
    ls = torch.tensor(4.03, device=dev)
    want = rng.random() < 0.4
    plan = F.VlfanBatchPlan(B, P, 4, dev, gated=gated, want_attn=want)
    plan.set_bags(bags)
    forced = rng.choice([0, 0, 4, 16, 64, 128, 256])
    if forced:
        plan.groups = forced
    plan.run(Q, T, ls, Wh, bh)
    out = plan.out.clone()
    for i in rng.sample(range(B), 24):
        o, A = F.vlfan_cross_attention(bags[i], Q, gated=gated, want_attn=want)
        e = (o - out[i]).abs().max().item() / max(1.0, o.abs().max().item())
        worst3["out"] = max(worst3["out"], e)
        assert e < 6e-5, ("wide out", it, i, sizes[i], dt, P, gated, B, forced, e)
        if want:
            ea = (A - plan.attn.views[i]).abs().max().item()
            worst3["A"] = max(worst3["A"], ea)
            assert ea < 2e-5, ("wide A", it, i, sizes[i], dt, P, gated, B, forced, ea)
    del plan
torch.cuda.synchronize()
print("fuzz ok (wide launches):", max(4, rounds // 6), "rounds; worst differences", worst3)
