"""Randomised cross-check of the fused (gated-)attention score kernel (vlsa_gated_scores / _batch through FusedAttnScores) against an
fp64 torch evaluation on the GPU: random N (every tile-shape boundary), module, bag dtype, row stride, several bags per launch.
python tools/fuzz_scores.py [n]"""
import os, sys, random
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from vlsa_amd import functional as F

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 300
rng = random.Random(4321)
dev = "cuda"
edges = [1, 2, 15, 16, 17, 31, 33, 63, 64, 65, 127, 128, 129, 255, 257, 1023, 1024, 1025, 2798, 4095, 4096, 4097, 16383, 16384, 16385,
         24575, 24576, 24577, 32767, 32768, 32769, 49151, 49152, 49153, 65535, 65536, 65537]
worst = 0.0


def ref(X, Wa, ba, Wg, bg, w2, c):
    Xd = X.double()
    h = torch.tanh(Xd @ Wa.double().T + ba.double())
    if Wg is not None:
        h = h * torch.sigmoid(Xd @ Wg.double().T + bg.double())
    return (h @ w2.double().T).squeeze(-1) + c.double()


for it in range(n_cases):
    gated = rng.random() < 0.5
    dtype = rng.choice([torch.bfloat16, torch.bfloat16, torch.float32])
    g = torch.Generator(device=dev).manual_seed(it)
    sc = rng.choice([1.0, 3.0])
    Wa = torch.randn(256, 512, device=dev, generator=g) / 22 * sc; ba = torch.randn(256, device=dev, generator=g) * 0.3
    Wg = torch.randn(256, 512, device=dev, generator=g) / 22 * sc if gated else None
    bg = torch.randn(256, device=dev, generator=g) * 0.3 if gated else None
    w2 = torch.randn(1, 256, device=dev, generator=g) / 16; c = torch.randn(1, device=dev, generator=g)
    fs = F.FusedAttnScores()
    if rng.random() < 0.25:                       # several bags per launch (the DeepMIL batch route's score launch)
        B = rng.randint(2, 12)
        big = rng.random() < 0.5                   # (round 5: batches of >= 16 384 rows take the one-launch scores + pooling route)
        Ns = [rng.choice(edges[:24] + [rng.randint(1, 6000)] + ([rng.randint(6000, 40000)] * 3 if big else [])) for _ in range(B)]
        bags = [torch.randn(n, 512, device=dev, generator=g).to(dtype) for n in Ns]
        pooled, scores, offs = fs.pool_bags(bags, Wa, ba, Wg, bg, w2, c)
        want, want_a = [], []
        for X in bags:
            a = ref(X, Wa, ba, Wg, bg, w2, c)
            want_a.append(a)
            want.append((torch.softmax(a, 0)[None] @ X.double()).squeeze(0))
        err = max(float((pooled.double() - torch.stack(want)).abs().max()), float((scores.double() - torch.cat(want_a)).abs().max()))
        tag = f"batch B={B} Ns={Ns[:4]}.."
    else:
        N = rng.choice(edges + [rng.randint(1, 9000), rng.randint(9000, 70000), rng.randint(70000, 220000)])
        X = torch.randn(N, 512, device=dev, generator=g).to(dtype)
        if rng.random() < 0.3:
            wide = torch.zeros(N, 512 + 8 * rng.randint(1, 40), dtype=dtype, device=dev); wide[:, :512] = X; X = wide[:, :512]
        got = fs(X, Wa, ba, Wg, bg, w2, c)
        want_a = ref(X, Wa, ba, Wg, bg, w2, c)
        err = float((got.double() - want_a).abs().max())
        tag = f"N={N} stride={X.stride(0)}"
        head = None
        if rng.random() < 0.5:                                    # ... with DeepMIL's Adapter head behind it in the same host call
            R = rng.choice([4, 36, 64, 128, 128, 128, 256, 512, 640])
            head = (torch.randn(R, 512, device=dev, generator=g) / 22, torch.randn(512, R, device=dev, generator=g) / R ** 0.5, rng.choice([0.2, 0.8, 1.0]))
        one = fs.scores_and_pool(X, Wa, ba, Wg, bg, w2, c, adapter=head)       # (round 5: scores + pooling in one launch, where it applies)
        if one is not None:
            want_p = (torch.softmax(want_a, 0)[None] @ X.double()).squeeze(0)
            err = max(err, float((one[1].double() - want_a).abs().max()), float((one[0].double().reshape(-1) - want_p).abs().max()))
            tag += " +pool"
            if head is not None:
                W1, W2, keep = head
                want_l = keep * want_p + (1 - keep) * torch.relu(W2.double() @ torch.relu(W1.double() @ want_p))
                err = max(err, float((one[2].double().reshape(-1) - want_l).abs().max()))
                tag += f" +head R={W1.shape[0]}"
    worst = max(worst, err)
    if not err < 1e-4:
        print(f"FAIL case {it}: gated={gated} {dtype} {tag}: {err:.3e}")
        sys.exit(1)
    if it % 50 == 49:
        print(f"{it + 1} cases, worst so far {worst:.2e}", flush=True)
print(f"{n_cases} cases ok, worst |d score| (or pooled row) {worst:.2e}")
