"""The single-slide tail alone (no streaming kernel in between): vlsa_vlfan_merge_head (round 5: merge + W-partials, finish) against
vlsa_vlfan_merge + vlsa_head_forward (round 4), G = 256 partial records of a 50k-patch bag.  `python tools/kbench_tail.py`"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vlsa_amd import _native as nat, functional as F
lib = nat.load()
dev = "cuda"
P, D, K, G = 12, 512, 4, 256
g = torch.Generator(device=dev).manual_seed(3)
pm = torch.randn(G, 16, device=dev, generator=g); pl = torch.rand(G, 16, device=dev, generator=g) + 0.5
pacc = torch.randn(G, P, D, device=dev, generator=g)
W = torch.randn(D, D, device=dev, generator=g) / 22; b = torch.randn(D, device=dev, generator=g)
That = torch.nn.functional.normalize(torch.randn(K, D, device=dev, generator=g), dim=-1); ls = torch.tensor([4.03], device=dev)
ws = torch.zeros(lib.vlsa_head_workspace_bytes(D), dtype=torch.uint8, device=dev)
f = lambda *s: torch.empty(*s, device=dev)
m2, l, out, pooled, v, vhat, vnorm, logits, inc = f(16), f(16), f(P, D), f(D), f(D), f(D), f(1), f(K), f(K)
p = F._p
big = torch.empty(64 << 20, device=dev)     # 256 MB: pushes W and the partials out of the caches between calls when asked


def new():
    nat.check(lib.vlsa_vlfan_merge_head(p(pm), p(pl), p(pacc), G, P, D, 0, None, p(W), p(b), p(That), K, p(ls), p(ws), p(m2), p(l), p(out),
                                        p(pooled), p(v), p(vhat), p(vnorm), p(logits), p(inc), F._stream()), "merge_head")


def old():
    nat.check(lib.vlsa_vlfan_merge(p(pm), p(pl), p(pacc), G, P, D, 1, p(m2), p(l), p(out), F._stream()), "merge")
    nat.check(lib.vlsa_head_forward(p(out), P, D, 0, None, p(W), p(b), p(That), K, p(ls), p(ws), p(pooled), p(v), p(vhat), p(vnorm),
                                    p(logits), p(inc), F._stream()), "head")


def t(fn, n=200, flush=False):
    for _ in range(20): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    if not flush:
        e0.record()
        for _ in range(n): fn()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) * 1e3 / n
    tot = 0.0
    for _ in range(40):
        big.zero_()
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        tot += e0.elapsed_time(e1) * 1e3
    return tot / 40


new(); ln = logits.clone(); old()
print("max |dlogit| new vs old", float((ln - logits).abs().max()))
print(f"new: {t(new):.1f} us back to back, {t(new, flush=True):.1f} us with cold caches")
print(f"old: {t(old):.1f} us back to back, {t(old, flush=True):.1f} us with cold caches")
