"""Host profile of a handler-shaped training step with a DeepMIL encoder inside VLSA (one net(X) per bag, cat, one backward), the
backward on the calling thread so that cProfile sees the Python side of the autograd nodes."""
import sys, os, time, cProfile, pstats
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import torch.nn as nn
from vlsa_amd.vlsa import VLSA
import gc; gc.collect(); gc.freeze()
dev = "cuda"
K = 12
pooling = sys.argv[1] if len(sys.argv) > 1 else "gated_attention"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 5000


class TextParam(nn.Module):
    def __init__(self):
        super().__init__()
        self.T = nn.Parameter(torch.randn(K, 512))


tp = TextParam()
cfg = dict(name="DeepMIL", dim_in=512, dim_hid=256, num_cls=512, use_feat_proj=False, drop_rate=0.25, pooling=pooling, pred_head="Adapter",
           dim_reduction=4, keep_ratio=0.8)
net = VLSA.from_modules(cfg, text_provider=lambda: tp.T, prompt_learner=tp).to(dev).train()
opt = torch.optim.Adam([p for p in net.parameters() if p.requires_grad], lr=1e-4)
bags = [torch.randn(n, 512, device=dev).to(torch.bfloat16) for _ in range(32)]
G = torch.randn(32, K, device=dev)


def step():
    logits = torch.cat([net(x[None])[0] for x in bags], dim=0)
    opt.zero_grad(set_to_none=True)
    (logits * G).sum().backward()
    opt.step()


for _ in range(5):
    step()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(10):
    step()
torch.cuda.synchronize()
print(f"DeepMIL({pooling}) N={n}: handler-shaped step of 32 bags {(time.perf_counter() - t0) / 10 * 1e3:.2f} ms")
torch.autograd.set_multithreading_enabled(False)
pr = cProfile.Profile(); pr.enable()
for _ in range(5):
    step()
torch.cuda.synchronize()
pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(45)
