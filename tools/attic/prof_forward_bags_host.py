"""cProfile of the HOST side of net.forward_bags(64 small bags) in eval mode and of one training step's forward_bags + backward"""
import sys, os, time, cProfile, pstats
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from vlsa_amd.vlsa import VLSA
import gc; gc.collect(); gc.freeze()
dev = "cuda"
P, K = 12, 4
cfg = dict(name="VLFAN", dim_in=512, use_feat_proj=False, query="Parameter", num_query=P, query_pooling="mean")
net = VLSA.from_modules(cfg, pretrained_text_features=torch.randn(K, 512)).to(dev).eval()
n = 2798
base = torch.randn(64 * n, 512, device=dev).to(torch.bfloat16)
bags = [base[i * n:(i + 1) * n][None] for i in range(64)]
with torch.no_grad():
    for _ in range(20): net.forward_bags(bags)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(100): net.forward_bags(bags)
    t_host = (time.perf_counter() - t0) / 100
    torch.cuda.synchronize(); t_all = (time.perf_counter() - t0) / 100
    print(f"forward_bags(64 x {n}): host {t_host * 1e6:.0f} us per call, with GPU {t_all * 1e6:.0f} us = {t_all / 64 * 1e6:.2f} us/bag")
    pr = cProfile.Profile(); pr.enable()
    for _ in range(100): net.forward_bags(bags)
    pr.disable()
    pstats.Stats(pr).sort_stats("tottime").print_stats(16)
net.train()
opt = torch.optim.Adam([p for p in net.parameters() if p.requires_grad], lr=1e-4)
bags32 = bags[:32]
def step():
    logits = net.forward_bags(bags32)[0]
    loss = logits.square().mean()
    opt.zero_grad(set_to_none=True); loss.backward(); opt.step()
for _ in range(10): step()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(50): step()
torch.cuda.synchronize(); print(f"train step (32 x {n}, no text side): {(time.perf_counter() - t0) / 50 * 1e6:.0f} us")
pr = cProfile.Profile(); pr.enable()
for _ in range(50): step()
pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(22)
