"""BASELINE config 4 ("training loop ... c-index vs reference") on synthetic data: the same short training run -- batched HIP
forward + backward through `forward_bags`, the fused IF-MLE + EMD loss kernel, Adam -- and its CPU twin through the oracle
(reference op sequence + torch.autograd) from identical seeds must follow the same loss curve and reach the same c-index
(oracle.concordance_index, pinned to the reference's evaluator by tests/golden/cindex.npz)."""
import pytest
import torch
import torch.nn as nn

import cases
from oracle import vlsa_oracle as O

pytestmark = pytest.mark.gpu

P, K, NPAT, STEPS, LR = 8, 4, 24, 40, 5e-3


def _data():
    g = cases.gen(9000)
    direction = torch.nn.functional.normalize(torch.randn(512, generator=g), dim=0)
    bags, t, e = [], [], []
    for i in range(NPAT):
        n = int(torch.randint(120, 700, (1,), generator=g))
        tb = int(torch.randint(0, K, (1,), generator=g))
        x = cases.make_bag(n, 9100 + i)
        x[: n // 3] += (2.0 - 1.2 * tb) * direction          # a planted signal: early bins lean along `direction`
        bags.append(x)
        t.append(tb)
        e.append(1.0 if float(torch.rand(1, generator=g)) < 0.6 else 0.0)
    return bags, torch.tensor(t), torch.tensor(e)


def _adam(named):
    decay = [p for n, p in named if p.dim() >= 2]
    rest = [p for n, p in named if p.dim() < 2]
    return torch.optim.Adam([{"params": rest, "weight_decay": 0.0}, {"params": decay, "weight_decay": 1e-5}], lr=LR)


def _cpu_run(bags, t, e, params):
    leaves = dict(resid=params["resid"].clone().requires_grad_(True), W=params["W"].clone().requires_grad_(True),
                  b=params["b"].clone().requires_grad_(True), T=params["T"].clone().requires_grad_(True),
                  logit_scale=torch.tensor(cases.LOGIT_SCALE, requires_grad=True))
    opt = _adam(list(leaves.items()))
    fwd = lambda: torch.cat([O.vlsa_vlfan_forward(x, 0.5 * leaves["resid"] + params["prompt"], leaves["T"], leaves["logit_scale"],  # noqa: E731
                                                  head_weight=leaves["W"], head_bias=leaves["b"])["logits"] for x in bags])
    losses = []
    for _ in range(STEPS):
        loss = O.vlsa_objective(fwd(), t, e, leaves["logit_scale"].exp())
        opt.zero_grad(); loss.backward(); opt.step()
        losses.append(float(loss.detach()))
    with torch.no_grad():
        inc = torch.softmax(fwd(), dim=-1)
    return losses, inc


def _gpu_run(bags, t, e, params):
    from vlsa_amd.losses import SurvObjective
    from vlsa_amd.prompt_adapter import PromptAdapter
    from vlsa_amd.vlsa import VLSA

    class TextParam(nn.Module):
        def __init__(self, T):
            super().__init__()
            self.T = nn.Parameter(T.clone())

    dev = torch.device("cuda", 0)
    tp = TextParam(params["T"])
    cfg = dict(name="VLFAN", dim_in=512, use_feat_proj=False, num_query=P, query="Text", query_pooling="mean")
    qnet = PromptAdapter(method="TaskRes", num_prompts=P, pretrained_prompt_features=params["prompt"], res_ratio=0.5)
    model = VLSA.from_modules(cfg, text_provider=lambda: tp.T, prompt_learner=tp, query_network=qnet, logit_scale_init=cases.LOGIT_SCALE)
    enc = model.mil_encoder
    with torch.no_grad():
        enc.Q.residual_features.copy_(params["resid"])
        enc.visual_adapter.weight.copy_(params["W"])
        enc.visual_adapter.bias.copy_(params["b"])
    model = model.to(dev).train()
    named = [("resid", enc.Q.residual_features), ("W", enc.visual_adapter.weight), ("b", enc.visual_adapter.bias),
             ("T", tp.T), ("logit_scale", model.logit_scale)]
    opt = _adam(named)
    dbags = [x.to(dev) for x in bags]
    td, ed = t.to(dev), e.to(dev)
    objective = SurvObjective()
    losses = []
    for _ in range(STEPS):
        logits = model.forward_bags(dbags)[0]
        loss = objective(logits, td, ed, model.get_logit_scale())
        opt.zero_grad(); loss.backward(); opt.step()
        losses.append(float(loss.detach()))
    model.eval()
    with torch.no_grad():
        inc = torch.softmax(model.forward_bags(dbags)[0], dim=-1).cpu()
    return losses, inc


def test_training_reaches_the_same_cindex_as_the_cpu_reference_path():
    bags, t, e = _data()
    params = cases.make_params(P, K, 9001)
    y = torch.stack([t.float(), e], dim=1)
    cpu_losses, cpu_inc = _cpu_run(bags, t, e, params)
    gpu_losses, gpu_inc = _gpu_run(bags, t, e, params)
    c_cpu, c_gpu = O.concordance_index(y, cpu_inc), O.concordance_index(y, gpu_inc)
    Q0 = 0.5 * params["resid"] + params["prompt"]
    with torch.no_grad():
        inc0 = torch.softmax(torch.cat([O.vlsa_vlfan_forward(x, Q0, params["T"], torch.tensor(cases.LOGIT_SCALE), head_weight=params["W"],
                                                            head_bias=params["b"])["logits"] for x in bags]), dim=-1)
    c0 = O.concordance_index(y, inc0)
    for i, (a, b) in enumerate(zip(gpu_losses, cpu_losses)):
        assert abs(a - b) < 2e-3 * max(1.0, abs(b)), (i, a, b)
    assert gpu_losses[-1] < gpu_losses[0] - 0.1                 # it actually trained
    assert c_gpu > c0 + 0.05 and abs(c_gpu - c_cpu) <= 0.01, (c0, c_cpu, c_gpu)
    assert (gpu_inc - cpu_inc).abs().max().item() < 5e-3
