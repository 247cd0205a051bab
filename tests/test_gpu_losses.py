"""Fused loss tail (vlsa_surv_loss) vs the CPU oracle's SurvIFMLE / SurvEMD restatements (pinned to the reference by the
train_step fixture) -- values and gradients -- and inside a full optimizer trajectory."""
import pytest
import torch

import cases
from oracle import vlsa_oracle as O

pytestmark = pytest.mark.gpu


def _inputs(B, K, seed, extreme=False):
    g = cases.gen(seed)
    logits = torch.randn(B, K, generator=g) * (12.0 if extreme else 2.0)
    t = torch.randint(0, K, (B,), generator=g)
    e = (torch.rand(B, generator=g) < 0.45).float()
    # a censored sample in the LAST bin makes the reference evaluate log(clamp(1 - cumsum(softmax)[K-1], 1e-7)): 1 - 1 up to
    # rounding, i.e. -log(1e-7) or -log(1.2e-7) depending on the last ulp of the cumsum -- covered by its own test below
    t = torch.where((e == 0) & (t == K - 1), torch.full_like(t, max(K - 2, 0)), t)
    if B >= 4:
        t[0], e[0] = 0, 1.0
        t[1], e[1] = max(K - 2, 0), 0.0
        t[2], e[2] = K - 1, 1.0
        t[3], e[3] = 0, 0.0
    return logits, t, e


def test_censored_in_last_bin_is_clamped_like_the_reference():
    from vlsa_amd.losses import SurvIFMLE
    K = 12
    logits = torch.randn(8, K, generator=cases.gen(2300))
    inc = torch.softmax(logits, dim=-1)
    t, e = torch.full((8,), K - 1), torch.zeros(8)
    ref = O.surv_ifmle(inc, t, e)
    xd = inc.cuda().requires_grad_(True)
    got = SurvIFMLE(reduction="none")(xd, t.cuda(), e.cuda())
    got.sum().backward()
    # every sample is -log of something in [1e-7 (clamped), 4e-7 (a few ulps of 1 - cumsum)]
    assert got.shape == (8, 1)
    assert (got.detach() <= 16.2).all() and (got.detach() >= 14.7).all()
    assert 14.7 <= ref.item() <= 16.2
    assert torch.isfinite(xd.grad).all()


@pytest.mark.parametrize("B,K", [(1, 4), (5, 12), (32, 4), (32, 12), (130, 8)])
@pytest.mark.parametrize("extreme", [False, True])
def test_objective_from_raw_logits(B, K, extreme):
    from vlsa_amd.losses import SurvObjective
    logits, t, e = _inputs(B, K, 2000 + B + K, extreme)
    # censored samples whose remaining mass 1 - CIF[t] is below 1e-4 are decided by the last ulp of an fp32 cumsum in the
    # reference itself (see _inputs): make those samples uncensored instead
    tail = 1.0 - torch.cumsum(torch.softmax(logits.double(), dim=-1), dim=-1).gather(1, t.view(-1, 1)).view(-1)
    e = torch.where((e == 0) & (tail < 1e-4), torch.ones_like(e), e)
    ls = torch.tensor(cases.LOGIT_SCALE).exp()
    x = logits.clone().requires_grad_(True)
    ref = O.vlsa_objective(x, t, e, ls)
    ref.backward()
    xd = logits.cuda().requires_grad_(True)
    got = SurvObjective()(xd, torch.stack([t.float(), e], dim=1).cuda(), cur_logit_scale=ls.cuda())
    got.backward()
    assert abs(got.item() - ref.item()) < 1e-5 * max(1.0, abs(ref.item()))
    assert (xd.grad.cpu() - x.grad).abs().max().item() < 1e-5 * max(1.0, x.grad.abs().max().item())
    # round 6: the raw (log) logit scale handed over instead, exponentiated inside the launch; an upstream factor on the scalar; and the
    # route of batches beyond one block's reach (B > 4096 falls back to the per-sample kernel + torch mean: same numbers)
    xd2 = logits.cuda().requires_grad_(True)
    got2 = SurvObjective()(xd2, t.cuda(), e.cuda(), log_logit_scale=torch.tensor(cases.LOGIT_SCALE).cuda())
    (3.0 * got2).backward()
    assert abs(got2.item() - got.item()) < 2e-6 * max(1.0, abs(got.item()))
    assert (xd2.grad - 3.0 * xd.grad).abs().max().item() < 1e-5 * max(1.0, xd.grad.abs().max().item())
    # value_and_grad (what TrainStep starts its backward pass from): the same launch outside autograd -- bit-equal to the route above
    v3, g3 = SurvObjective().value_and_grad(logits.cuda(), t.cuda(), e.cuda(), log_logit_scale=torch.tensor(cases.LOGIT_SCALE).cuda())
    assert torch.equal(v3, got2.detach()) and torch.equal(3.0 * g3, xd2.grad)


def test_objective_of_a_large_batch_takes_the_per_sample_route():
    from vlsa_amd.losses import SurvObjective
    logits, t, e = _inputs(5000, 6, 2999, False)
    tail = 1.0 - torch.cumsum(torch.softmax(logits.double(), dim=-1), dim=-1).gather(1, t.view(-1, 1)).view(-1)
    e = torch.where((e == 0) & (tail < 1e-4), torch.ones_like(e), e)
    ls = torch.tensor(cases.LOGIT_SCALE).exp()
    ref = O.vlsa_objective(logits, t, e, ls)
    got = SurvObjective()(logits.cuda(), t.cuda(), e.cuda(), cur_logit_scale=ls.cuda())
    got_log = SurvObjective()(logits.cuda(), t.cuda(), e.cuda(), log_logit_scale=torch.tensor(cases.LOGIT_SCALE).cuda())
    assert abs(got.item() - ref.item()) < 1e-5 * max(1.0, abs(ref.item())) and abs(got_log.item() - got.item()) < 2e-6 * max(1.0, abs(got.item()))


@pytest.mark.parametrize("alpha", [0.0, 0.3])
@pytest.mark.parametrize("reduction", ["mean", "sum", "none"])
def test_ifmle_module(alpha, reduction):
    from vlsa_amd.losses import SurvIFMLE
    logits, t, e = _inputs(16, 12, 2100)
    inc = torch.softmax(logits, dim=-1)
    inc[5, int(t[5])] = 1e-9            # below eps: clamped, zero gradient through the clamp
    x = inc.clone().requires_grad_(True)
    B = len(t)
    c = 1 - e.view(B, 1)
    cif = torch.cumsum(x, dim=1)
    unc = -(1 - c) * torch.log(torch.gather(x, 1, t.view(B, 1)).clamp(min=1e-7))
    cen = -c * torch.log((1 - torch.gather(cif, 1, t.view(B, 1))).clamp(min=1e-7))
    ref = (1.0 - alpha) * (cen + unc) + alpha * unc
    ref = ref.mean() if reduction == "mean" else ref.sum() if reduction == "sum" else ref
    if reduction == "mean" and alpha == 0.0:
        assert abs(O.surv_ifmle(inc, t, e).item() - ref.item()) < 1e-6
    w = torch.randn_like(ref) if reduction == "none" else None
    (ref if w is None else (ref * w).sum()).backward()
    xd = inc.cuda().requires_grad_(True)
    got = SurvIFMLE(alpha=alpha, reduction=reduction)(xd, t.cuda(), e.cuda())
    (got if w is None else (got * w.cuda()).sum()).backward()
    assert got.shape == ref.shape
    assert (got.detach().cpu() - ref.detach()).abs().max().item() < 1e-5 * max(1.0, ref.detach().abs().max().item())
    # d/d inc of -log(1 - CIF[t]) is 1 / (1 - CIF[t]): an fp32 cumsum's rounding (6e-8) over a remaining mass of 1e-3
    assert (xd.grad.cpu() - x.grad).abs().max().item() < 2e-4 * max(1.0, x.grad.abs().max().item())


@pytest.mark.parametrize("p,raw", [(2, True), (2, False), (1, True)])
def test_emd_module(p, raw):
    from vlsa_amd.losses import SurvEMD
    logits, t, e = _inputs(24, 8, 2200)
    if not raw:  # censored samples: prediction and target are both ~uniform over the bins >= t, the squared distance underflows to
        e = torch.ones_like(e)  # exactly 0 and sqrt'(0) is NaN in torch's autograd (the kernel returns a zero gradient there)
    inc = torch.softmax(logits, dim=-1)
    ls = 56.3
    x = inc.clone().requires_grad_(True)
    if p == 2 and raw:
        ref = O.surv_emd(x, t, e, ls, p=2)
    else:  # same construction, other distance (loss/loss_surv_ext.py:26-38)
        B, K = x.shape
        tt, ee = t.view(-1, 1), e.view(-1, 1).long()
        target = torch.zeros(B, K).scatter_(1, tt, 1)
        for i in range(B):
            if int(tt[i, 0]) + 1 < K:
                target[i, int(tt[i, 0]) + 1:] += (1 - ee[i, 0])
        td = torch.softmax((2 * target - 1) * ls, dim=-1)
        pred = (1 - ee) * ((1 - target) * x + target * ls) + ee * x
        d = torch.cumsum(torch.softmax(pred, dim=-1), dim=-1) - torch.cumsum(td, dim=-1)
        ref = (d.abs().sum(dim=-1) if p == 1 else torch.sqrt((d ** 2).sum(dim=-1))).mean()
    ref.backward()
    xd = inc.cuda().requires_grad_(True)
    got = SurvEMD(p=p, raw_distance=raw)(xd, t.cuda(), e.cuda(), ls)
    got.backward()
    assert abs(got.item() - ref.item()) < 1e-5 * max(1.0, abs(ref.item()))
    assert (xd.grad.cpu() - x.grad).abs().max().item() < 2e-5 * max(1.0, x.grad.abs().max().item())
