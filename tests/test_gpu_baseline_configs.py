"""One parity test per BASELINE.json config, at the config's own size, against the CPU oracle (pinned to the reference by
tests/golden): incidence logits and attention weights within the north-star tolerance 1e-4.

  configs[0]  2798 x 512 fp32 single slide through the cfg_vlsa_conch.yaml-shaped plumbing (P = K = 12)
  configs[1]  10k x 512 fp32, K = 4
  configs[2]  50k x 512 bf16, K = 4
  configs[3]  200k x 512 bf16, K = 8, patch-sharded over 8 ranks (emulated on one GPU: 8 shards, strided record merge)
  configs[4]  training: tests/test_train_step.py (the reference's Adam trajectory on synthetic bags)
"""
import ctypes

import pytest
import torch

import cases
from oracle import vlsa_oracle as O

pytestmark = pytest.mark.gpu
TOL = 1e-4


def _model(P, K, params, dev):
    from vlsa_amd.prompt_adapter import PromptAdapter
    from vlsa_amd.vlsa import VLSA
    # image_encoder_cfg exactly as VLSAHandler strips it out of cfg_vlsa_conch.yaml (runner/vlsa_handler.py:110-111)
    cfg = dict(name="VLFAN", dim_in=512, dim_hid=256, use_feat_proj=False, drop_rate=0.25, num_query=P, query="Text",
               gated_query=False, query_pooling="mean", pred_head="default", pooling="logit_top10")
    qnet = PromptAdapter(method="TaskRes", num_prompts=P, pretrained_prompt_features=params["prompt"], res_ratio=0.5)
    m = VLSA.from_modules(cfg, pretrained_text_features=params["T"].clone(), query_network=qnet, logit_scale_init=cases.LOGIT_SCALE)
    with torch.no_grad():
        m.mil_encoder.Q.residual_features.copy_(params["resid"])
        m.mil_encoder.visual_adapter.weight.copy_(params["W"])
        m.mil_encoder.visual_adapter.bias.copy_(params["b"])
    return m.to(dev).eval()


def _check(model, X, params, P, K):
    Q = 0.5 * params["resid"] + params["prompt"]
    ref = O.vlsa_vlfan_forward(X.float(), Q, params["T"], torch.tensor(cases.LOGIT_SCALE), head_weight=params["W"],
                               head_bias=params["b"])
    dev = next(model.parameters()).device
    with torch.no_grad():
        logits, feats, That = model(X[None].to(dev))
        v, A = model.mil_encoder(X[None].to(dev), ret_with_attn=True)
    assert (logits.cpu() - ref["logits"]).abs().max().item() < TOL
    assert (torch.softmax(logits, -1).cpu() - ref["incidence"]).abs().max().item() < TOL
    assert (feats.cpu() - ref["v_hat"]).abs().max().item() < TOL
    assert (A[0].cpu() - ref["A"]).abs().max().item() < TOL
    return ref


def test_config0_single_slide_2798_fp32_through_cfg_plumbing():
    P = K = 12
    params = cases.make_params(P, K, 7000)
    X = cases.make_bag(2798, 7001, "clustered")
    _check(_model(P, K, params, torch.device("cuda", 0)), X, params, P, K)


def test_config1_10k_fp32_k4():
    P, K = 12, 4
    params = cases.make_params(P, K, 7010)
    X = cases.make_bag(10_000, 7011)
    model = _model(P, K, params, torch.device("cuda", 0))
    ref = _check(model, X, params, P, K)
    with torch.no_grad():                                  # the batched path on the same bag (exact-fp32 persistent kernel)
        lb, _, _ = model.forward_bags([X.cuda(), X.cuda()[:5000]])
    assert (lb[0].cpu() - ref["logits"][0]).abs().max().item() < TOL


def test_config2_50k_bf16_k4():
    P, K = 12, 4
    params = cases.make_params(P, K, 7020)
    X = cases.make_bag(50_000, 7021).to(torch.bfloat16)   # oracle = fp32 reference math on the bf16-rounded values
    model = _model(P, K, params, torch.device("cuda", 0))
    ref = _check(model, X, params, P, K)
    with torch.no_grad():
        lb, _, _ = model.forward_bags([X.cuda()] * 3)      # the benchmarked path
    assert (lb.cpu() - ref["logits"]).abs().max().item() < TOL


def test_config3_200k_bf16_k8_sharded_over_8_ranks():
    from vlsa_amd import _native as nat, functional as F
    from vlsa_amd.sharded import REC_HDR, record_floats, shard_bounds
    P, K, D, world, N = 12, 8, 512, 8, 200_000
    params = cases.make_params(P, K, 7030)
    X = cases.make_bag(N, 7031).to(torch.bfloat16)
    Q = 0.5 * params["resid"] + params["prompt"]
    ref = O.vlsa_vlfan_forward(X.float(), Q, params["T"], torch.tensor(cases.LOGIT_SCALE), head_weight=params["W"],
                               head_bias=params["b"])
    dev = torch.device("cuda", 0)
    Xd, Qd = X.to(dev), Q.to(dev)
    qp = F.prepare_queries(Qd)
    rf = record_floats(P, D)
    gathered = torch.zeros(world, rf, device=dev)          # what the RCCL all-gather delivers: one record per rank
    A_parts = []
    for r in range(world):
        a, b = shard_bounds(N, world, r)                   # 25 000 rows per rank
        pm, pl, pacc, sc = F.vlfan_partial(Xd[a:b], qp, want_scores=True)
        m2, l, acc = F.vlfan_merge(pm, pl, pacc, normalise=False)
        gathered[r, :16], gathered[r, 16:32], gathered[r, 32:] = m2, l, acc.reshape(-1)
        A_parts.append(sc)
    lib = nat.load()
    m2g, lg, out = torch.empty(16, device=dev), torch.empty(16, device=dev), torch.empty(P, D, device=dev)
    base = gathered.data_ptr()
    nat.check(lib.vlsa_vlfan_merge_strided(ctypes.c_void_p(base), rf, ctypes.c_void_p(base + 64), rf,
                                           ctypes.c_void_p(base + 4 * REC_HDR), rf, world, P, D, 1, F._p(m2g), F._p(lg),
                                           F._p(out), F._stream()), "merge_strided")
    That, _ = F.normalize_rows(params["T"].to(dev))
    h = F.head_forward(out, "mean", None, params["W"].to(dev), params["b"].to(dev), That,
                       torch.tensor(cases.LOGIT_SCALE, device=dev), want_incidence=True)
    A = torch.cat([F.attn_normalise(sc, m2g, lg) for sc in A_parts], dim=1)   # weights stay sharded, global (m, l)
    assert (h["logits"].cpu() - ref["logits"][0]).abs().max().item() < TOL
    assert (h["incidence"].cpu() - ref["incidence"][0]).abs().max().item() < TOL
    assert (A.cpu() - ref["A"]).abs().max().item() < TOL
