"""vlsa_amd.train_step.TrainStep (round 6): the optimizer step of the reference's training loop (runner/vlsa_handler.py:260-289:
32 x net(X), cat, calc_objective_loss, one backward, optimizer.step) owned by one object -- eager for batches it has not seen twice,
then ONE hipGraph replay per step.  The replayed steps must be the eager steps: same losses, same parameters, and nothing that is cached
on parameter versions (text features, prepared queries, look-ahead state) may survive a replay."""
import pytest
import torch

import cases
import text_cases as TC

pytestmark = pytest.mark.gpu

K, P, TOWER, TSEED = 12, 12, "train", 9300


def _model():
    from vlsa_amd.prompt_adapter import PromptAdapter
    from vlsa_amd.prompt_encoder import CONCHPromptEncoder
    from vlsa_amd.prompt_learner import RankPromptLearner
    from vlsa_amd.vlsa import VLSA
    params = cases.make_params(P, K, 9201)
    c = TC.TOWERS[TOWER]
    enc = CONCHPromptEncoder(width=c["width"], heads=c["heads"], layers=c["layers"], vocab_size=c["vocab"], output_dim=c["out_dim"])
    enc.load_state_dict(TC.make_tower_weights(TOWER, TSEED))
    for p_ in enc.parameters():
        p_.requires_grad_(False)
    table, ctx_key, names = TC.synthetic_prompt_table(c["vocab"], TSEED)
    pl = RankPromptLearner(dict(max_num_tokens=127, embedding_dim=c["width"], embedding_dtype=torch.float32), TC.ReplayTokenizer(table),
                           enc.token_embedding, num_base_ranks=4, num_ranks=K, num_tokens_per_rank=4, num_context_tokens=8,
                           init_context=ctx_key, init_rank_names=names)
    qnet = PromptAdapter(method="TaskRes", num_prompts=P, pretrained_prompt_features=params["prompt"], res_ratio=0.5)
    cfg = dict(name="VLFAN", dim_in=512, use_feat_proj=False, num_query=P, query="Text", query_pooling="mean", pred_head="default")
    net = VLSA.from_modules(cfg, prompt_learner=pl, prompt_encoder=enc, query_network=qnet, logit_scale_init=cases.LOGIT_SCALE).cuda().train()
    with torch.no_grad():
        qnet.residual_features.copy_(params["resid"])
        net.mil_encoder.visual_adapter.weight.copy_(params["W"])
        net.mil_encoder.visual_adapter.bias.copy_(params["b"])
    named = [net.mil_encoder.Q.residual_features, net.mil_encoder.visual_adapter.weight, net.mil_encoder.visual_adapter.bias,
             pl.context_embeds, pl.rank_embeds, net.logit_scale]
    return net, named


def _groups(ps):
    return [{"params": [p for p in ps if p.dim() < 2], "weight_decay": 0.0}, {"params": [p for p in ps if p.dim() >= 2], "weight_decay": 1e-5}]


def _batches():
    g = torch.Generator().manual_seed(77)
    out = []
    for b in range(2):
        bags = [cases.make_bag(int(torch.randint(200, 1500, (1,), generator=g)), 9600 + 40 * b + i, "clustered").to(torch.bfloat16).cuda()
                for i in range(8)]
        t = torch.randint(0, K - 1, (8,), generator=g).cuda()
        e = (torch.rand(8, generator=g) < 0.5).float().cuda()
        out.append((bags, t, e))
    return out


def _run(graph, n_steps, order, lr_change_at=None):
    from vlsa_amd.losses import SurvObjective
    from vlsa_amd.optim import FusedAdam
    from vlsa_amd.train_step import TrainStep
    net, named = _model()
    opt = FusedAdam(_groups(named), lr=1e-3)
    ts = TrainStep(net, SurvObjective(), opt, graph=graph)
    batches = _batches()
    losses, text = [], []
    for i in range(n_steps):
        if lr_change_at is not None and i == lr_change_at:
            for gr in opt.param_groups:
                gr["lr"] = 3e-3
        bags, t, e = batches[order[i % len(order)]]
        v0 = [p._version for p in named]
        loss = ts.step(bags, t, e)
        losses.append(float(loss))                      # (reads the value before the next replay overwrites the static tensor)
        assert all(p._version > v for p, v in zip(named, v0)), i
        if i % 4 == 3:                                   # an evaluation-style read between steps: must see the CURRENT prompts
            net.eval()
            with torch.no_grad():
                text.append(net.forward_text_only().clone())
            net.train()
    return losses, [p.detach().clone() for p in named], text, ts


def test_replayed_steps_equal_the_eager_steps_one_batch():
    le, pe, te, _ = _run(False, 12, [0])
    lg, pg, tg, ts = _run(True, 12, [0])
    d = ts.describe()
    assert d["replays"] >= 9 and d["captures"] == 1 and d["why_eager"] is None, d
    for a, b in zip(le, lg):
        assert abs(a - b) <= 2e-6 * max(1.0, abs(a)), (le, lg)
    for a, b in zip(pe, pg):
        assert (a - b).abs().max().item() <= 2e-6 * max(1.0, a.abs().max().item())
    assert len(te) == 3
    for a, b in zip(te, tg):                             # the text cache was not served from before a replay
        assert (a - b).abs().max().item() < 1e-5
    assert (te[0] - te[2]).abs().max().item() > 1e-4     # ... and the prompts did move


def test_two_alternating_batches_get_a_graph_each_and_the_learning_rate_is_followed():
    order = [0, 1, 0, 1, 0, 1, 1, 0]
    le, pe, _, _ = _run(False, 16, order, lr_change_at=9)
    lg, pg, _, ts = _run(True, 16, order, lr_change_at=9)
    d = ts.describe()
    assert d["captures"] == 2 and d["replays"] >= 8, d
    for a, b in zip(le, lg):
        assert abs(a - b) <= 5e-6 * max(1.0, abs(a)), (le, lg)
    for a, b in zip(pe, pg):
        assert (a - b).abs().max().item() <= 5e-6 * max(1.0, a.abs().max().item())


def test_batches_that_never_repeat_stay_eager():
    from vlsa_amd.losses import SurvObjective
    from vlsa_amd.optim import FusedAdam
    from vlsa_amd.train_step import TrainStep
    net, named = _model()
    ts = TrainStep(net, SurvObjective(), FusedAdam(_groups(named), lr=1e-3))
    g = torch.Generator().manual_seed(5)
    for i in range(5):
        bags = [cases.make_bag(300 + 17 * i + j, 9900 + 10 * i + j, "iid").to(torch.bfloat16).cuda() for j in range(4)]
        ts.step(bags, torch.randint(0, K - 1, (4,), generator=g).cuda(), torch.ones(4).cuda())
    d = ts.describe()
    assert d["captures"] == 0 and d["replays"] == 0 and d["eager_steps"] == 5, d


def _dp_worker(rank, world, port, ret):
    import os
    import torch.distributed as dist
    from vlsa_amd.losses import SurvObjective
    from vlsa_amd.optim import FusedAdam
    from vlsa_amd.train_step import TrainStep
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.cuda.set_device(0)
        net, named = _model()
        ts = TrainStep(net, SurvObjective(), FusedAdam(_groups(named), lr=1e-3), dist=dist, world=world)
        bags, t, e = _batches()[0]
        mine = list(range(len(bags)))[rank::world]
        losses = []
        for _ in range(4):
            loss = ts.step([bags[i] for i in mine], t[mine], e[mine])
            lt = loss.detach().cpu().clone()
            dist.all_reduce(lt)
            losses.append(float(lt) / world)
        ret[rank] = (losses, [p.detach().cpu().clone() for p in named])
    finally:
        dist.destroy_process_group()


def test_bag_parallel_steps_reproduce_the_single_process_steps():
    """TrainStep(dist=...): bags are the data-parallel unit, one flat gradient all-reduce per step (SURVEY.md 8(e) "Training DP").  Two
    ranks (gloo, both on this GPU) with half of the batch each must follow the single-process trajectory."""
    import os
    import torch.multiprocessing as mp
    from vlsa_amd.losses import SurvObjective
    from vlsa_amd.optim import FusedAdam
    from vlsa_amd.train_step import TrainStep
    net, named = _model()
    ts = TrainStep(net, SurvObjective(), FusedAdam(_groups(named), lr=1e-3), graph=False)
    bags, t, e = _batches()[0]
    ref = [float(ts.step(bags, t, e)) for _ in range(4)]
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_dp_worker, args=(2, 29700 + os.getpid() % 90, ret), nprocs=2, join=True)
    assert len(ret) == 2
    for r in range(2):
        losses, params = ret[r]
        for a, b in zip(losses, ref):
            assert abs(a - b) < 1e-4 * max(1.0, abs(b)), (r, losses, ref)
        for p, q in zip(params, named):
            assert (p - q.detach().cpu()).abs().max().item() < 1e-4 * max(1.0, q.abs().max().item())
    for p, q in zip(ret[0][1], ret[1][1]):
        assert torch.equal(p, q)                          # both ranks hold identical parameters after the all-reduce
