"""Patch-sharded zero-shot (top-k / mean / max logit pooling) and DeepMIL (attention / gated attention / mean / max pooling
over N) with two ranks on ONE GPU over gloo (as tests/test_gpu_two_ranks.py): real kernels on both ranks, results must equal
the CPU oracle on the whole bag.  Includes an empty shard (a 5-patch bag on 2 ranks leaves rank 1 with nothing)."""
import os

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import cases

pytestmark = pytest.mark.gpu


def _worker(rank, world, port, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle import vlsa_oracle as O
        from vlsa_amd.deepmil import DeepMIL
        from vlsa_amd.sharded import shard_bounds, sharded_deepmil_forward, sharded_zeroshot_logits
        dev = torch.device("cuda", 0)
        torch.cuda.set_device(dev)
        errs = {}
        K = 4
        T = cases.make_params(1, K, 8600)["T"]
        ls = torch.tensor(cases.LOGIT_SCALE)
        for N, dt in ((9000, torch.bfloat16), (5000, torch.float32), (5, torch.float32), (40, torch.bfloat16)):
            X = cases.make_bag(N, 8610 + N).to(dt)
            a, c = shard_bounds(N, world, rank)
            for pooling in ("logit_top10", "logit_max", "logit_mean", "logit_top3"):
                got = sharded_zeroshot_logits(X[a:c].to(dev), T.to(dev), ls.to(dev), pooling, N)
                ref = O.vlsa_zeroshot_forward(X.float(), T, ls, pooling)[0]
                errs[f"zs {pooling} N={N}"] = (got.cpu() - ref).abs().max().item()
        for pooling in ("gated_attention", "attention", "mean", "max"):
            torch.manual_seed(8700)
            enc = DeepMIL(dim_in=512, dim_hid=256, num_cls=512, use_feat_proj=False, pooling=pooling, pred_head="Adapter").eval()
            pp = cases.make_pool_params(pooling, 8701)
            ad = cases.make_adapter_params(8702)
            with torch.no_grad():
                enc.visual_adapter.fc[0].weight.copy_(ad["down"]); enc.visual_adapter.fc[2].weight.copy_(ad["up"])
                sg = enc.sigma
                if pooling == "attention":
                    sg.attention[0].weight.copy_(pp["w1"]); sg.attention[0].bias.copy_(pp["b1"])
                    sg.attention[2].weight.copy_(pp["w2"]); sg.attention[2].bias.copy_(pp["b2"])
                elif pooling == "gated_attention":
                    sg.fc1[0].weight.copy_(pp["wa"]); sg.fc1[0].bias.copy_(pp["ba"])
                    sg.score[0].weight.copy_(pp["wg"]); sg.score[0].bias.copy_(pp["bg"])
                    sg.fc2.weight.copy_(pp["w2"]); sg.fc2.bias.copy_(pp["b2"])
            enc = enc.to(dev)
            for N, dt in ((6000, torch.bfloat16), (3000, torch.float32), (7, torch.float32)):
                X = cases.make_bag(N, 8710 + N, "clustered" if N > 100 else "iid").to(dt)
                a, c = shard_bounds(N, world, rank)
                v, attn = sharded_deepmil_forward(enc, X[a:c].to(dev)[None], ret_with_attn=True)
                r = O.deepmil_forward(X.float(), pooling, pp, pred_head="Adapter", adapter=(ad["down"], ad["up"]))
                errs[f"dm {pooling} N={N}"] = (v[0].cpu() - r["v"]).abs().max().item() / max(1.0, r["v"].abs().max().item())
                if attn is not None and c > a:
                    errs[f"dm attn {pooling} N={N}"] = (attn[0].cpu() - r["raw"][a:c]).abs().max().item()
        ret[rank] = errs
    finally:
        dist.destroy_process_group()


def test_two_ranks_zeroshot_and_deepmil():
    world = 2
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, 29400 + (os.getpid() % 150), ret), nprocs=world, join=True)
    assert len(ret) == world
    for r in range(world):
        for what, err in ret[r].items():
            assert err < 1e-4, (r, what, err)
    assert ret[0].keys() >= {"zs logit_top10 N=9000", "dm gated_attention N=6000"}
