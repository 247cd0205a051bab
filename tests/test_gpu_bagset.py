"""``vlsa_amd.functional.BagSet``: a list of resident bags checked ONCE, whose descriptor rows are kept -- ``forward_bags`` / the batched
autograd functions skip the per-bag validation and table building (round 4: for slide-sized bags that host work was most of a call).
Same numbers as a plain list, bit for bit; sub-sets; more than one launch; training; the look-ahead windows use it."""
import pytest
import torch

import cases
from oracle import vlsa_oracle as O

pytestmark = pytest.mark.gpu


def _net(P=12, K=5, pooling="mean", seed=811):
    from vlsa_amd.vlsa import VLSA
    params = cases.make_params(P, K, seed)
    cfg = dict(name="VLFAN", dim_in=512, use_feat_proj=False, query="Parameter", num_query=P, query_pooling=pooling)
    m = VLSA.from_modules(cfg, pretrained_text_features=params["T"].clone()).cuda()
    with torch.no_grad():
        m.mil_encoder.Q.copy_((0.5 * params["resid"] + params["prompt"]).cuda())
        m.mil_encoder.visual_adapter.weight.copy_(params["W"].cuda())
        m.mil_encoder.visual_adapter.bias.copy_(params["b"].cuda())
    return m, params


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
def test_bagset_equals_the_plain_list_in_eval(dtype):
    from vlsa_amd.functional import BagSet
    sizes = [2798, 17, 1, 900, 64, 65, 4100, 333] * 10                      # 80 bags: two launches (64 + 16)
    bags = [cases.make_bag(n, 8200 + i, "clustered" if i % 2 else "iid").to(dtype).cuda() for i, n in enumerate(sizes)]
    net, params = _net()
    net.eval()
    bs = BagSet(bags)
    assert isinstance(bs, list) and len(bs) == 80 and bs.sizes == tuple(sizes) and bs.rows.shape == (80, 3)
    with torch.no_grad():
        a = net.forward_bags(bags)
        b = net.forward_bags(bs)
        b2 = net.forward_bags(bs)                                             # chunks and their bags-in-flight choice are cached
        sub = net.forward_bags(bs.take([3, 70, 5]))
        subl = net.forward_bags([bags[3], bags[70], bags[5]])
        c, _, _, attn = net.forward_bags(bs.take(range(8)), ret_with_attn=True)
    for x, y in zip(a, b):
        assert torch.equal(x, y)
    assert torch.equal(b[0], b2[0]) and torch.equal(sub[0], subl[0])
    assert (c - a[0][:8]).abs().max().item() < 2e-5 and tuple(attn[0].shape) == (1, 12, sizes[0])
    Q = 0.5 * params["resid"] + params["prompt"]
    for i in (0, 2, 79):
        ref = O.vlsa_vlfan_forward(bags[i].float().cpu(), Q, params["T"], net.logit_scale.detach().cpu(), head_weight=params["W"],
                                   head_bias=params["b"])["logits"]
        assert (a[0][i:i + 1].cpu() - ref).abs().max().item() < 1e-4


def test_bagset_training_step_equals_the_plain_list():
    from vlsa_amd.functional import BagSet
    sizes = [700, 64, 1, 2798, 333, 4100, 65, 900, 17, 1200, 300, 300]
    bags = [cases.make_bag(n, 8300 + i, "clustered").to(torch.bfloat16).cuda() for i, n in enumerate(sizes)]
    G = torch.randn(len(sizes), 5, generator=cases.gen(8399)).cuda()
    grads = []
    for use_set in (False, True):
        net, _ = _net()
        net.train()
        arg = BagSet(bags).take(range(len(bags))) if use_set else bags
        logits = net.forward_bags(arg)[0]
        (logits * G).sum().backward()
        grads.append([logits.detach().clone()] + [p.grad.clone() for p in net.parameters() if p.grad is not None])
    assert len(grads[0]) == len(grads[1]) >= 4
    for x, y in zip(*grads):
        assert torch.equal(x, y)


def test_bagset_rejects_what_it_cannot_hold():
    from vlsa_amd._native import VlsaNativeError
    from vlsa_amd.functional import BagSet
    good = cases.make_bag(50, 1).to(torch.bfloat16).cuda()
    with pytest.raises(VlsaNativeError):
        BagSet([good, cases.make_bag(50, 2)])                                  # a CPU tensor
    with pytest.raises(VlsaNativeError):
        BagSet([good, cases.make_bag(50, 2).cuda()])                           # fp32 next to bf16
    with pytest.raises(VlsaNativeError):
        BagSet([good, good[:0]])                                               # an empty bag
    with pytest.raises(VlsaNativeError):
        BagSet([torch.zeros(10, 256, dtype=torch.bfloat16, device="cuda")])    # D != 512
    with pytest.raises(VlsaNativeError):
        BagSet([torch.zeros(10, 512, device="cuda", requires_grad=True)])


def test_resident_bags_and_arena_hand_out_bag_sets():
    from vlsa_amd.functional import BagSet
    from vlsa_amd.ingest import ArenaLayout, DeviceBagArena, ResidentBags

    class Items(torch.utils.data.Dataset):
        def __init__(self):
            self.x = [cases.make_bag(n, 8400 + n) for n in (70, 300, 1, 129)]

        def __len__(self):
            return len(self.x)

        def __getitem__(self, i):
            return torch.Tensor([i]).to(torch.int), (self.x[i], torch.Tensor([0])), torch.Tensor([1.0, 1.0])

    rb = ResidentBags(Items(), dtype=torch.float32)
    bs = rb.bag_set()                                                          # reads + uploads what is not resident yet
    assert isinstance(bs, BagSet) and bs.sizes == (70, 300, 1, 129) and rb.reads == 4
    assert all(torch.equal(v.cpu(), x) for v, x in zip(bs, rb.dataset.x))
    arena = DeviceBagArena(ArenaLayout.rows_needed([70, 300]), torch.device("cuda", 0))
    arena.add("a", rb.dataset.x[0]); arena.add("b", rb.dataset.x[1])
    s2 = arena.bag_set(["b", "a"])
    assert s2.sizes == (300, 70) and s2[0].dtype == torch.bfloat16
    net, _ = _net()
    net.eval()
    with torch.no_grad():
        assert torch.equal(net.forward_bags(bs)[0], net.forward_bags(list(bs))[0])
