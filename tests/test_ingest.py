"""Bag ingest: arena bookkeeping on the host; on the GPU the uploaded bags must equal the reference's
host-side ``torch.cat(slides).to(float)`` (dataset/PatchWSI.py:214) rounded once to bf16, and feed forward_bags."""
import os

import pytest
import torch

import cases


def test_arena_layout_bookkeeping():
    from vlsa_amd.ingest import ArenaLayout
    lay = ArenaLayout(1000, align=64)
    assert lay.reserve("a", 100) == 0
    assert lay.reserve("b", 0) == 128          # empty bag takes no rows, next bag still tile-aligned
    assert lay.reserve("c", 64) == 128
    assert lay.reserve("d", 1) == 192
    assert lay.rows_free() == 1000 - 256
    assert "a" in lay and "z" not in lay and len(lay) == 4
    with pytest.raises(KeyError):
        lay.reserve("a", 3)
    with pytest.raises(MemoryError):
        lay.reserve("e", 745)
    assert lay.reserve("e", 744) == 256        # exactly to the end
    assert lay.rows_free() == 0
    assert ArenaLayout.rows_needed([100, 0, 64, 1]) == 128 + 0 + 64 + 64
    lay.reset()
    assert len(lay) == 0 and lay.reserve("a", 10) == 0


def test_read_patch_data_formats(tmp_path):
    import numpy as np
    from vlsa_amd.ingest import read_patch_data
    x = torch.randn(5, 512)
    torch.save(x, os.path.join(tmp_path, "s.pt"))
    np.save(os.path.join(tmp_path, "s.npy"), x.numpy())
    assert torch.equal(read_patch_data(os.path.join(tmp_path, "s.pt")), x)
    assert torch.equal(read_patch_data(os.path.join(tmp_path, "s.npy")), x)
    with pytest.raises(ValueError):
        read_patch_data(os.path.join(tmp_path, "s.h5x"))


def test_arena_refuses_cpu():
    from vlsa_amd.ingest import DeviceBagArena
    from vlsa_amd._native import VlsaNativeError
    with pytest.raises(VlsaNativeError):
        DeviceBagArena(128, "cpu")


@pytest.mark.gpu
@pytest.mark.parametrize("convert", ["device", "host"])
def test_arena_upload_equals_host_concat(convert, tmp_path):
    from vlsa_amd.ingest import DeviceBagArena
    dev = torch.device("cuda", 0)
    arena = DeviceBagArena(40_000, dev, chunk_rows=1000, convert=convert)   # small chunks: many staging round trips
    patients = {"p0": [2798], "p1": [1, 999, 1000, 1001], "p2": [5000, 3], "p3": [64], "p4": [7000]}
    host = {}
    for i, (pid, sizes) in enumerate(patients.items()):
        slides = [cases.make_bag(n, 1000 + 10 * i + j) * (3.0 if j == 1 else 1.0) for j, n in enumerate(sizes)]
        if pid == "p2":                            # one slide already stored as bf16, one as fp16
            slides = [slides[0].to(torch.bfloat16), slides[1].to(torch.float16)]
        if pid == "p4":                            # special values survive the cast
            slides[0][5, 7], slides[0][6, 0], slides[0][7, 1] = float("inf"), -float("inf"), 0.0
        if pid == "p3":                            # through the file reader
            torch.save(slides[0], os.path.join(tmp_path, "s.pt"))
            arena.add_files(pid, [os.path.join(tmp_path, "s.pt")])
        else:
            arena.add(pid, slides if len(slides) > 1 else slides[0])
        host[pid] = torch.cat([s.to(torch.float) for s in slides]).to(torch.bfloat16)
    assert len(arena) == 5 and "p1" in arena
    for pid, ref in host.items():
        got = arena.bag(pid)
        assert got.dtype == torch.bfloat16 and got.shape == ref.shape
        assert got.data_ptr() % (64 * 512 * 2) == arena.data.data_ptr() % (64 * 512 * 2)   # tile-aligned start
        assert torch.equal(got.cpu().view(torch.int16), ref.view(torch.int16)), pid         # bit-exact
    with pytest.raises(KeyError):
        arena.add("p0", torch.zeros(4, 512))
    with pytest.raises(MemoryError):
        arena.add("big", torch.zeros(40_000, 512))


@pytest.mark.gpu
def test_fp32_arena_keeps_features_bit_exact():
    from vlsa_amd.ingest import DeviceBagArena
    dev = torch.device("cuda", 0)
    arena = DeviceBagArena(8192, dev, chunk_rows=512, dtype=torch.float32)
    slides = [cases.make_bag(700, 1200), cases.make_bag(1300, 1201).to(torch.float16), cases.make_bag(5, 1202).to(torch.bfloat16)]
    got = arena.add("p", slides)
    arena.wait()
    assert got.dtype == torch.float32 and torch.equal(got.cpu(), torch.cat([s.to(torch.float) for s in slides]))


@pytest.mark.gpu
def test_arena_feeds_forward_bags():
    from vlsa_amd.ingest import DeviceBagArena
    from vlsa_amd.vlsa import VLSA
    dev = torch.device("cuda", 0)
    P, K = 12, 4
    params = cases.make_params(P, K, 1100)
    sizes = [2798, 130, 4000, 64, 1]
    arena = DeviceBagArena(ArenaLayout_rows(sizes), dev, chunk_rows=2048)
    hosts = [cases.make_bag(n, 1110 + i) for i, n in enumerate(sizes)]
    for i, x in enumerate(hosts):
        arena.add(i, x)
    cfg = dict(name="VLFAN", dim_in=512, use_feat_proj=False, query="Parameter", num_query=P, query_pooling="mean")
    m = VLSA.from_modules(cfg, pretrained_text_features=params["T"].clone()).to(dev).eval()
    with torch.no_grad():
        m.mil_encoder.Q.copy_((0.5 * params["resid"] + params["prompt"]).to(dev))
        m.mil_encoder.visual_adapter.weight.copy_(params["W"].to(dev))
        m.mil_encoder.visual_adapter.bias.copy_(params["b"].to(dev))
        got = torch.cat([m.forward_bags(b)[0] for b in arena.batches(list(range(len(sizes))), 3)])
        ref = torch.cat([m(x.to(torch.bfloat16).to(dev)[None])[0] for x in hosts])
    assert (got - ref).abs().max().item() < 1e-4


def ArenaLayout_rows(sizes):
    from vlsa_amd.ingest import ArenaLayout
    return ArenaLayout.rows_needed(sizes)


class _FakePatchDataset(torch.utils.data.Dataset):
    """items shaped like WSIPatchSurv's 'patch' mode (dataset/PatchWSI.py:197-215)"""

    def __init__(self, sizes, seed=0):
        g = torch.Generator().manual_seed(seed)
        self.feats = [torch.randn(n, 512, generator=g) for n in sizes]
        self.uid = [f"p{i}" for i in range(len(sizes))]
        self.hits = 0

    def __len__(self):
        return len(self.feats)

    def __getitem__(self, i):
        self.hits += 1
        return torch.Tensor([i]).to(torch.int), (self.feats[i].to(torch.float), torch.Tensor([0])), torch.Tensor([float(i), 1.0]).to(torch.float)


def test_resident_bags_passes_items_through_in_a_loader_worker(monkeypatch):
    from vlsa_amd.ingest import ResidentBags
    ds = _FakePatchDataset([5, 9])
    rb = ResidentBags(ds, device="cuda")              # no device is touched until an item is uploaded
    assert len(rb) == 2 and rb.uid == ["p0", "p1"]
    monkeypatch.setattr(torch.utils.data, "get_worker_info", lambda: object())
    idx, (feats, extra), label = rb[1]
    assert not feats.is_cuda and torch.equal(feats, ds.feats[1]) and rb.reads == 0 and rb.resident_bytes() == 0


@pytest.mark.gpu
def test_resident_bags_uploads_once_and_feeds_the_handlers_loop():
    """Two epochs of the handler's loader loop (runner/vlsa_handler.py:195-205: batch_size 1, default collate, ``data_x[0].cuda()``):
    the wrapped dataset is read once per item, the features arrive on the device as the bf16 rounding of the source."""
    from vlsa_amd.ingest import ResidentBags
    sizes = [70, 1, 3000, 257]
    ds = _FakePatchDataset(sizes)
    rb = ResidentBags(ds, segment_rows=2048)          # forces a second segment and an oversized bag of its own
    loader = torch.utils.data.DataLoader(rb, batch_size=1, shuffle=True, num_workers=0, generator=torch.Generator().manual_seed(1))
    for epoch in range(2):
        seen = set()
        for data_idx, data_x, data_y in loader:
            X = data_x[0].cuda()
            i = int(data_idx[0, 0])
            seen.add(i)
            assert X.is_cuda and X.dtype == torch.bfloat16 and tuple(X.shape) == (1, sizes[i], 512)
            assert torch.equal(X[0].cpu(), ds.feats[i].to(torch.bfloat16))
            assert float(data_y[0, 0]) == float(i) and data_x[1].shape == (1, 1)
        assert seen == set(range(len(sizes)))
        assert ds.hits == len(sizes) and rb.reads == len(sizes)      # epoch 2 never touched the wrapped dataset
    assert len(rb._segments) >= 2
    fp = ResidentBags(_FakePatchDataset(sizes), dtype=torch.float32)
    assert torch.equal(fp[2][1][0].cpu(), fp.dataset.feats[2])       # fp32: bit for bit
