"""CPU-side checks of the drop-in boundary: the C-ABI library loads, exports every symbol the header
declares, and the product path refuses CPU tensors loudly (no fallback)."""
import os
import re

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_symbols():
    txt = open(os.path.join(ROOT, "include", "vlsa_hip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(vlsa_[a-z0-9_]+)\s*\(", txt)))


def test_library_builds_and_exports_every_declared_symbol():
    from vlsa_amd import build, _native
    build.build_native()
    lib = _native.load()
    declared = _header_symbols()
    assert declared, "no declarations parsed from include/vlsa_hip.h"
    for name in declared:
        assert hasattr(lib, name), f"libvlsa_hip.so does not export {name}"
    assert set(_native.exported_symbols()) == set(declared), "binding and header disagree"
    assert lib.vlsa_abi_version() == _native.ABI_VERSION


def test_host_side_queries_need_no_gpu():
    from vlsa_amd import _native
    lib = _native.load()
    assert lib.vlsa_num_partials(0) == 1
    assert lib.vlsa_num_partials(1) == 1
    assert lib.vlsa_num_partials(33) == 2
    assert lib.vlsa_num_partials(50_000) == 256
    assert lib.vlsa_qprep_bytes(512) == 16 * 512 * 4 + 3 * 16 * 512 * 2 + 17 * 512 * 4 + 128
    assert lib.vlsa_error_string(-1) == b"invalid argument"
    # tile geometry of the attention-score kernel (what the batched caller sizes its tile table with)
    import ctypes
    for dt in (_native.DT_BF16, _native.DT_F32):
        for gated in (0, 1):
            mr, rt = ctypes.c_int(0), ctypes.c_int(0)
            assert lib.vlsa_gated_scores_tiling(dt, gated, ctypes.addressof(mr), ctypes.addressof(rt)) == 0
            assert mr.value in (64, 128, 256) and rt.value >= 64
    assert lib.vlsa_gated_scores_tiling(99, 0, ctypes.addressof(mr), ctypes.addressof(rt)) != 0
    # the persistent LDS-DMA score kernel: where it applies, and the tiling of a single-bag scores + pooling launch (host arithmetic):
    # the tiles of 256 walkers cover the bag with heights of 16 .. 256 rows whose per-walker sums differ by at most one 16-row unit
    rows, mn = ctypes.c_int(0), ctypes.c_int64(0)
    assert lib.vlsa_gated_scores_big_tile(_native.DT_BF16, 1, ctypes.addressof(rows), ctypes.addressof(mn)) == 0
    assert rows.value in (0, 256) and (rows.value == 0 or mn.value >= 1)       # (0: switched off in a -DVLSA_EXPERIMENT build)
    assert lib.vlsa_gated_scores_big_tile(_native.DT_F32, 1, ctypes.addressof(rows), ctypes.addressof(mn)) == 0 and rows.value == 0
    assert lib.vlsa_gated_scores_pool_ws_floats(0) == 0
    for N in (1, 31, 32, 33, 8191, 8192, 8193, 16384, 50_000, 65_536, 65_537, 393_216, 400_000, 4_000_000):
        ws = lib.vlsa_gated_scores_pool_ws_floats(N)
        units = -(-N // (16 * 256))                 # 16-row units per walker
        rounds = -(-units // 16)
        lo, hi = (units // rounds) * 16, -(-units // rounds) * 16      # the two tile heights of the plan
        assert 16 <= lo <= hi <= 256
        assert ws >= 514 * -(-N // hi), (N, ws, hi)                  # one (m, l, acc[512]) record per tile of the bf16 route
        assert ws >= 544 * lib.vlsa_pool_num_partials(N) + 32        # (pm, pl, pacc) + (m2, l) of the chained fp32 route
        assert ws <= max(514 * min(-(-N // lo), rounds * 256), 544 * lib.vlsa_pool_num_partials(N) + 64)


def test_default_build_reads_no_environment():
    """SURVEY.md 8(b) / include/vlsa_hip.h: "no global state".  The A/B switches of the measurement tools (VLSA_GS_*, VLSA_TT_*,
    VLSA_MAX_PARTIALS, VLSA_EXP ...) exist only in a -DVLSA_EXPERIMENT build: the default library neither imports getenv nor carries
    any of their names."""
    import subprocess
    from vlsa_amd import build
    if os.environ.get("VLSA_EXTRA_HIPCC_FLAGS") or os.environ.get("VLSA_HIP_LIB"):
        pytest.skip("not the default build")
    lib = build.build_native()
    blob = open(lib, "rb").read()
    assert b"getenv" not in blob
    names = sorted(set(re.findall(rb"VLSA_[A-Z0-9_]{2,}", blob)))
    assert names == [], names
    nm = subprocess.run(["nm", "-D", "--undefined-only", lib], capture_output=True, text=True)
    if nm.returncode == 0:
        assert "getenv" not in nm.stdout


def test_cpu_tensor_is_refused_loudly():
    from vlsa_amd import VlsaNativeError, functional as F
    X = torch.randn(8, 512)
    Q = torch.randn(4, 512)
    with pytest.raises(VlsaNativeError):
        F.vlfan_aggregate(X, Q)
