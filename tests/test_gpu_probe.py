"""Pins the hardware-layout assumptions the MFMA kernel is built on (run first on the GPU box)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_mfma_16x16x32_bf16_fragment_layout():
    from vlsa_amd import functional as F
    out = F.debug_probe(0).cpu()
    # A[i][k] = (i+1) if k == 5*(i%6) else 0 ; B[k][j] = 4k + j  ->  C[i][j] = (i+1) * (4*k_i + j)
    for lane in range(64):
        g, j = lane >> 4, lane & 15
        for r in range(4):
            i = 4 * g + r
            expect = (i + 1) * (4 * (5 * (i % 6)) + j)
            assert out[lane, r].item() == expect, (lane, r, out[lane, r].item(), expect)


def test_ds_read_b64_tr_b16_lane_map():
    from vlsa_amd import functional as F
    out = F.debug_probe(1).cpu()
    # lane (g, i) supplied the address of row 4g + (i>>2), cols 4(i&3)..; it must receive column i of rows 4g..4g+3
    for lane in range(64):
        g, i = lane >> 4, lane & 15
        for r in range(4):
            expect = (4 * g + r) * 16 + i
            assert out[lane, r].item() == expect, (lane, r, out[lane, r].item(), expect)
