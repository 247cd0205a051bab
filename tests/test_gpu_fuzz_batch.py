"""Randomised cross-check of every persistent multi-bag launch against its single-bag counterpart (tools/fuzz_batch.py): random
bag counts and sizes around the tile / unit boundaries, both dtypes, P up to 16, gated queries, with and without attention
weights, forward and backward; zero-shot pooling and the DeepMIL score + pooling launches."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("seed", [101, 202])
def test_fuzz_batched_launches_against_single_bag_kernels(seed):
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "fuzz_batch.py"), "60", str(seed)], capture_output=True,
                       text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "fuzz ok:" in r.stdout and "fuzz ok (other encoders):" in r.stdout and "fuzz ok (wide launches):" in r.stdout
