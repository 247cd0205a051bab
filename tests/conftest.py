import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    try:
        import torch
        has_gpu = torch.cuda.is_available()
    except Exception:  # pragma: no cover
        has_gpu = False
    if has_gpu:
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


def pytest_sessionfinish(session, exitstatus):
    """VLSA_GRAD_ERRORS_OUT=<file>: the observed error of every gradient comparison of the session (tests/golden/cases.py
    ``record_grad_error``), relative to the largest entry of the reference gradient -- the figure the gates are set from."""
    out = os.environ.get("VLSA_GRAD_ERRORS_OUT")
    if not out:
        return
    try:
        import cases
    except Exception:  # pragma: no cover
        return
    write_grad_table(cases.GRAD_LOG, out)


def write_grad_table(log, out):
    """two sections: gradients with a non-zero reference, worst relative error first; then the tensors whose reference gradient is
    (numerically) zero -- softmax-invariant biases, one-patch bags -- where only the absolute error means anything"""
    real = sorted((r for r in log if r[3] > 1e-5), key=lambda r: -(r[2] / r[3]))
    zero = sorted((r for r in log if r[3] <= 1e-5), key=lambda r: -r[2])
    with open(out, "w") as f:
        f.write(f"# observed gradient errors of the GPU suite: {len(log)} comparisons (tests/golden/cases.py::record_grad_error)\n")
        f.write("# rel = max|got - ref| / max|ref| per tensor; gate = the absolute bound the test applied (1e-4 x max|ref| + 1e-5 where fixtures are\n")
        f.write("# compared; a bf16 bag's dX is a bf16 tensor: half an ulp = 3.9e-3)\n")
        big = [r for r in real if r[3] >= 1e-2 and not (r[4] > 1e-3 * r[3] + 2e-5)]      # fp32 results (not the bf16 dX rows), max|ref| >= 1e-2
        if big:
            w = max(big, key=lambda r: r[2] / r[3])
            f.write(f"# worst relative error among the {len(big)} fp32 gradients with max|ref| >= 1e-2: {w[2] / w[3]:.2e} ({w[1]} @ {w[0]})\n")
            f.write("# (the rows above it in the table: bf16 dX tensors -- rounding of their own dtype -- and fixtures whose whole gradient is < 1e-3, "
                    "where the 1e-5 absolute term of the gate decides)\n")
        f.write(f"\n## {len(real)} tensors with a non-zero reference gradient, worst first\n")
        f.write(f"{'rel':>10s} {'abs err':>10s} {'max|ref|':>10s} {'gate':>10s}  tensor @ test\n")
        for tid, what, err, ref, tol in real:
            f.write(f"{err / ref:10.2e} {err:10.2e} {ref:10.2e} {tol:10.2e}  {what} @ {tid}\n")
        f.write(f"\n## {len(zero)} tensors whose reference gradient is zero up to rounding (max|ref| <= 1e-5): absolute errors\n")
        f.write(f"{'abs err':>10s} {'max|ref|':>10s} {'gate':>10s}  tensor @ test\n")
        for tid, what, err, ref, tol in zero:
            f.write(f"{err:10.2e} {ref:10.2e} {tol:10.2e}  {what} @ {tid}\n")
