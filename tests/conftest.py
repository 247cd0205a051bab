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
    rows = sorted(cases.GRAD_LOG, key=lambda r: -(r[2] / max(r[3], 1e-30)))
    with open(out, "w") as f:
        f.write("# observed gradient errors, worst first: rel = max|got - ref| / max|ref| per tensor (gate = absolute bound the test applied)\n")
        f.write(f"# {len(rows)} comparisons\n")
        f.write(f"{'rel':>10s} {'abs err':>10s} {'max|ref|':>10s} {'gate':>10s}  tensor @ test\n")
        for tid, what, err, ref, tol in rows:
            f.write(f"{err / max(ref, 1e-30):10.2e} {err:10.2e} {ref:10.2e} {tol:10.2e}  {what} @ {tid}\n")
