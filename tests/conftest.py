import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


def pytest_collection_modifyitems(config, items):
    # GPU tests are skipped (not failed) when no device is visible and they were
    # not explicitly selected with -m gpu.
    try:
        import torch
        have = torch.cuda.is_available()
    except Exception:
        have = False
    if have:
        return
    skip = pytest.mark.skip(reason="no GPU visible")
    for it in items:
        if "gpu" in it.keywords:
            it.add_marker(skip)


def pytest_sessionfinish(session, exitstatus):
    """Measured err_build / err_ref of every parity comparison of a GPU session (tests/_util.within_ref)."""
    try:
        import _util
    except Exception:
        return
    if not _util.RATIO_LOG:
        return
    out = os.path.join(ROOT, "gpurun_out")
    try:
        os.makedirs(out, exist_ok=True)
        with open(os.path.join(out, "parity_ratios.txt"), "w") as f:
            f.write("# label  err_build  err_ref  err_build/err_ref   (bound: err_build <= 2 err_ref + 1e-6)\n")
            for label, eb, er in _util.RATIO_LOG:
                f.write(f"{label}  {eb:.3e}  {er:.3e}  {eb / er if er > 0 else float('inf'):.2f}\n")
    except OSError:
        pass
