"""pytest configuration: the ``gpu`` marker and shared fixture helpers.

CPU suite  : python -m pytest tests/ -x -q -m "not gpu"   (oracle vs golden, host logic, C-ABI exports, gloo)
GPU suite  : python -m pytest tests/ -x -q -m gpu          (HIP path vs oracle / golden, through the C-ABI)
"""
import json
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run through gpurun / the driver)")


def pytest_collection_modifyitems(config, items):
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for it in items:
        if "gpu" in it.keywords:
            it.add_marker(skip)


def load_golden(name):
    """-> (meta dict, {key: torch tensor})"""
    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    meta = json.loads(bytes(z["meta"]).decode())
    arrs = {k: torch.from_numpy(z[k]) for k in z.files if k != "meta"}
    return meta, arrs


@pytest.fixture(autouse=True)
def _no_grad():
    with torch.no_grad():
        yield


def pytest_terminal_summary(terminalreporter):
    """parity report: worst absolute and scale-relative error per test (both readings of "within 1e-4 fp32")."""
    try:
        from helpers import NOTES, PARITY_LOG
    except Exception:
        return
    for line in NOTES:
        terminalreporter.write_line("note: " + line)
    if not PARITY_LOG:
        return
    worst = {}
    for tid, err, scale, rel in PARITY_LOG:
        w = worst.get(tid)
        if w is None or err > w[0]:
            worst[tid] = (err, scale, rel)
    tr = terminalreporter
    tr.write_sep("-", "parity report: max |diff| (absolute), |ref|inf, |diff| / max(1, |ref|inf)")
    for tid, (err, scale, rel) in sorted(worst.items(), key=lambda kv: -kv[1][0])[:25]:
        tr.write_line(f"{err:10.3e}  {scale:10.3e}  {rel:10.3e}  {tid[-110:]}")
