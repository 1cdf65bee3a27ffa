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
