"""TEST INFRASTRUCTURE ONLY -- build-container use (golden generation); never on the GPU box.

Lets the reference's *own* ``models/*.py`` import and run unmodified on CPU by placing
pure-torch stand-ins (oracle/pyg_primitives.py) for exactly the third-party symbols they import
into ``sys.modules``:

    torch_scatter.{scatter_max, scatter_add}                         models/rignet.py:3, deformnet.py:5
    torch_geometric.nn.conv.MessagePassing                           models/basic_modules.py:4
    torch_geometric.utils.{remove_self_loops, add_self_loops}        models/basic_modules.py:5
    torch_geometric.nn.{knn_interpolate, fps, radius, global_max_pool,
                        PointConv, knn}                              models/basic_modules.py:6, corrnet.py:3
    torch_geometric.data.{Data, InMemoryDataset}                     datasets/dataset_rig.py:6

Nothing of the reference is copied; it is imported from /root/reference where it lies.
"""
from __future__ import annotations

import contextlib
import importlib
import sys
import types

import torch

from . import pyg_primitives as P

REFERENCE_ROOT = "/root/reference"


def install() -> None:
    def mod(name):
        m = types.ModuleType(name)
        m.__path__ = []          # mark as package so submodule imports resolve
        sys.modules[name] = m
        return m

    ts = mod("torch_scatter")
    ts.scatter_max, ts.scatter_add = P.scatter_max, P.scatter_add
    tg = mod("torch_geometric")
    tgn = mod("torch_geometric.nn")
    tgc = mod("torch_geometric.nn.conv")
    tgu = mod("torch_geometric.utils")
    tg.nn, tg.utils, tgn.conv = tgn, tgu, tgc
    tgc.MessagePassing = P.MessagePassing
    tgu.remove_self_loops, tgu.add_self_loops = P.remove_self_loops, P.add_self_loops
    for name in ("knn_interpolate", "fps", "radius", "global_max_pool", "PointConv", "knn"):
        setattr(tgn, name, getattr(P, name))
    # torch_geometric.data.{Data, InMemoryDataset}: datasets/dataset_rig.py:6
    tgd = mod("torch_geometric.data")
    tg.data = tgd
    tgd.Data, tgd.InMemoryDataset, tgd.Dataset = P.Data, P.InMemoryDataset, P.InMemoryDataset


def import_reference_models():
    """Returns the reference ``models`` package (factories via ``models.__dict__[arch]``)."""
    install()
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)
    # our repo also has packages called `models`-like names nowhere on sys.path root, so this is
    # unambiguous: /root/reference/models
    return importlib.import_module("models")


@contextlib.contextmanager
def pretend_cuda_available():
    """The reference picks its GPU branch (deterministic ``radius``, cosine ``knn``) with
    ``torch.cuda.is_available()`` (models/basic_modules.py:76, models/corrnet.py:63). The product
    replaces that branch, so goldens are generated with the check forced to True while all tensors
    stay on CPU."""
    orig = torch.cuda.is_available
    torch.cuda.is_available = lambda: True
    try:
        yield
    finally:
        torch.cuda.is_available = orig
