"""Shared helpers for tests (CPU and GPU)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from morig_amd.synth import MeshData  # noqa: E402


def data_from(arrs, device="cpu"):
    d = MeshData()
    for k in ("pos", "tpl_edge_index", "geo_edge_index", "batch", "pred_flow", "skin_input", "pts", "pts_batch"):
        if k in arrs:
            setattr(d, k, arrs[k].to(device))
    if "pos" in arrs:
        d.vtx, d.vtx_batch = d.pos, d.batch
    return d


def maxdiff(a, b):
    return (a.detach().cpu().double() - b.detach().cpu().double()).abs().max().item()
