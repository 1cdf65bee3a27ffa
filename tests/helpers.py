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


def excess(a, b, atol, rtol):
    """max over elements of |a-b| - (atol + rtol*|b|); <= 0 means within tolerance."""
    a = a.detach().cpu().double()
    b = b.detach().cpu().double()
    return ((a - b).abs() - (atol + rtol * b.abs())).max().item()


def rel_excess(a, b, tol):
    """|a-b|_inf relative to max(1, |b|_inf), minus tol: the parity criterion used for network outputs
    ("within 1e-4 fp32" of BASELINE.json, normalised by the output scale when that exceeds 1)."""
    a = a.detach().cpu().double()
    b = b.detach().cpu().double()
    return (a - b).abs().max().item() / max(1.0, b.abs().max().item()) - tol
