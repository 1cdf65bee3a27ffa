"""The caller-side pieces of the reference's eval loop that sit right after the hot path
(training/train_rig.py:198-268, training/train_skin.py:185-254): post-ops and the two file
formats evaluate/eval_rigging.py reads back. Host-side, not on the timed path."""
from __future__ import annotations

import os

import numpy as np
import torch


def joint_positions(pred_shift: torch.Tensor, pos: torch.Tensor) -> torch.Tensor:
    """y_pred = tanh(pred_shift) + pos   (training/train_rig.py:224-225)"""
    return torch.tanh(pred_shift) + pos


def attention_probability(pred_mask: torch.Tensor) -> torch.Tensor:
    """sigmoid of the MaskNet logits (training/train_rig.py:253)"""
    return torch.sigmoid(pred_mask)


def skin_probability(skin_cls_pred: torch.Tensor, loss_mask: torch.Tensor = None) -> torch.Tensor:
    """softmax over the nearest bones, masked (training/train_skin.py:235-236)"""
    p = torch.softmax(skin_cls_pred, dim=1)
    return p if loss_mask is None else p * loss_mask


def ply_bytes(points) -> bytes:
    """ASCII .ply exactly as utils/io_utils.py:41-55 writes it (7 header lines, '%f %f %f')."""
    pts = points.detach().cpu().numpy() if torch.is_tensor(points) else np.asarray(points)
    lines = ["ply", "format ascii 1.0", "element vertex %d" % pts.shape[0],
             "property float x", "property float y", "property float z", "end_header"]
    lines += ["%f %f %f" % (p[0], p[1], p[2]) for p in pts]
    return ("\n".join(lines) + "\n").encode()


def write_eval_outputs(folder: str, names, batch: torch.Tensor, y_pred: torch.Tensor = None,
                       attn: torch.Tensor = None) -> None:
    """per-mesh ``{name}.ply`` (shifted points) and ``{name}_attn.npy`` (sigmoid mask, V x 1), the
    files evaluate/eval_rigging.py:53-76 consumes."""
    os.makedirs(folder, exist_ok=True)
    for i, name in enumerate(names):
        sel = batch == i
        if y_pred is not None:
            with open(os.path.join(folder, f"{int(name)}.ply"), "wb") as f:
                f.write(ply_bytes(y_pred[sel]))
        if attn is not None:
            np.save(os.path.join(folder, f"{int(name)}_attn.npy"), attn[sel].detach().cpu().numpy())
