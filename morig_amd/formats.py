"""On-disk formats either side of the hot path. Host-side Python, as in the reference.

IN  -- what ``datasets/dataset_rig.py:78-140`` (``RigDataset.process``) reads per model ``{id}`` and how it turns the files
       into the tensors the networks consume:
         {id}_vtx_traj.npy  V x T x 3 trajectory; pos = frame 0; gt_flow = frames 20,40,..,100 minus frame 0   (:105-108)
         {id}_attn.txt      V attention ground truth                                                           (:83)
         {id}_tpl_e.txt / {id}_geo_e.txt   rows "src dst", np.loadtxt(...).T, + one self loop per vertex        (:84-85,119-122)
         {id}_rig.txt       joints / root / skin / hier lines, utils/rig_parser.py:21-45 (``Rig``)
         {id}_skin.txt      bones / bind / influence lines, dataset_rig.py:30-76 (``load_skin``)
         pred_flow/{id}_{1..5}_pred_flow.npy   DeformNet's output per keyframe, concatenated to V x 15           (:111-115)
OUT -- ``{id}.ply`` (ASCII, 7 header lines, '%f %f %f'; utils/io_utils.py:29-41, read back by ``readPly`` :18-26) and
       ``{id}_attn.npy``: writers in morig_amd/harness.py, reader here.
"""
from __future__ import annotations

import os
from typing import List

import numpy as np
import torch

from .synth import MeshData

NUM_NEAREST_BONE = 20          # dataset_rig.py:80
NUM_MAX_JOINT = 48             # dataset_rig.py:81


def read_ply(filename: str) -> np.ndarray:
    """utils/io_utils.py:18-26: skips the 7 header lines, float64 [n, 3]."""
    with open(filename, "r") as f:
        lines = f.readlines()
    return np.array([[float(w) for w in li.split()[:3]] for li in lines[7:]], dtype=np.float64).reshape(-1, 3)


class Rig:
    """utils/rig_parser.py:4-80: names, pos (after the reference's forward-kinematics pass with identity frames),
    hierarchy (parent id, -1 for the root), skins (V x J), root_id."""

    def __init__(self, filename: str):
        self.names: List[str] = []
        pos, skins = [], []
        self.hierarchy = None
        self.root_name, self.root_id = None, None
        with open(filename, "r") as f:
            for line in f.readlines():
                w = line.split()
                if w[0] == "joints":
                    self.names.append(w[1])
                    pos.append(np.array([float(w[2]), float(w[3]), float(w[4])]))
                elif w[0] == "root":
                    self.root_name = w[1]
                    self.root_id = self.names.index(w[1])
                    self.hierarchy = np.zeros(len(self.names), dtype=int)
                    self.hierarchy[self.root_id] = -1
                elif w[0] == "skin":
                    row = np.zeros(len(self.names))
                    for i in range(2, len(w), 2):
                        row[self.names.index(w[i])] = float(w[i + 1])
                    skins.append(row)
                elif w[0] == "hier":
                    self.hierarchy[self.names.index(w[2])] = self.names.index(w[1])
        pos = np.stack(pos, axis=0)
        self.skins = np.stack(skins, axis=0) if skins else []
        # calc_frames_and_offsets + FK (:47-78) with identity local frames: positions are rebuilt parent-first as
        # offset + parent position -- kept because (p - parent) + parent is not always p in floating point
        offset = np.zeros((len(self.names), 3))
        for i in range(len(pos)):
            offset[i] = pos[i] - pos[self.hierarchy[i]] if i != self.root_id else pos[i]
        res = np.zeros_like(pos)
        res[self.root_id] = pos[self.root_id]
        frontier = [self.root_id]
        eye = np.eye(3)
        while frontier:
            nxt = []
            for j in range(len(self.names)):
                if self.hierarchy[j] in frontier:
                    res[j] = np.matmul(eye, offset[j][:, None]).squeeze(axis=1) + res[self.hierarchy[j]]
                    nxt.append(j)
            frontier = nxt
        self.offset = offset
        self.pos = res


def load_skin(filename: str, num_nearest_bone: int = NUM_NEAREST_BONE):
    """dataset_rig.py:30-76 -> (skin_input V x 8k, nearest_bone_ids V x k, label, loss_mask V x k, bone_names)."""
    bones, bone_names, inputs, labels, nn_ids, masks = [], [], [], [], [], []
    with open(filename, "r") as f:
        for li in f.readlines():
            w = li.strip().split()
            if w[0] == "bones":
                bone_names.append([w[1], w[2]])
                bones.append([float(x) for x in w[3:]])
            elif w[0] == "bind":
                v = [float(x) for x in w[1:]]
                row, ids, mask = [], [], []
                for i in range(num_nearest_bone):
                    valid = int(v[3 * i + 1]) != -1
                    b = 3 * i + 1 if valid else 1                     # invalid slot: repeat the nearest bone, masked out (:50-56)
                    ids.append(int(v[b]))
                    row += bones[int(v[b])]
                    row.append(v[b + 1])
                    row.append(int(v[b + 2]))
                    mask.append(1 if valid else 0)
                inputs.append(np.array(row)[np.newaxis, :])
                nn_ids.append(np.array(ids)[np.newaxis, :])
                masks.append(np.array(mask)[np.newaxis, :])
            elif w[0] == "influence":
                labels.append(np.array([float(x) for x in w[1:]])[np.newaxis, :])
    return (np.concatenate(inputs, axis=0), np.concatenate(nn_ids, axis=0), np.concatenate(labels, axis=0),
            np.concatenate(masks, axis=0), bone_names)


def _with_self_loops(e: torch.Tensor, n: int) -> torch.Tensor:
    """torch_geometric.utils.add_self_loops(e, num_nodes=n) as the dataset applies it (:121-122): existing loops stay."""
    loops = torch.arange(n, dtype=torch.long).unsqueeze(0).repeat(2, 1)
    return torch.cat([e, loops], dim=1)


def load_rig_sample(vtx_filename: str) -> MeshData:
    """One ``Data`` of RigDataset.process (dataset_rig.py:82-138) from ``.../{id}_vtx_traj.npy`` and its siblings."""
    root = os.path.dirname(vtx_filename)
    sib = lambda suffix: vtx_filename.replace("_vtx_traj.npy", suffix)
    v_traj = np.load(vtx_filename)
    m = np.loadtxt(sib("_attn.txt"))
    tpl_e = np.loadtxt(sib("_tpl_e.txt")).T
    geo_e = np.loadtxt(sib("_geo_e.txt")).T
    rig = Rig(sib("_rig.txt"))
    joints = rig.pos
    name = int(os.path.basename(vtx_filename).split("_")[0])
    first = v_traj[:, 0, :]
    nearest_jid = np.argmin(np.sum((joints[:, None, :] - first[None, ...]) ** 2, axis=-1), axis=0)
    offsets = joints[nearest_jid] - first
    gt_skin = np.zeros((rig.skins.shape[0], NUM_MAX_JOINT))
    gt_skin[:, 0:rig.skins.shape[1]] = rig.skins
    skin_input, skin_nn, skin_label, loss_mask, bone_names = load_skin(sib("_skin.txt"))
    skin_nnjids = np.stack([np.array([rig.names.index(bone_names[b][0]) for b in skin_nn[v]]) for v in range(len(v_traj))], 0)
    gt_flow = np.concatenate([v_traj[:, t, :] - first for t in np.arange(20, 110, 20)], axis=1)
    pred_flow = np.concatenate([np.load(os.path.join(root, f"pred_flow/{name}_{t}_pred_flow.npy")) for t in np.arange(1, 6)], axis=1)

    d = MeshData()
    d.pos = torch.from_numpy(first).float()
    n = d.pos.size(0)
    d.mask = torch.from_numpy(m).float()
    d.tpl_edge_index = _with_self_loops(torch.from_numpy(tpl_e).long(), n)
    d.geo_edge_index = _with_self_loops(torch.from_numpy(geo_e).long(), n)
    d.offsets = torch.from_numpy(offsets).float()
    d.gt_flow = torch.from_numpy(gt_flow).float()
    d.pred_flow = torch.from_numpy(pred_flow).float()
    d.joints = torch.from_numpy(joints).float()
    d.skin_input = torch.from_numpy(skin_input).float()
    d.skin_label = torch.from_numpy(skin_label).float()
    d.skin_nn = torch.from_numpy(skin_nn).long()
    d.skin_nnjids = torch.from_numpy(skin_nnjids).long()
    d.loss_mask = torch.from_numpy(loss_mask).long()
    d.gt_skin = torch.from_numpy(gt_skin).float()
    d.name = name
    d.batch = torch.zeros(n, dtype=torch.long)
    return d
