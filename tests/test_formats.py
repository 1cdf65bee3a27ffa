"""CPU: on-disk formats either side of the path: morig_amd/formats.py against tensors produced by the
reference's own RigDataset.process / readPly on the same raw files (oracle/make_golden.py::dataset_fixtures)."""
import os

import numpy as np
import torch

from conftest import GOLDEN, load_golden
from morig_amd import formats, harness


def _lay_down(tmp_path):
    import json
    z = np.load(os.path.join(GOLDEN, "rig_dataset_files.npz"))
    meta = json.loads(bytes(z["meta"]).decode())
    blob = bytes(z["file_blob"])
    off = 0
    for name, size in zip(meta["file_names"], meta["file_sizes"]):
        path = os.path.join(tmp_path, name)
        os.makedirs(os.path.dirname(path), exist_ok=True)
        with open(path, "wb") as f:
            f.write(blob[off:off + size])
        off += size
    return meta, {k: torch.from_numpy(z[k]) for k in z.files if k.startswith("m")}


def test_rig_sample_matches_reference_loader(tmp_path):
    meta, want = _lay_down(str(tmp_path))
    for model in meta["models"]:
        d = formats.load_rig_sample(os.path.join(str(tmp_path), f"{model}_vtx_traj.npy"))
        keys = [k.split("__", 1)[1] for k in want if k.startswith(f"m{model}__")]
        assert sorted(keys) == sorted(["pos", "tpl_edge_index", "geo_edge_index", "pred_flow", "gt_flow", "mask", "joints", "offsets",
                                       "gt_skin", "skin_input", "skin_label", "skin_nn", "skin_nnjids", "loss_mask"])
        for k in keys:
            got, w = getattr(d, k), want[f"m{model}__{k}"]
            assert got.dtype == w.dtype and got.shape == w.shape, k
            assert torch.equal(got, w), k                       # same arithmetic in the same order: bit-identical
        assert d.name == model
        V = d.pos.shape[0]
        assert d.skin_input.shape == (V, 160) and d.pred_flow.shape == (V, 15) and d.gt_skin.shape[1] == 48
        assert int(d.loss_mask.min()) == 0 and int(d.loss_mask.max()) == 1          # the -1 slots of load_skin
        loops = d.tpl_edge_index[:, -V:]
        assert torch.equal(loops[0], torch.arange(V)) and torch.equal(loops[0], loops[1])


def test_loaded_sample_feeds_the_networks(tmp_path):
    """the loader's output is what the drop-in modules consume: collate two loaded models and run the oracle on them"""
    from morig_amd import synth
    from oracle import nets
    meta, _ = _lay_down(str(tmp_path))
    ds = [formats.load_rig_sample(os.path.join(str(tmp_path), f"{m}_vtx_traj.npy")) for m in meta["models"]]
    batch = synth.collate(ds)
    m = synth.load_recipe(nets.jointnet_motion(num_keyframes=5, chn_output=3, aggr_method="attn").eval(), 5, mild=True)
    out = m(batch, batch.pred_flow)[2]
    assert out.shape == (batch.pos.shape[0], 3) and bool(torch.isfinite(out).all())


def test_ply_round_trip(tmp_path):
    meta, a = load_golden("rig_dataset_files")
    p = os.path.join(str(tmp_path), "kat.ply")
    with open(p, "w") as f:
        f.write(meta["ply_text"])
    got = formats.read_ply(p)
    assert got.dtype == np.float64 and np.array_equal(got, a["ply_points"].numpy())
    # our writer's bytes read back to the values '%f' keeps
    pts = torch.tensor([[0.1, 0.25, -0.3], [1.0, 2.0, 3.0], [1e-7, -1e-7, 0.5]])
    with open(p, "wb") as f:
        f.write(harness.ply_bytes(pts))
    assert np.array_equal(formats.read_ply(p), got)


import pytest  # noqa: E402


def _pipeline(tmp_path, dev, run_hip):
    """files -> loader -> jointnet + masknet -> post-ops -> writers -> readers -> joint extraction, on device ``dev``."""
    from morig_amd import joints, models, synth
    from oracle import joints as ojoints, nets
    from helpers import rel_excess
    meta, _ = _lay_down(str(tmp_path))
    samples = [formats.load_rig_sample(os.path.join(str(tmp_path), f"{m}_vtx_traj.npy")) for m in meta["models"]]
    batch = synth.collate(samples)
    names = [int(n) for n in batch.name.tolist()]
    kwj = dict(num_keyframes=5, chn_output=3, aggr_method="attn")
    kwm = dict(num_keyframes=5, chn_output=1, aggr_method="attn")
    ours_j = synth.load_recipe(models.jointnet_motion(**kwj).eval(), 61, mild=True).to(dev)
    ours_m = synth.load_recipe(models.masknet_motion(**kwm).eval(), 62, mild=True).to(dev)
    ref_j = synth.load_recipe(nets.jointnet_motion(**kwj).eval(), 61, mild=True)
    ref_m = synth.load_recipe(nets.masknet_motion(**kwm).eval(), 62, mild=True)
    # stage 1: the networks (training/train_rig.py:213-226)
    d = batch.to(dev)
    shift, mask = ours_j(d, d.pred_flow)[2], ours_m(d, d.pred_flow)[2]
    w_shift, w_mask = ref_j(batch, batch.pred_flow)[2], ref_m(batch, batch.pred_flow)[2]
    assert rel_excess(shift, w_shift, 1e-4) <= 0 and rel_excess(mask, w_mask, 1e-4) <= 0
    # stage 2: post-ops + writers (training/train_rig.py:224-225, 253-263)
    out_hip, out_ref = os.path.join(str(tmp_path), "out_hip"), os.path.join(str(tmp_path), "out_ref")
    harness.write_eval_outputs(out_hip, names, batch.batch, y_pred=harness.joint_positions(shift, d.pos).cpu(),
                               attn=harness.attention_probability(mask).cpu())
    harness.write_eval_outputs(out_ref, names, batch.batch, y_pred=torch.tanh(w_shift) + batch.pos, attn=torch.sigmoid(w_mask))
    for name in names:
        # stage 3: readers (evaluate/eval_rigging.py:53-71)
        p_hip, p_ref = formats.read_ply(os.path.join(out_hip, f"{name}.ply")), formats.read_ply(os.path.join(out_ref, f"{name}.ply"))
        a_hip, a_ref = np.load(os.path.join(out_hip, f"{name}_attn.npy")), np.load(os.path.join(out_ref, f"{name}_attn.npy"))
        assert p_hip.shape == p_ref.shape and np.abs(p_hip - p_ref).max() <= 1e-4 + 2e-6       # '%f' keeps 6 decimals
        assert a_hip.shape == a_ref.shape == (p_hip.shape[0], 1) and np.abs(a_hip - a_ref).max() <= 1e-4
        # stage 4: joint extraction (evaluate/eval_rigging.py:72-95; no voxel grid in the synthetic fixture)
        want_same = ojoints.extract_joints(p_hip, a_hip)                                            # oracle on the SAME files
        got = joints.extract_joints(torch.from_numpy(p_hip), a_hip, device=torch.device(dev))
        assert got["joints"].shape == want_same["joints"].shape
        assert np.abs(got["joints"] - want_same["joints"]).max() <= 1e-9 and abs(got["bandwidth"] - want_same["bandwidth"]) <= 1e-12
        want_ref = ojoints.extract_joints(p_ref, a_ref)                                             # the oracle's own pipeline
        assert want_ref["joints"].shape == got["joints"].shape, "joint count differs between the two pipelines"
        assert np.abs(got["joints"] - want_ref["joints"]).max() <= 1e-3
    return len(names)


@pytest.mark.gpu
def test_files_to_hip_to_files_pipeline(tmp_path):
    """SURVEY 8 f-3 / configs[4] stand-in (the ModelsResources data and checkpoints are not available offline): raw dataset
    files -> formats.load_rig_sample -> HIP jointnet + masknet -> harness writers -> formats.read_ply / np.load -> HIP joint
    extraction, every stage against the oracle pipeline."""
    assert _pipeline(tmp_path, "cuda", True) == 2


def test_files_to_files_pipeline_host_wiring(tmp_path):
    """the same pipeline with the op layer emulated on CPU: host wiring only."""
    import sys
    sys.path.insert(0, os.path.dirname(__file__))
    from emulate import EmuOps
    from morig_amd import runtime
    runtime._test_ops = EmuOps()
    try:
        assert _pipeline(tmp_path, "cpu", False) == 2
    finally:
        runtime._test_ops = None
