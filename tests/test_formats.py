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
