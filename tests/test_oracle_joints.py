"""CPU: the joint-extraction oracle (oracle/joints.py) against fixtures produced by the REFERENCE's own
utils/cluster_utils.py, utils/mst_utils.py and sklearn's estimate_bandwidth (oracle/make_golden.py::joints_fixtures)."""
import numpy as np
import pytest

from conftest import load_golden
from oracle import joints as J


def _load(name):
    meta, a = load_golden(name)
    a = {k: v.numpy() for k, v in a.items()}
    vox = np.unpackbits(a["vox_data"])[:88 ** 3].reshape(88, 88, 88).astype(bool)
    return meta, a, (vox, meta["vox_translate"], meta["vox_scale"], meta["vox_dims"])


@pytest.mark.parametrize("name", ["joints_small", "joints_medium"])
def test_extraction_sequence(name):
    meta, a, vox = _load(name)
    pts, inside = J.inside_check(a["shifted"], *vox)
    assert np.array_equal(inside, a["index_inside"]) and 0 < len(inside) < len(a["shifted"])
    out = J.extract_joints(a["shifted"], a["attn_raw"], vox, meta["quantile"], meta["threshold1"], meta["threshold2"],
                           meta["max_iter"])
    assert out["bandwidth"] == pytest.approx(float(a["bandwidth"][0]), rel=1e-12)
    assert np.abs(out["modes"] - a["modes"]).max() <= 1e-12
    assert out["joints"].shape == a["joints"].shape and np.abs(out["joints"] - a["joints"]).max() <= 1e-12
    assert 2 <= len(a["joints"]) < len(a["modes"]) // 4          # a real clustering, not a pass-through


@pytest.mark.parametrize("name", ["joints_small", "joints_medium"])
def test_pieces(name):
    meta, a, _ = _load(name)
    bw = float(a["bandwidth"][0])
    assert J.estimate_bandwidth(a["pts_mirrored"], meta["quantile"]) == pytest.approx(bw, rel=1e-12)
    assert np.abs(J.meanshift_cluster(a["pts_mirrored"], bw, a["attn_mirrored"], 3) - a["modes_two_steps"]).max() <= 1e-13
    assert np.abs(J.meanshift_cluster(a["pts_mirrored"], bw, None, 5) - a["modes_unweighted"]).max() <= 1e-13
    kept, alive, counts = J.nms_meanshift(a["modes"], a["attn_mirrored"], bw, meta["threshold2"])
    assert np.array_equal(kept, a["joints_nms"])
    j, side = J.flip(a["joints_nms"])
    assert np.array_equal(j, a["joints"]) and np.array_equal(side, a["side"])


def test_known_answers():
    # flip: left kept, middle snapped to x = 0, right dropped and replaced by the mirror of left
    j, side = J.flip(np.array([[-0.3, 1, 2], [0.01, 3, 4], [0.4, 5, 6], [-0.02, 7, 8]]))
    assert j.tolist() == [[-0.3, 1, 2], [0.0, 3, 4], [0.0, 7, 8], [0.3, 1, 2]] and side.tolist() == [-1, 0, 0, 1]
    # bandwidth on a line: k = int(5 * 0.4) = 2 -> distance to the nearest other point
    x = np.array([[0.0, 0, 0], [1, 0, 0], [3, 0, 0], [6, 0, 0], [10, 0, 0]])
    assert J.estimate_bandwidth(x, 0.4) == pytest.approx((1 + 1 + 2 + 3 + 4) / 5)
    # mean-shift: two well separated pairs contract towards their midpoints, never across
    p = np.array([[0.0, 0, 0], [0.1, 0, 0], [5, 0, 0], [5.1, 0, 0]])
    m = J.meanshift_cluster(p, 0.5, None, max_iter=20)
    assert abs(m[0, 0] - 0.05) < 1e-2 and abs(m[1, 0] - 0.05) < 1e-2 and abs(m[2, 0] - 5.05) < 1e-2
    # nms: the dense pair survives as one point; an isolated, weakly attended point is dropped
    pts = np.array([[0.0, 0, 0], [0.01, 0, 0], [3.0, 0, 0]])
    attn = np.array([[0.9], [0.2], [0.1]], dtype=np.float32)
    kept, alive, counts = J.nms_meanshift(pts, attn, 0.1, thrd_density=0.5)
    assert counts.tolist() == [2, 2, 1] and alive.sum() == 1 and kept[0, 0] in (0.0, 0.01)
