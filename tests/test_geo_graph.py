"""The geodesic-ball graph build (/root/reference/data_proc/common_ops.py:214-226 get_geo_edges).

CPU: the numpy restatement (oracle/graph_build.py) against the fixture the reference's OWN function produced
(tests/golden/geo_edges_kat.npz, oracle/make_golden.py geo_edges): bit-exact, including the np.random.choice draws when it is
handed the same numpy stream.
GPU (-m gpu, through the C-ABI): morig_geo_ball_graph / _dist / _fill -- bit-exact wherever the reference is deterministic (rows
within the cap, in index order), and for over-full rows the properties np.random.choice(replace=False) guarantees: exactly
max_nn members, all inside the ball, no repeats, reproducible under the seed, uniform over the members.
"""
import numpy as np
import pytest
import torch

from conftest import load_golden
from oracle import graph_build as og


def _fixture():
    meta, a = load_golden("geo_edges_kat")
    return meta, {k: v.numpy() for k, v in a.items()}


def test_oracle_reproduces_the_reference_function_bit_for_bit():
    meta, a = _fixture()
    d2 = og.euclid_sq_f32(a["pos"]).astype(np.float64)
    r2 = float(np.float32(meta["r_euclid"]) * np.float32(meta["r_euclid"]))
    cases = ((a["dist"], meta["r_exact"], meta["max_exact"], 1, "edges_exact"), (a["dist"], meta["r_over"], meta["max_over"], 2, "edges_over"),
             (d2, r2, meta["max_euclid"], 3, "edges_euclid"), (d2, r2, meta["max_euclid_over"], 4, "edges_euclid_over"))
    for m, r, k, seed, key in cases:
        np.random.seed(seed)                                        # the reference draws from numpy's global stream
        got = og.get_geo_edges_from_distance(m, r, k)
        assert got.dtype == np.int64 and np.array_equal(got, a[key]), key
    # no self loops, rows ascending, never more than max_nn per row
    e = a["edges_over"]
    assert (e[:, 0] != e[:, 1]).all() and (np.diff(e[:, 0]) >= 0).all() and np.bincount(e[:, 0]).max() == meta["max_over"]
    assert np.array_equal(np.bincount(a["edges_exact"][:, 0], minlength=len(a["pos"])), a["counts_exact"])


def test_oracle_member_lists_agree_with_the_distance_form():
    meta, a = _fixture()
    lists = og.member_lists(a["pos"], [0, len(a["pos"])], meta["r_euclid"])
    e = a["edges_euclid"]
    for i, mem in enumerate(lists):
        assert np.array_equal(mem, e[e[:, 0] == i, 1])


def test_c_abi_declares_the_graph_build():
    import os, re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    hdr = open(os.path.join(root, "include", "morig_hip.h")).read()
    for sym in ("morig_geo_ball_graph", "morig_geo_ball_graph_dist", "morig_geo_ball_fill"):
        assert re.search(r"\bint\s+%s\s*\(" % sym, hdr), sym


# ------------------------------------------------------------------------------------------------------------------------------
DEV = "cuda"


def _rows(ei):
    """[2, E] device edge_index -> numpy [E, 2] rows [i, member] (the reference function's own layout)"""
    return ei.cpu().numpy().T


def _check_over(e, counts, mx, inside, n):
    """rows within the cap: all members in index order; over-full rows: the np.random.choice(replace=False) properties"""
    per_row = [e[e[:, 0] == i, 1] for i in range(n)]
    for i in range(n):
        got = per_row[i]
        if counts[i] <= mx:
            assert np.array_equal(got, np.flatnonzero(inside[i])), i
        else:
            assert len(got) == mx and len(set(got.tolist())) == mx and inside[i][got].all(), i
    assert (np.diff(e[:, 0]) >= 0).all()


@pytest.mark.gpu
def test_distance_variant_against_the_reference_fixture():
    from morig_amd import graph_build as gb
    meta, a = _fixture()
    n = len(a["pos"])
    dist = torch.from_numpy(a["dist"]).to(DEV)
    ei, members = gb.get_geo_edges_from_distance(dist, meta["r_exact"], meta["max_exact"], seed=5, return_members=True)
    assert ei.dtype == torch.int64 and ei.shape[0] == 2 and np.array_equal(_rows(ei), a["edges_exact"])
    assert np.array_equal(members.cpu().numpy(), a["counts_exact"])
    mx = meta["max_over"]
    e1, members = gb.get_geo_edges_from_distance(dist, meta["r_over"], mx, seed=11, return_members=True)
    assert np.array_equal(members.cpu().numpy(), a["counts_over"])
    inside = (a["dist"] + 10.0 * np.eye(n)) <= meta["r_over"]
    _check_over(_rows(e1), a["counts_over"], mx, inside, n)
    assert _rows(e1).shape == a["edges_over"].shape                                   # the same number of edges as the reference's draw
    assert np.array_equal(_rows(gb.get_geo_edges_from_distance(dist, meta["r_over"], mx, seed=11)), _rows(e1))     # reproducible
    assert not np.array_equal(_rows(gb.get_geo_edges_from_distance(dist, meta["r_over"], mx, seed=12)), _rows(e1))  # a new draw per seed
    torch.manual_seed(3)
    ea = _rows(gb.get_geo_edges_from_distance(dist, meta["r_over"], mx))
    torch.manual_seed(3)
    assert np.array_equal(_rows(gb.get_geo_edges_from_distance(dist, meta["r_over"], mx)), ea)                      # torch.manual_seed


@pytest.mark.gpu
def test_positions_variant_against_the_reference_fixture():
    from morig_amd import graph_build as gb
    meta, a = _fixture()
    n = len(a["pos"])
    pos = torch.from_numpy(a["pos"]).to(DEV)
    ei, members = gb.get_geo_edges(pos, None, meta["r_euclid"], meta["max_euclid"], seed=1, return_members=True)
    assert np.array_equal(_rows(ei), a["edges_euclid"]) and np.array_equal(members.cpu().numpy(), a["counts_euclid"])
    mx = meta["max_euclid_over"]
    e1 = _rows(gb.get_geo_edges(pos, None, meta["r_euclid"], mx, seed=2))
    r2 = np.float32(meta["r_euclid"]) * np.float32(meta["r_euclid"])
    inside = og.euclid_sq_f32(a["pos"]) <= r2
    np.fill_diagonal(inside, False)
    _check_over(e1, a["counts_euclid"], mx, inside, n)
    assert e1.shape == a["edges_euclid_over"].shape
    # self loops appended last, as add_self_loops does (datasets/dataset_rig.py:122)
    e2 = _rows(gb.get_geo_edges(pos, None, meta["r_euclid"], meta["max_euclid"], seed=1, self_loops=True))
    assert np.array_equal(e2[:-n], a["edges_euclid"]) and np.array_equal(e2[-n:], np.stack([np.arange(n)] * 2, 1))


@pytest.mark.gpu
@pytest.mark.parametrize("sizes", [[100, 1500, 37, 1, 2200], [1], [15, 17], [3000]])
def test_batched_ragged_meshes_against_the_oracle(sizes):
    """meshes of different sizes in one launch: blocks whose centres straddle two meshes, meshes longer than one LDS tile, a
    one-vertex mesh (no members), never an edge across meshes"""
    from morig_amd import graph_build as gb
    rng = np.random.default_rng(sum(sizes))
    pos = (rng.random((sum(sizes), 3)) * 0.6).astype(np.float32)
    ptr = np.concatenate([[0], np.cumsum(sizes)])
    batch = torch.from_numpy(np.repeat(np.arange(len(sizes)), sizes)).to(DEV)
    r = 0.08
    lists = og.member_lists(pos, ptr, r)
    ei, members = gb.get_geo_edges(torch.from_numpy(pos).to(DEV), batch, r, 64, seed=9, return_members=True, num_graphs=len(sizes))
    e = _rows(ei)
    assert max(len(m) for m in lists) <= 64
    want = np.concatenate([np.stack([np.full(len(m), i), m], 1) for i, m in enumerate(lists)]).astype(np.int64)
    assert np.array_equal(e, want)
    assert np.array_equal(members.cpu().numpy(), [len(m) for m in lists])
    mesh_of = np.repeat(np.arange(len(sizes)), sizes)
    assert (mesh_of[e[:, 0]] == mesh_of[e[:, 1]]).all() if len(e) else True
    # the capped form on the same batch
    mx = 4
    e4 = _rows(gb.get_geo_edges(torch.from_numpy(pos).to(DEV), batch, r, mx, seed=10, num_graphs=len(sizes)))
    inside = np.zeros((len(pos), len(pos)), bool)
    for i, m in enumerate(lists):
        inside[i, m] = True
    _check_over(e4, np.array([len(m) for m in lists]), mx, inside, len(pos))


@pytest.mark.gpu
def test_over_full_rows_are_uniform_subsets():
    """4 000 seeds on one row with 40 members, 8 kept: every member is kept with probability 1/5 (5-sigma binomial band), and the
    kept SET is what is uniform -- pairs of members co-occur at the hypergeometric rate"""
    from morig_amd import graph_build as gb
    n, keep, trials = 41, 8, 4000
    pos = torch.zeros((n, 3), device=DEV)
    pos[1:, 0] = torch.linspace(0.001, 0.04, n - 1, device=DEV)         # vertex 0 sees all 40 others
    hits = np.zeros(n)
    pair = 0
    for s in range(trials):
        e = _rows(gb.get_geo_edges(pos, None, 0.05, keep, seed=1000 + s))
        row0 = e[e[:, 0] == 0, 1]
        assert len(row0) == keep
        hits[row0] += 1
        pair += int(1 in row0 and 2 in row0)
    p = keep / (n - 1)
    sd = np.sqrt(trials * p * (1 - p))
    assert (np.abs(hits[1:] - trials * p) < 5 * sd).all(), hits
    pp = p * (keep - 1) / (n - 2)
    assert abs(pair - trials * pp) < 5 * np.sqrt(trials * pp * (1 - pp))


@pytest.mark.gpu
def test_headline_batch_graph_feeds_the_forward():
    """the 64 x 4096-vertex batch of bench.py with its geo graph built on the device: degree statistics of the synthetic recipe
    (ball r = 0.06, <= 15 members), symmetric membership where no cap applied, and the jointnet forward accepts it"""
    from morig_amd import graph_build as gb, models, synth
    b = synth.make_batch(range(3), n_side=64)
    d = b.to(DEV)
    ei, members = gb.get_geo_edges(d.pos, d.batch, 0.06, 15, seed=4, self_loops=True, return_members=True, num_graphs=3)
    n = d.pos.shape[0]
    e = ei.cpu().numpy()
    deg = np.bincount(e[0], minlength=n)
    assert deg.max() == 16 and deg.min() >= 1                                 # <= 15 members + the self loop
    mem = members.cpu().numpy()
    host = np.bincount(b.geo_edge_index[0].numpy(), minlength=n)               # the host recipe: same balls, another random subset
    # (the host recipe squares torch.cdist's rooted distance: a member exactly on the sphere may differ by one rounding)
    assert (np.minimum(mem, 15) + 1 != host).mean() < 1e-3
    m = synth.load_recipe(models.jointnet_motion(num_keyframes=5, chn_output=3, aggr_method="attn").eval(), 0, mild=True).to(DEV)
    d.geo_edge_index = ei
    out = m(d, d.pred_flow)[2]
    assert bool(torch.isfinite(out).all())
