"""CPU: host logic of morig_amd/joints.py (thresholds, mirror, k from the quantile, numpy visiting order, flip) on the
emulated op layer against the reference-generated fixtures; GPU: the same through the HIP kernels (float64)."""
import types

import numpy as np
import pytest
import torch

import morig_amd.runtime as runtime
from conftest import load_golden
from emulate import EmuOps
from morig_amd import joints as J


def _load(name):
    meta, a = load_golden(name)
    a = {k: v.numpy() for k, v in a.items()}
    vox = types.SimpleNamespace(data=np.unpackbits(a["vox_data"])[:88 ** 3].reshape(88, 88, 88).astype(bool),
                                translate=meta["vox_translate"], scale=meta["vox_scale"], dims=meta["vox_dims"])
    return meta, a, vox


def _check_against_fixture(name, device, tol):
    meta, a, vox = _load(name)
    out = J.extract_joints(a["shifted"], a["attn_raw"], vox, meta["quantile"], meta["threshold1"], meta["threshold2"],
                           meta["max_iter"], device=device)
    assert out["bandwidth"] == pytest.approx(float(a["bandwidth"][0]), rel=1e-12)
    assert np.array_equal(out["attn"].cpu().numpy(), a["attn_mirrored"])
    assert np.abs(out["modes"].cpu().numpy() - a["modes"]).max() <= tol
    assert out["joints"].shape == a["joints"].shape and np.abs(out["joints"] - a["joints"]).max() <= tol
    assert np.array_equal(out["side"], a["side"])
    # the pieces, through the reference-named functions
    dev = torch.device(device)
    pm = torch.from_numpy(a["pts_mirrored"]).to(dev)
    am = torch.from_numpy(a["attn_mirrored"]).to(dev)
    bw = float(a["bandwidth"][0])
    assert np.abs(J.meanshift_cluster(pm, bw, am, 3).cpu().numpy() - a["modes_two_steps"]).max() <= tol
    assert np.abs(J.meanshift_cluster(pm, bw, None, 5).cpu().numpy() - a["modes_unweighted"]).max() <= tol
    kept = J.nms_meanshift(torch.from_numpy(a["modes"]).to(dev), am, bw, meta["threshold2"])
    assert np.array_equal(kept.cpu().numpy(), a["joints_nms"])
    pts_in, idx = J.inside_check(torch.from_numpy(a["shifted"]).to(dev), vox)
    assert np.array_equal(idx.cpu().numpy(), a["index_inside"])


@pytest.fixture
def emulated_ops():
    runtime._test_ops = EmuOps()
    yield
    runtime._test_ops = None


@pytest.mark.parametrize("name", ["joints_small", "joints_medium"])
def test_extract_joints_host_logic(emulated_ops, name):
    _check_against_fixture(name, "cpu", 1e-11)


def _check_batched(device, tol):
    """both fixtures (different sizes, their own voxel grids) + an all-rejected mesh as ONE batch through extract_joints_batched:
    per mesh the reference-generated fixture, as the one-mesh path"""
    loaded = [_load(n) for n in ("joints_small", "joints_medium")]
    meta = loaded[0][0]
    pts = [a["shifted"] for _, a, _ in loaded]
    att = [a["attn_raw"] for _, a, _ in loaded]
    # third mesh: nothing passes the attention threshold except one point (min-max normalisation needs a range)
    pts.append(np.random.default_rng(0).uniform(-0.2, 0.2, (50, 3)))
    a3 = np.zeros((50, 1), dtype=np.float32); a3[7] = 1.0
    att.append(a3)
    voxs = [v for _, _, v in loaded] + [None]
    batch = torch.cat([torch.full((len(p),), b, dtype=torch.long) for b, p in enumerate(pts)])
    outs = J.extract_joints_batched(torch.from_numpy(np.concatenate(pts)).to(device), torch.from_numpy(np.concatenate(att)).to(device),
                                    batch.to(device), voxs, meta["quantile"], meta["threshold1"], meta["threshold2"], meta["max_iter"],
                                    num_graphs=3)
    assert len(outs) == 3
    for out, (_, a, _) in zip(outs, loaded):
        assert out["bandwidth"] == pytest.approx(float(a["bandwidth"][0]), rel=1e-12)
        assert np.array_equal(out["attn"].cpu().numpy(), a["attn_mirrored"])
        assert np.abs(out["modes"].cpu().numpy() - a["modes"]).max() <= tol
        assert out["joints"].shape == a["joints"].shape and np.abs(out["joints"] - a["joints"]).max() <= tol
        assert np.array_equal(out["side"], a["side"])
    one = J.extract_joints(pts[2], att[2], None, meta["quantile"], meta["threshold1"], meta["threshold2"], meta["max_iter"], device=device)
    assert outs[2]["modes"].shape[0] == 2 and np.abs(outs[2]["joints"] - one["joints"]).max() <= tol


def test_extract_joints_batched_host_logic(emulated_ops):
    _check_batched("cpu", 1e-11)


@pytest.mark.gpu
def test_extract_joints_batched_on_gpu():
    _check_batched("cuda:0", 1e-11)


@pytest.mark.gpu
def test_batched_bandwidth_selection_is_exact():
    """the k-th nearest-neighbour distance (radix select: four rows per workgroup, 12 key bits per pass over the range [2^-31, 2) of
    squared distances, short list; generic 8-bit passes for rows outside that range) against torch.kthvalue of the float64 distance
    matrix computed with the kernel's operation order: ragged meshes, duplicated points (mirror pairs: equal distances), k = 1, tiny
    meshes, a set 100 x larger than the unit box and one 10^-6 x smaller (both outside the first pass's range), coincident points,
    quantiles up to 1"""
    from morig_amd import native
    ops = native.get_ops()
    rng = np.random.default_rng(11)
    dev = torch.device("cuda:0")
    sets = [rng.normal(0, 0.1, (n, 3)) for n in (1000, 37, 2600, 9, 513, 300, 300, 70, 1, 2)]
    sets[2][1300:] = sets[2][:1300] * np.array([[-1, 1, 1]])              # mirror images
    sets[1][5:10] = sets[1][0]                                              # exact duplicates
    sets[5] *= 1000.0                                                       # squared distances >= 2: above the first pass's range
    sets[6] *= 1e-6                                                         # ... and below it
    sets[7][:] = sets[7][0]                                                 # all points coincide
    sets[0][:200] = np.round(sets[0][:200], 2)                              # a coarse lattice: many equal non-zero distances
    P = torch.from_numpy(np.concatenate(sets)).to(dev)
    sizes = [len(x) for x in sets]
    ptr = torch.tensor(np.concatenate([[0], np.cumsum(sizes)]), dtype=torch.int32, device=dev)
    for q in (0.04, 0.3, 0.0, 0.9, 1.0):
        bw = ops.knn_bandwidth_batched(P, ptr, max(sizes), q).cpu().numpy()
        for b, x in enumerate(sets):
            t = torch.from_numpy(x)
            d = t[:, None, :] - t[None, :, :]
            d2 = (d[..., 0] * d[..., 0] + d[..., 1] * d[..., 1]) + d[..., 2] * d[..., 2]
            k = max(int(len(x) * q), 1)
            want = float(torch.sqrt(torch.kthvalue(d2, k, dim=1).values).mean())
            assert bw[b] == pytest.approx(want, rel=1e-13), (q, b)


@pytest.mark.gpu
def test_neighbour_counts_match_numpy_rooted_compare():
    """counts within the bandwidth: the kernels compare d2 <= T (T = the largest double whose root is <= h) instead of
    sqrt(d2) <= h, and the sorted form skips far boxes -- both must give numpy's integers, also when the bandwidth IS the rooted
    distance of some pair (equality decides) and for bandwidth 0 on coincident points"""
    from morig_amd import native
    ops = native.get_ops()
    rng = np.random.default_rng(23)
    dev = torch.device("cuda:0")
    sets = [rng.normal(0, 0.2, (n, 3)) for n in (700, 33, 1500, 64, 5)]
    sets[2] = (sets[2][:20][rng.integers(0, 20, 1500)] + rng.normal(0, 1e-3, (1500, 3)))          # tight clusters, as modes are
    sets[3][:] = sets[3][0]
    P = torch.from_numpy(np.concatenate(sets)).to(dev)
    sizes = [len(x) for x in sets]
    ptr = torch.tensor(np.concatenate([[0], np.cumsum(sizes)]), dtype=torch.int32, device=dev)
    hs = []
    for x in sets:
        d = np.sqrt(((x[:, None, :] - x[None, :, :]) ** 2).sum(-1))
        hs.append(float(np.sort(d[0])[min(len(x) - 1, 7)]))                  # a distance that occurs: d == h for that pair
    hs[3] = 0.0
    bw = torch.tensor(hs, dtype=torch.float64, device=dev)
    plain = ops.nms_counts_batched(P, ptr, max(sizes), bw).cpu().numpy()
    keys = torch.empty(P.shape[0], dtype=torch.int64, device=dev)
    native.check(ops.lib.morig_morton_keys(native._p(P), native._p(ptr), len(sets), P.shape[0], native._p(keys), native._stream()), "keys")
    perm = torch.argsort(keys)
    Ps = P[perm].contiguous()
    cs = torch.empty(P.shape[0], dtype=torch.int32, device=dev)
    bbox = torch.empty(len(sets) * ((max(sizes) + 31) // 32) * 6, dtype=torch.float64, device=dev)
    native.check(ops.lib.morig_nms_counts_sorted(native._p(Ps), native._p(ptr), len(sets), P.shape[0], max(sizes), native._p(bw), native._p(bbox),
                                                 native._p(cs), native._stream()), "counts_sorted")
    srt = torch.empty_like(cs)
    srt[perm] = cs
    srt = srt.cpu().numpy()
    off = 0
    for b, x in enumerate(sets):
        dx = x[:, None, :] - x[None, :, :]
        d = np.sqrt((dx[..., 0] * dx[..., 0] + dx[..., 1] * dx[..., 1]) + dx[..., 2] * dx[..., 2])
        want = (d <= hs[b]).sum(0)
        assert (plain[off:off + len(x)] == want).all(), b
        assert (srt[off:off + len(x)] == want).all(), b
        off += len(x)


@pytest.mark.gpu
def test_sorted_kernels_on_large_and_ragged_sets():
    """point sets beyond one 256-box chunk (20 000 and 9 000 points next to a 40-point one): the box-granular mean-shift and the
    culled neighbour counts walk several chunks of source boxes per target block; beyond 65 535 points the bandwidth selection
    takes the one-row kernel (16-bit bins no longer hold the counts). Against the plain (unsorted, unculled) kernels."""
    from morig_amd import native
    ops = native.get_ops()
    rng = np.random.default_rng(41)
    dev = torch.device("cuda:0")
    sizes = [20000, 40, 9000]
    sets = []
    for n in sizes:
        c = rng.uniform(-0.4, 0.4, (12, 3))
        sets.append(c[rng.integers(0, 12, n)] + rng.normal(0, 0.04, (n, 3)))
    P = torch.from_numpy(np.concatenate(sets)).to(dev)
    A = torch.from_numpy((rng.random(sum(sizes)) ** 2).astype(np.float32)).to(dev)
    ptr = torch.tensor(np.concatenate([[0], np.cumsum(sizes)]), dtype=torch.int32, device=dev)
    bw = ops.knn_bandwidth_batched(P, ptr, max(sizes), 0.02)
    plain = ops.meanshift_batched(P, A, ptr, max(sizes), bw, 12)
    modes, counts = ops.meanshift_batched_sorted(P, A, ptr, max(sizes), bw, 12, with_counts=True)
    assert float((modes - plain).abs().max()) <= 1e-11
    want = ops.nms_counts_batched(modes, ptr, max(sizes), bw)
    assert bool((counts == want).all())
    # 70 000 points in one set: the one-row selection kernel; every 977th row against torch.kthvalue
    big = torch.from_numpy(rng.normal(0, 0.2, (70000, 3))).to(dev)
    pb = torch.tensor([0, 70000], dtype=torch.int32, device=dev)
    k = int(70000 * 0.001)
    kth = torch.empty(70000, dtype=torch.float64, device=dev)
    out = torch.empty(1, dtype=torch.float64, device=dev)
    native.check(ops.lib.morig_knn_bandwidth_batched(native._p(big), native._p(pb), 1, 70000, 70000, 0.001, native._p(kth), native._p(out),
                                                     native._stream()), "bandwidth")
    rows = torch.arange(0, 70000, 977, device=dev)
    d = big[rows][:, None, :] - big[None, :, :]
    d2 = (d[..., 0] * d[..., 0] + d[..., 1] * d[..., 1]) + d[..., 2] * d[..., 2]
    ref = torch.sqrt(torch.kthvalue(d2, k, dim=1).values)
    assert torch.equal(kth[rows], ref)
    assert float(out[0]) == pytest.approx(float(kth.mean()), rel=1e-12)


@pytest.mark.gpu
def test_batched_joint_extraction_equals_per_mesh_at_bench_size():
    """8 meshes x 4096 shifted points (+ mirror images), the workload of bench.py's secondary line: every mesh of the batched run
    gives the joints of its own one-mesh run. (The batched mean-shift adds the same non-zero terms in Morton order and skips source
    tiles whose kernel values are all 0: modes agree to rounding, 1e-12, not bit for bit.)"""
    rng = np.random.default_rng(5)
    dev = torch.device("cuda:0")
    P, A = [], []
    for b in range(8):
        centres = rng.uniform(-0.4, 0.4, (20, 3)); centres[:, 0] = -np.abs(centres[:, 0])
        n = 4096 - 64 * b
        P.append(centres[rng.integers(0, 20, n)] + rng.normal(0, 0.03, (n, 3)))
        A.append((rng.random((n, 1)) ** 2).astype(np.float32))
    batch = torch.cat([torch.full((len(p),), b, dtype=torch.long) for b, p in enumerate(P)]).to(dev)
    outs = J.extract_joints_batched(torch.from_numpy(np.concatenate(P)).to(dev), torch.from_numpy(np.concatenate(A)).to(dev), batch,
                                    None, 0.04, -1.0, 0.02, 30, num_graphs=8)
    for b in range(8):
        one = J.extract_joints(torch.from_numpy(P[b]).to(dev), torch.from_numpy(A[b]).to(dev), None, 0.04, -1.0, 0.02, 30)
        assert outs[b]["bandwidth"] == one["bandwidth"]
        assert float((outs[b]["modes"] - one["modes"]).abs().max()) <= 1e-11
        assert outs[b]["joints"].shape == one["joints"].shape and np.abs(outs[b]["joints"] - one["joints"]).max() <= 1e-11
        assert np.array_equal(outs[b]["side"], one["side"])
    # the culled, sorted mean-shift against the plain batched one on the same sets
    from morig_amd import native
    ops = native.get_ops()
    pts = torch.cat([o["modes"] for o in outs]) * 0 + torch.from_numpy(np.concatenate([np.concatenate([p, p * np.array([[-1, 1, 1]])]) for p in P])).to(dev)
    att = torch.cat([o["attn"] for o in outs]).reshape(-1).contiguous()
    sizes = [2 * len(p) for p in P]
    ptr = torch.tensor(np.concatenate([[0], np.cumsum(sizes)]), dtype=torch.int32, device=dev)
    bw = torch.tensor([o["bandwidth"] for o in outs], dtype=torch.float64, device=dev)
    plain = ops.meanshift_batched(pts, att, ptr, max(sizes), bw, 30)
    culled = ops.meanshift_batched_sorted(pts, att, ptr, max(sizes), bw, 30)
    assert float((plain - culled).abs().max()) <= 1e-11


def test_flip_known_answer():
    j, side = J.flip(np.array([[-0.3, 1, 2], [0.01, 3, 4], [0.4, 5, 6], [-0.02, 7, 8]]))
    assert j.tolist() == [[-0.3, 1, 2], [0.0, 3, 4], [0.0, 7, 8], [0.3, 1, 2]] and side.tolist() == [-1, 0, 0, 1]


def test_no_cpu_fallback():
    """without the emulation seam the product path refuses CPU tensors / a missing GPU"""
    from morig_amd import native
    with pytest.raises((native.MorigNativeError, RuntimeError, AssertionError, OSError)):
        J.estimate_bandwidth(torch.zeros(10, 3, dtype=torch.float64), 0.3)


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["joints_small", "joints_medium"])
def test_extract_joints_on_gpu(name):
    _check_against_fixture(name, "cuda:0", 1e-11)


@pytest.mark.gpu
def test_joint_kernels_larger_set_against_reference_fixture():
    """6144 mirrored points: bandwidth (k = 245), 29 mean-shift steps with device-side convergence, NMS -- against
    tests/golden/joints_larger.npz, the outputs of the reference's own utils/cluster_utils.py functions + sklearn's
    estimate_bandwidth on the same seeded points (oracle/make_golden.py joints_larger). The float64 CPU restatement that used to be
    re-run here on every GPU run (52 s of host time) is held to the same file in the CPU suite."""
    _, a = load_golden("joints_larger")
    pts, attn = a["pts"].numpy(), a["attn"].numpy()
    want_bw = float(a["bandwidth"][0])
    want_modes, want_kept = a["modes"].numpy(), a["kept"].numpy()
    dev = torch.device("cuda:0")
    p = torch.from_numpy(pts).to(dev)
    at = torch.from_numpy(attn).to(dev)
    bw = J.estimate_bandwidth(p, 0.04)
    assert float(bw.item()) == pytest.approx(want_bw, rel=1e-12)
    modes = J.meanshift_cluster(p, bw, at, max_iter=30)
    assert np.abs(modes.cpu().numpy() - want_modes).max() <= 1e-10
    kept = J.nms_meanshift(torch.from_numpy(want_modes).to(dev), at, want_bw, 0.02)
    assert np.array_equal(kept.cpu().numpy(), want_kept) and 4 <= len(want_kept) <= 200
    # early exit: a generous bandwidth converges well before max_iter; later launches must pass the points through
    quick = J.meanshift_cluster(p, 10.0, None, max_iter=200)
    assert np.abs(quick.cpu().numpy() - a["quick"].numpy()).max() <= 1e-10


def test_oracle_reproduces_the_larger_reference_fixture():
    """the CPU restatement (oracle/joints.py) against the larger reference fixture where that is cheap on the CPU: the bandwidth
    (exact k-th-neighbour selection over 6144 points) and the NMS on the reference's modes (one O(n^2) pass each). Its mean-shift
    iteration is held to the reference on the two smaller fixtures (test_oracle_joints.py)."""
    from oracle import joints as O
    _, a = load_golden("joints_larger")
    pts, attn = a["pts"].numpy(), a["attn"].numpy()
    assert O.estimate_bandwidth(pts, 0.04) == pytest.approx(float(a["bandwidth"][0]), rel=1e-12)
    kept, _, _ = O.nms_meanshift(a["modes"].numpy(), attn, float(a["bandwidth"][0]), 0.02)
    assert np.array_equal(kept, a["kept"].numpy())
