"""GPU: the drop-in modules (morig_amd.models) on the HIP path, through the C ABI, against
 (1) golden vectors produced by the reference's own models/*.py (tests/golden), and
 (2) the CPU oracle on fresh seeded inputs.
Criterion (BASELINE.json: "within 1e-4 fp32"): max |diff| <= 1e-4 * max(1, max |reference|)."""
import pytest
import torch

from conftest import load_golden
from helpers import POINT_MODULE_CASES, check_point_module, data_from, maxdiff, rel_excess
from morig_amd import models, synth
from morig_amd.models import basic_modules as bm, rignet as rn

pytestmark = pytest.mark.gpu
DEV = "cuda"
TOL = 1e-4


def test_native_library_is_loaded_and_gfx950():
    from morig_amd import native
    info = native.device_info()
    assert info["status"] == 0 and info["arch"].startswith("gfx950"), info
    assert native.get_ops().name == "hip"


def test_edgeconvmotion_layer():
    meta, a = load_golden("edgeconvmotion_c64_h128")
    m = bm.EdgeConvMotion(nn_x=bm.MLP([128, 128, 128]), nn_pos=bm.MLP([6, 16, 16])).eval()
    synth.load_recipe(m, meta["recipe_seed"]).to(DEV)
    out = m(a["pos"].to(DEV), a["x"].to(DEV), a["edge_index"].to(DEV))
    assert rel_excess(out, a["out"], TOL) <= 0


def test_edgeconvmotion_1d_feature():
    meta, a = load_golden("edgeconvmotion_x1d")
    m = bm.EdgeConvMotion(nn_x=bm.MLP([2, 32, 32]), nn_pos=bm.MLP([6, 16, 16])).eval()
    synth.load_recipe(m, meta["recipe_seed"]).to(DEV)
    assert rel_excess(m(a["pos"].to(DEV), a["x"].to(DEV), a["edge_index"].to(DEV)), a["out"], TOL) <= 0


def test_gcumotion_layer():
    meta, a = load_golden("gcumotion_256_512")
    m = synth.load_recipe(bm.GCUMotion(256, 512).eval(), meta["recipe_seed"]).to(DEV)
    out = m(a["pos"].to(DEV), a["x"].to(DEV), a["tpl_edge_index"].to(DEV), a["geo_edge_index"].to(DEV))
    assert rel_excess(out, a["out"], TOL) <= 0


def test_gcu_layer():
    meta, a = load_golden("gcu_3_32")
    m = synth.load_recipe(bm.GCU(3, 32).eval(), meta["recipe_seed"]).to(DEV)
    assert rel_excess(m(a["x"].to(DEV), a["tpl_edge_index"].to(DEV), a["geo_edge_index"].to(DEV)), a["out"], TOL) <= 0


@pytest.mark.parametrize("name", ["gcnrig_f3_o32", "gcnrig_f64_o3"])
def test_gcnrig(name):
    meta, a = load_golden(name)
    m = synth.load_recipe(rn.GCNRig(meta["chn_feature"], meta["chn_output"]).eval(), meta["recipe_seed"]).to(DEV)
    out = m(a["pos"].to(DEV), a["feature"].to(DEV), a["tpl_edge_index"].to(DEV), a["geo_edge_index"].to(DEV), a["batch"].to(DEV))
    assert rel_excess(out, a["out"], TOL) <= 0


def test_temporal_attention():
    meta, a = load_golden("temporalattn_32_64")
    m = synth.load_recipe(rn.TemporalAttn(32, 2, 64, 512, 64).eval(), meta["recipe_seed"]).to(DEV)
    assert rel_excess(m(a["x"].to(DEV)), a["out"], TOL) <= 0


@pytest.mark.parametrize("name,outs", [
    ("jointnet_ragged", ("motion_all", "motion_aggr", "pred_shift")),
    ("jointnet_mean", (None, "motion_aggr", "pred_shift")),
    ("jointnet_max", (None, "motion_aggr", "pred_shift")),
    ("masknet_ragged", ("motion_all", "motion_aggr", "pred_mask")),
    ("skinnet_ragged", ("motion_all", "motion_aggr", "skin_cls_pred")),
    ("skinnet_dg1_lf1", (None, None, "skin_cls_pred")),
    ("skinnet_dg1_lf0", (None, None, "skin_cls_pred")),
    ("skinnet_dg0_lf1", (None, None, "skin_cls_pred")),
])
def test_networks_against_reference_goldens(name, outs):
    meta, a = load_golden(name)
    m = models.__dict__[meta["arch"]](**meta["kwargs"]).eval()
    synth.load_recipe(m, meta["recipe_seed"]).to(DEV)
    d = data_from(a, DEV)
    res = m(d, d.pred_flow)
    torch.cuda.synchronize()
    for r, key in zip(res, outs):
        if key is not None:
            assert r.is_cuda and r.shape == a[key].shape
            # the SkinNet logits of the Dg = 1 switch combinations reach |37|: the relative reading passes them at 2.5e-6 of scale; the
            # ABSOLUTE reading of "within 1e-4" holds as well (9.3e-5 / 8.2e-5 measured, 24 ulp at that magnitude) and is asserted, so
            # that a regression of the margin is seen
            strict = True if (name.startswith("skinnet_dg1") and key == "skin_cls_pred") else None
            assert rel_excess(r, a[key], TOL, strict=strict) <= 0, key


def test_jointnet_headline_size_mesh_against_reference_golden():
    """one 4096-vertex mesh, the size BASELINE.json's metric is quoted on."""
    meta, a = load_golden("jointnet_4k")
    mesh = synth.collate([synth.make_mesh(meta["mesh_seed"], n_side=meta["n_side"])])
    m = models.jointnet_motion(**meta["kwargs"]).eval()
    synth.load_recipe(m, meta["recipe_seed"], mild=meta["mild"]).to(DEV)
    d = mesh.to(DEV)
    _, aggr, shift = m(d, d.pred_flow)
    assert rel_excess(aggr, a["motion_aggr"], TOL) <= 0
    assert rel_excess(shift, a["pred_shift"], TOL) <= 0


def test_batched_equals_per_mesh_and_is_deterministic():
    """size-independent properties at a larger batch: a mesh's outputs do not depend on its batch
    mates and two runs are bit-identical (integer-atomic max is order-free)."""
    meshes = [synth.make_mesh(50 + i, n_side=s) for i, s in enumerate((32, 24, 32, 16))]
    m = models.jointnet_motion(num_keyframes=5, chn_output=3, aggr_method="attn").eval()
    synth.load_recipe(m, 77, mild=True).to(DEV)
    full = synth.collate(meshes).to(DEV)
    ma, mg, ps = m(full, full.pred_flow)
    ma2, mg2, ps2 = m(full, full.pred_flow)
    assert torch.equal(ps, ps2) and torch.equal(ma, ma2) and torch.equal(mg, mg2)
    off = 0
    for mesh in meshes:
        one = synth.collate([mesh]).to(DEV)
        _, _, p1 = m(one, one.pred_flow)
        n = mesh.pos.shape[0]
        assert rel_excess(ps[off:off + n], p1, 2e-5) <= 0
        off += n


def test_against_oracle_on_fresh_inputs():
    from oracle import nets
    kw = dict(num_keyframes=5, chn_output=1, aggr_method="attn")
    ours = synth.load_recipe(models.masknet_motion(**kw).eval(), 909)
    ref = synth.load_recipe(nets.masknet_motion(**kw).eval(), 909)
    batch = synth.make_batch([71, 72, 73], n_side=20)
    want = ref(batch, batch.pred_flow)
    d = batch.to(DEV)
    got = ours.to(DEV)(d, d.pred_flow)
    for g, w in zip(got, want):
        assert rel_excess(g, w, TOL) <= 0


def test_degenerate_graphs_against_oracle():
    """graphs the generator never produces: a vertex with no incoming edge in either graph (only the self loop the network adds),
    duplicated edges, explicit self loops in the input (removed and re-added by the reference: models/basic_modules.py:188-189),
    a mesh of ONE vertex with no edges at all, a 9-vertex mesh (far below one kernel tile), all in one ragged batch; jointnet and
    masknet against the CPU oracle."""
    from oracle import nets
    tiny = synth.make_mesh(301, n_side=3, with_skin=False)
    holes = synth.make_mesh(302, n_side=7, with_skin=False)
    for name, dead in (("tpl_edge_index", (0, 11, 48)), ("geo_edge_index", (0, 5, 30))):
        ei = getattr(holes, name)
        keep = ~torch.isin(ei[1], torch.tensor(dead))                     # vertices 0 (both), 11 / 48 (tpl), 5 / 30 (geo): no incoming edge
        ei = ei[:, keep]
        loops = torch.tensor([[3, 3, 17], [3, 3, 17]])                    # explicit self loops, one of them twice
        setattr(holes, name, torch.cat([ei, ei[:, :40], loops], 1))       # + 40 duplicated edges
    one = synth.make_mesh(303, n_side=3, with_skin=False)
    one.pos, one.pred_flow = one.pos[:1].clone(), one.pred_flow[:1].clone()
    one.tpl_edge_index = torch.zeros((2, 0), dtype=torch.long)
    one.geo_edge_index = torch.zeros((2, 0), dtype=torch.long)
    normal = synth.make_mesh(304, n_side=10, with_skin=False)
    batch = synth.collate([tiny, holes, one, normal])
    d = batch.to(DEV)
    for arch, kw in (("jointnet_motion", dict(num_keyframes=5, chn_output=3, aggr_method="attn")),
                     ("masknet_motion", dict(num_keyframes=5, chn_output=1, aggr_method="max"))):
        ours = synth.load_recipe(models.__dict__[arch](**kw).eval(), 404)
        ref = synth.load_recipe(nets.__dict__[arch](**kw).eval(), 404)
        want = ref(batch, batch.pred_flow)
        got = ours.to(DEV)(d, d.pred_flow)
        for g, w in zip(got, want):
            assert g.shape == w.shape and rel_excess(g, w, TOL) <= 0, arch


def test_corrnet_tiny_and_lopsided_clouds_against_oracle():
    """clouds far below every tile and sampling size: 37, 700 and 9 points (the last one keeps 5 / 2 / 1 points through the three
    sampling levels) against meshes of 9, 81 and 25 vertices; deterministic FPS start, CPU oracle."""
    from oracle import nets
    ms = [synth.make_mesh(311, n_side=3, with_skin=False), synth.make_mesh(312, n_side=9, with_skin=False),
          synth.make_mesh(313, n_side=5, with_skin=False)]
    clouds = [synth.make_point_cloud(ms[0], 1, 37), synth.make_point_cloud(ms[1], 2, 700), synth.make_point_cloud(ms[2], 3, 9)]
    batch = synth.collate(ms, clouds)
    kw = dict(input_feature=3, output_feature=64, temprature=0.07)
    ref = synth.load_recipe(nets.corrnet(**kw).eval(), 5)
    ours = synth.load_recipe(models.corrnet(**kw).eval(), 5).to(DEV)
    want = ref(batch, True, False)
    got = ours(batch.to(DEV), True, False)
    for g, w in zip(got[:3], want[:3]):
        assert g.shape == w.shape and rel_excess(g, w, TOL) <= 0


def test_corrnet_against_reference_golden():
    """vertex branch + PointNet++ point branch + cosine matching + vis-mask, deterministic FPS start."""
    meta, a = load_golden("corrnet_ragged")
    m = models.corrnet(**meta["kwargs"]).eval()
    synth.load_recipe(m, meta["recipe_seed"]).to(DEV)
    d = data_from(a, DEV)
    ov, op, vis, tau = m(d, True, False)
    torch.cuda.synchronize()
    assert rel_excess(ov, a["out_vtx"], TOL) <= 0
    assert rel_excess(op, a["out_pts"], TOL) <= 0
    assert rel_excess(vis, a["out_vismask"], TOL) <= 0
    assert float(tau) == pytest.approx(0.07)
    assert m(d, False, True)[2] is None


@pytest.mark.parametrize("name", ["deformnet_ragged", "deformnet_three"])
def test_deformnet_against_reference_golden(name):
    """CorrNet -> mask normalisation -> k-NN votes -> GCNDeform, against the reference's own outputs;
    FPS starts from the recorded torch seed (the reference's default random_start=True, deformnet.py:41)."""
    meta, a = load_golden(name)
    m = models.deformnet(**meta["kwargs"]).eval()
    synth.load_recipe(m, meta["recipe_seed"], mild=meta["mild"]).to(DEV)
    d = data_from(a, DEV)
    torch.manual_seed(meta["rng_seed"])
    pf, vf, ptf, vis, tau = m(d)
    torch.cuda.synchronize()
    assert rel_excess(vf, a["vtx_feature"], TOL) <= 0 and rel_excess(ptf, a["pts_feature"], TOL) <= 0
    assert rel_excess(vis, a["pred_vismask"], TOL) <= 0
    assert torch.equal(vis.cpu() >= 0.5, a["pred_vismask"] >= 0.5)
    assert rel_excess(pf, a["out_pred_flow"], TOL) <= 0
    assert float(tau) == pytest.approx(0.07)


def test_deformnet_larger_clouds_against_oracle():
    """At larger sizes the k-NN choice is ill-conditioned (symmetric meshes give vertex features whose similarities tie
    to ~1e-7, and one swapped neighbour moves pred_flow through the global max-pool), so: (1) features and mask against
    the oracle, (2) the product's neighbour tables have the oracle's similarity profile, (3) everything downstream of
    the neighbour choice against the oracle evaluated on the SAME tables."""
    from oracle import nets, pyg_primitives as P
    kw = dict(tau_nce=0.07, num_interp=5)
    ours = synth.load_recipe(models.deformnet(**kw).eval(), 61, mild=True)
    ref = synth.load_recipe(nets.deformnet(**kw).eval(), 61, mild=True)
    batch = synth.make_batch([91, 92], n_side=24, n_pts=2048)
    torch.manual_seed(5)
    want = ref(batch)
    torch.manual_seed(5)
    got = ours.to(DEV)(batch.to(DEV))
    for g, w in zip(got[1:4], want[1:4]):
        assert rel_excess(g, w, TOL) <= 0
    assert float((want[3] - 0.5).abs().min()) > 2e-4                 # the split itself is not decided by a rounding
    assert torch.equal(got[3].cpu() >= 0.5, want[3] >= 0.5)
    to_pts, to_vis = [t.cpu() for t in ours.last_neighbours]
    vf, pf, vis = want[1], want[2], want[3]
    k = 5
    # (2) similarity profiles
    yi, xi = P.knn(pf, vf, k, batch.pts_batch, batch.vtx_batch, cosine=True)
    want_sim = (pf[xi] * vf[yi]).sum(-1).view(-1, k)
    got_sim = (pf[to_pts.long()] * vf[:, None, :]).sum(-1)
    assert bool((to_pts >= 0).all()) and (got_sim - want_sim).abs().max().item() <= 2e-6
    seen = (vis >= 0.5).squeeze(1); hid = (vis < 0.5).squeeze(1)
    assert bool((to_vis[seen] == -1).all()) and bool((to_vis[hid] >= 0).all())
    yi2, xi2 = P.knn(vf[seen], vf[hid], k, batch.vtx_batch[seen], batch.vtx_batch[hid], cosine=True)
    want_sim2 = (vf[seen][xi2] * vf[hid][yi2]).sum(-1).view(-1, k)
    got_sim2 = (vf[to_vis[hid].long()] * vf[hid][:, None, :]).sum(-1)
    assert (got_sim2 - want_sim2).abs().max().item() <= 2e-6
    assert bool(seen[to_vis[hid].long()].all())
    assert bool((batch.vtx_batch[to_vis[hid].long()] == batch.vtx_batch[hid][:, None]).all())
    # (3) downstream of the neighbour choice
    torch.manual_seed(5)
    want_same = ref(batch, neighbours=(to_pts, to_vis))
    # pred_flow = votes (softmax-weighted differences of ~4-unit positions) + GCNDeform: relative to the output scale
    # (beyond the 1e-4-absolute networks of rows a8-a12); measured 1.8e-4 absolute at scale 4.0
    assert rel_excess(got[0], want_same[0], TOL, strict=False) <= 0


def test_corrnet_larger_clouds_against_oracle():
    from oracle import nets
    kw = dict(input_feature=3, output_feature=64, temprature=0.07)
    ours = synth.load_recipe(models.corrnet(**kw).eval(), 31, mild=True)
    ref = synth.load_recipe(nets.corrnet(**kw).eval(), 31, mild=True)
    batch = synth.make_batch([81, 82], n_side=24, n_pts=2048)
    want = ref(batch, True, False)
    got = ours.to(DEV)(batch.to(DEV), True, False)
    for g, w in zip(got[:3], want[:3]):
        assert rel_excess(g, w, TOL) <= 0


def test_corrnet_three_streams_are_deterministic_and_equal_one_stream(monkeypatch):
    """CorrNet runs its vertex branch, the point-feature chain and the point geometry (FPS levels, k-NN searches) on three HIP
    streams tied together by events (morig_amd/models/corrnet.py): eight forwards of a full-size batch must be bit-identical
    to each other and to the single-stream schedule -- a missing dependency or a buffer recycled too early shows up here."""
    kw = dict(input_feature=3, output_feature=64, temprature=0.07)
    m = synth.load_recipe(models.corrnet(**kw).eval(), 3, mild=True).to(DEV)
    batch = synth.make_batch(range(40, 48), n_side=64, n_pts=8192).to(DEV)
    monkeypatch.setenv("MORIG_TWO_STREAMS", "0")
    ref = [t.clone() for t in m(batch, True, False)[:3]]
    monkeypatch.setenv("MORIG_TWO_STREAMS", "1")
    for _ in range(8):
        got = m(batch, True, False)
        torch.cuda.synchronize()
        for g, r in zip(got[:3], ref):
            assert torch.equal(g, r)


def test_deformnet_full_size_properties():
    """BASELINE.json configs[3] pair size (4096-vertex mesh + 8192-point cloud): size-independent properties of the DeformNet
    stages -- mask spans [0, 1] per mesh, neighbour tables stay inside the pair, invisible vertices only consult visible
    ones, the voted flow of a visible vertex lies in the convex hull of its candidate displacements."""
    kw = dict(tau_nce=0.07, num_interp=5)
    m = synth.load_recipe(models.deformnet(**kw).eval(), 7, mild=True).to(DEV)
    batch = synth.make_batch([301, 302], n_side=64, n_pts=8192).to(DEV)
    torch.manual_seed(1)
    pred, vf, pf, vis, tau = m(batch)
    torch.cuda.synchronize()
    V, P = 4096, 8192
    assert pred.shape == (2 * V, 3) and bool(torch.isfinite(pred).all())
    assert float((vf.norm(dim=1) - 1).abs().max()) < 1e-5 and float((pf.norm(dim=1) - 1).abs().max()) < 1e-5
    for b in range(2):
        v = vis[b * V:(b + 1) * V]
        assert float(v.min()) == 0.0 and float(v.max()) == 1.0
    to_pts, to_vis = [t.long() for t in m.last_neighbours]
    mesh = torch.arange(2 * V, device=DEV) // V
    assert bool((to_pts >= 0).all()) and bool(((to_pts // P) == mesh[:, None]).all())
    hidden = (vis < 0.5).squeeze(1)
    assert bool((to_vis[~hidden] == -1).all()) and bool((to_vis[hidden] >= 0).all())
    assert bool((vis[to_vis[hidden]].squeeze(-1) >= 0.5).all()) and bool(((to_vis[hidden] // V) == mesh[hidden][:, None]).all())
    # visible vertices with positive weights: flow_init inside the per-axis range of (pts[j] - vtx[i])
    sim = (pf[to_pts] * vf[:, None, :]).sum(-1)
    ok = (~hidden) & (sim > 0).all(1) & (vis.squeeze(1) > 0)
    disp = batch.pts[to_pts] - batch.vtx[:, None, :]
    # re-derive flow_init from the tables (fp32 on the device) and compare with what GCNDeform was fed
    w = sim * vis
    flow = (disp * w[..., None]).sum(1) / w.sum(1, keepdim=True)
    lo, hi = disp.min(1).values, disp.max(1).values
    assert bool(((flow >= lo - 1e-5) & (flow <= hi + 1e-5))[ok].all())


@pytest.mark.parametrize("kind,ratio,r", POINT_MODULE_CASES)
def test_point_modules_standalone_forward(kind, ratio, r):
    """SAModule / GlobalSAModule / FPModule called on their own (models/basic_modules.py:74-86,121-125,133-138) on the HIP
    path vs the oracle, at CorrNet's three (ratio, r) settings (models/corrnet.py:24-26)."""
    check_point_module(kind, ratio, r, DEV)


def test_out_of_range_edge_index_raises():
    """the CSR kernels drop an out-of-range index and raise a status word; the forward reads it with the precision flag
    and raises, as the reference's gather would."""
    from morig_amd import native
    mesh = synth.collate([synth.make_mesh(5, n_side=12)])
    m = synth.load_recipe(models.jointnet_motion(num_keyframes=5, chn_output=3, aggr_method="attn").eval(), 1, mild=True).to(DEV)
    d = mesh.to(DEV)
    m(d, d.pred_flow)                                   # sane input: no error
    d.tpl_edge_index = d.tpl_edge_index.clone()
    d.tpl_edge_index[0, 7] = d.pos.shape[0] + 3
    with pytest.raises(native.MorigNativeError):
        m(d, d.pred_flow)


def test_model_on_second_device_context_uses_its_own_stream():
    """launches follow the MODEL's device, not the caller's current device. With one GPU visible the guard is
    exercised by running the forward from a thread whose current stream is a side stream and checking the result."""
    mesh = synth.collate([synth.make_mesh(6, n_side=12)])
    m = synth.load_recipe(models.jointnet_motion(num_keyframes=5, chn_output=3, aggr_method="attn").eval(), 2, mild=True).to(DEV)
    d = mesh.to(DEV)
    want = m(d, d.pred_flow)[2]
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        got = m(d, d.pred_flow)[2]
    side.synchronize()
    assert torch.equal(got, want)


_BATCH64 = []


def _batch64_with_both_golden_meshes():
    """ONE host-built batch of 64 meshes x 4096 vertices (bench.py's host recipe; the geodesic-ball graphs of 64 meshes are ~10-25 s
    of host time) for both batch-64 tests: the jointnet golden mesh at slot 37, the masknet / skinnet golden mesh at slot 21,
    skin inputs attached (they do not change positions, graphs or flows: the mesh recipe draws them last)"""
    if not _BATCH64:
        import bench
        jm, _ = load_golden("jointnet_4k")
        mm, _ = load_golden("masknet_4k_harsh")
        assert jm["n_side"] == mm["n_side"]
        seeds = [2000 + i for i in range(64)]
        seeds[37], seeds[21] = jm["mesh_seed"], mm["mesh_seed"]
        _BATCH64.append(bench.build_batch(seeds, jm["n_side"], with_skin=True))
    return _BATCH64[0]


def test_headline_batch_64_meshes_contains_the_golden_mesh_and_is_deterministic():
    """BASELINE.json configs[1] as bench.py runs it -- 64 meshes x 4096 vertices in ONE batch -- with the committed
    4096-vertex golden mesh at position 37: its rows must equal the reference's single-mesh outputs (a mesh's outputs do not
    depend on its batch mates), and two runs of the whole batch must be bit-identical (VERDICT r1 #4a)."""
    meta, a = load_golden("jointnet_4k")
    batch = _batch64_with_both_golden_meshes()
    m = models.jointnet_motion(**meta["kwargs"]).eval()
    synth.load_recipe(m, meta["recipe_seed"], mild=meta["mild"]).to(DEV)
    d = batch.to(DEV)
    n = meta["n_side"] ** 2
    assert torch.equal(d.pos[37 * n:38 * n].cpu(), a["pos_check"]) if a["pos_check"].shape[0] == n else True
    ma, aggr, shift = m(d, d.pred_flow)
    ma2, aggr2, shift2 = m(d, d.pred_flow)
    assert torch.equal(shift, shift2) and torch.equal(aggr, aggr2) and torch.equal(ma, ma2)
    assert bool(torch.isfinite(shift).all())
    sl = slice(37 * n, 38 * n)
    assert rel_excess(aggr[sl], a["motion_aggr"], TOL) <= 0
    assert rel_excess(shift[sl], a["pred_shift"], TOL) <= 0


def test_ragged_batch_64_meshes_contains_the_golden_mesh():
    """The config-5 stand-in as bench.py times it (`--workload jointnet_ragged`): 64 meshes of 1 024 ... 6 400 vertices in
    ONE batch (rig datasets hold characters of different sizes: datasets/dataset_rig.py:85-138), the reference's fixed ball radius,
    with the committed 4096-vertex golden mesh at a MIDDLE slot: its rows equal the reference's single-mesh outputs -- the tile runs /
    XCD partition of the EdgeConv kernels and the per-mesh pooling on meshes of unequal length -- and two runs are bit-identical."""
    import bench
    meta, a = load_golden("jointnet_4k")
    seeds = [1000 + i for i in range(64)]
    sides = bench.ragged_sides(seeds)
    slot = 29
    seeds[slot], sides[slot] = meta["mesh_seed"], meta["n_side"]
    assert len(set(sides)) > 20 and min(sides) >= 32 and max(sides) <= 80
    batch = bench.build_batch_ragged(seeds, sides)                  # host recipe (the golden mesh's own geo graph: seeded host draw)
    m = models.jointnet_motion(**meta["kwargs"]).eval()
    synth.load_recipe(m, meta["recipe_seed"], mild=meta["mild"]).to(DEV)
    d = batch.to(DEV)
    off = sum(s_ * s_ for s_ in sides[:slot])
    n = meta["n_side"] ** 2
    sl = slice(off, off + n)
    if a["pos_check"].shape[0] == n:
        assert torch.equal(d.pos[sl].cpu(), a["pos_check"])
    ma, aggr, shift = m(d, d.pred_flow)
    ma2, aggr2, shift2 = m(d, d.pred_flow)
    assert torch.equal(shift, shift2) and torch.equal(aggr, aggr2) and torch.equal(ma, ma2)
    assert shift.shape[0] == sum(s_ * s_ for s_ in sides) and bool(torch.isfinite(shift).all())
    assert rel_excess(aggr[sl], a["motion_aggr"], TOL) <= 0
    assert rel_excess(shift[sl], a["pred_shift"], TOL) <= 0
    # ... and two more meshes of the batch -- the smallest one and one of about 2.5 k vertices -- against the CPU oracle run on each of
    # them ALONE (a mesh's outputs do not depend on its batch mates): ragged batches are held to the reference at more sizes than the
    # <= 1 k-vertex `*_ragged` goldens
    from oracle import nets
    ref = synth.load_recipe(nets.jointnet_motion(**meta["kwargs"]).eval(), meta["recipe_seed"], mild=meta["mild"])
    order = sorted(range(64), key=lambda i: sides[i])
    picks = [order[0], min(range(64), key=lambda i: abs(sides[i] - 50) + (1000 if i == slot else 0))]
    for i in picks:
        one = synth.collate([synth.make_mesh(seeds[i], n_side=sides[i], geo_radius=bench.GEO_RADIUS, with_skin=False)])
        with torch.no_grad():
            _, waggr, wshift = ref(one, one.pred_flow)
        o_i = sum(s_ * s_ for s_ in sides[:i])
        sl_i = slice(o_i, o_i + sides[i] ** 2)
        assert torch.equal(d.pos[sl_i].cpu(), one.pos)
        assert rel_excess(aggr[sl_i], waggr, TOL) <= 0, (i, sides[i])
        assert rel_excess(shift[sl_i], wshift, TOL) <= 0, (i, sides[i])


def test_mask_skin_batch_64_contains_the_golden_mesh():
    """BASELINE.json configs[2] at its stated size (VERDICT r3 #4a): masknet_motion + skinnet_motion over ONE batch of 64 meshes x
    4096 vertices, as bench.py's `mask_skin` workload runs them, with the committed harsh-recipe 4096-vertex golden mesh at
    position 21: its rows must equal the outputs of the reference's own models on that mesh alone, and two runs of the whole
    batch must be bit-identical."""
    mm, ma = load_golden("masknet_4k_harsh")
    sm, sa = load_golden("skinnet_4k_harsh")
    assert mm["mesh_seed"] == sm["mesh_seed"] and mm["n_side"] == sm["n_side"]
    slot, n = 21, mm["n_side"] ** 2
    d = _batch64_with_both_golden_meshes().to(DEV)
    sl = slice(slot * n, (slot + 1) * n)
    assert maxdiff(d.pos[sl][:8], ma["pos_check"]) == 0
    step = mm["row_step"]
    for meta, a, key in ((mm, ma, "pred_mask"), (sm, sa, "skin_cls_pred")):
        m = models.__dict__[meta["arch"]](**meta["kwargs"]).eval()
        synth.load_recipe(m, meta["recipe_seed"], mild=meta["mild"]).to(DEV)
        _, aggr, last = m(d, d.pred_flow)
        _, aggr2, last2 = m(d, d.pred_flow)
        assert torch.equal(last, last2) and torch.equal(aggr, aggr2), key
        assert last.shape[0] == 64 * n and bool(torch.isfinite(last).all())
        assert rel_excess(aggr[sl][::step], a["motion_aggr_rows"], TOL) <= 0, key
        assert rel_excess(last[sl], a[key], TOL) <= 0, key
        del m, aggr, last, aggr2, last2
        torch.cuda.empty_cache()


def test_corrnet_batch_32_pairs_contains_the_golden_pair():
    """BASELINE.json configs[3] at the per-GPU size bench.py times: corrnet over ONE batch of 32 (4096-vertex mesh,
    8192-point cloud) pairs, harsh recipe, two runs bit-identical, compared at full size in two ways.
    (a) The committed golden pair (outputs of the reference's own models/corrnet.py) sits at position 0. The VERTEX branch of a pair
    never depends on its batch mates; the POINT branch of the FIRST cloud does not either. Later clouds do, in the reference itself:
    PyG's PointConv adds self loops (i, i) for i < min(N_src, N_dst) on the BATCH-GLOBAL bipartite index ([PyG-recall]),
    so centre i of cloud b > 0 receives a message from point i of an EARLIER cloud (measured here: cloud 1's features move by 6e-2
    between a batch of one and a batch of two). The build reproduces that (morig_csr_build_bipartite), hence:
    (b) the second pair is compared with the CPU oracle's run of the first TWO pairs as one batch (a cloud's self-loop partners
    come from earlier clouds only, so pair 1 of the 32-batch equals pair 1 of the 2-batch)."""
    import bench
    from oracle import nets
    meta, a = load_golden("corrnet_4k_8k_harsh")
    nb = 32
    n, npts, step = meta["n_side"] ** 2, meta["n_pts"], meta["row_step"]
    seeds = [meta["mesh_seed"]] + [3000 + i for i in range(1, nb)]
    host = bench.build_batch(seeds, meta["n_side"], n_pts=npts)
    d = host.to(DEV)
    assert maxdiff(d.pos[:8], a["pos_check"]) == 0 and maxdiff(d.pts[:8], a["pts_check"]) == 0
    m = models.corrnet(**meta["kwargs"]).eval()
    synth.load_recipe(m, meta["recipe_seed"], mild=meta["mild"]).to(DEV)
    ov, op, vis, _ = m(d, True, False)
    ov2, op2, vis2, _ = m(d, True, False)
    assert torch.equal(ov, ov2) and torch.equal(op, op2) and torch.equal(vis, vis2)
    assert ov.shape[0] == nb * n and op.shape[0] == nb * npts and bool(torch.isfinite(vis).all())

    def vis_check(got_v, got_p, got_vis, ref_vis):
        # rows whose two best cosine similarities tie to within the feature tolerance are decided by the last bit of the features
        # (helpers.check_full_size): identified from OUR features, they must stay a sliver of the mesh
        off = (got_vis - ref_vis).abs().flatten() > TOL
        if bool(off.any()):
            top2 = (got_v.double() @ got_p.double().t()).topk(2, dim=1).values
            assert bool(((top2[:, 0] - top2[:, 1]) <= 4 * TOL)[off].all()), "a visibility row differs although its nearest point is well separated"
            assert float(off.float().mean()) <= 0.001, float(off.float().mean())
        assert rel_excess(got_vis[~off], ref_vis[~off], TOL) <= 0

    # (a) pair 0 against the reference's own outputs
    assert rel_excess(ov[:n][::step], a["out_vtx_rows"], TOL) <= 0
    assert rel_excess(op[:npts][::step], a["out_pts_rows"], TOL) <= 0
    vis_check(ov[:n], op[:npts], vis[:n], a["out_vismask"].to(DEV))
    # (b) pair 1 against the CPU oracle's batch of the first two pairs
    two = synth.make_batch(seeds[:2], n_side=meta["n_side"], n_pts=npts)
    assert torch.equal(two.pts, host.pts[:2 * npts]) and torch.equal(two.geo_edge_index, host.geo_edge_index[:, :two.geo_edge_index.shape[1]])
    ref = synth.load_recipe(nets.corrnet(**meta["kwargs"]).eval(), meta["recipe_seed"], mild=meta["mild"])
    with torch.no_grad():
        wv, wp, wvis, _ = ref(two, True, False)
    sv, sp = slice(n, 2 * n), slice(npts, 2 * npts)
    assert float((wp[sp] - wp[:npts]).abs().max()) > 1e-3          # (the two clouds differ: the slices are not mixed up)
    assert rel_excess(ov[sv], wv[sv], TOL) <= 0
    assert rel_excess(op[sp], wp[sp], TOL) <= 0
    vis_check(ov[sv], op[sp], vis[sv], wvis[sv].to(DEV))


def test_batch_past_the_32_bit_row_offsets_still_equals_the_golden():
    """260 meshes x 4096 vertices: the [A | B] operand of the 256-wide EdgeConv layers is 4.4 GB, past what the persistent
    kernels' 32-bit gather offsets reach -- the launcher must take the 64-bit-address kernel there (tile_gemm.hip `pp_ok`),
    not wrap around. Every mesh is the committed 4096-vertex golden mesh: first, last and one in the middle are compared."""
    meta, a = load_golden("jointnet_4k")
    mesh = synth.make_mesh(meta["mesh_seed"], n_side=meta["n_side"], with_skin=False)
    nb = 260
    d = synth.collate([mesh] * nb).to(DEV)
    m = models.jointnet_motion(**meta["kwargs"]).eval()
    synth.load_recipe(m, meta["recipe_seed"], mild=meta["mild"]).to(DEV)
    n = meta["n_side"] ** 2
    _, aggr, shift = m(d, d.pred_flow)
    assert shift.shape[0] == nb * n and bool(torch.isfinite(shift).all())
    for k in (0, 131, nb - 1):
        sl = slice(k * n, (k + 1) * n)
        assert rel_excess(aggr[sl], a["motion_aggr"], TOL) <= 0, k
        assert rel_excess(shift[sl], a["pred_shift"], TOL) <= 0, k
    del d, aggr, shift
    torch.cuda.empty_cache()


@pytest.mark.parametrize("name", __import__("helpers").FULL_SIZE_GOLDENS)
def test_full_size_harsh_recipe_goldens(name):
    """4096-vertex mesh (and the 8192-point cloud of configs[3]) with the harsh BatchNorm recipe on the HIP path against
    outputs of the reference's own models/*.py (VERDICT r1 #4b)."""
    from helpers import check_full_size, full_size_inputs
    meta, a = load_golden(name)
    m = models.__dict__[meta["arch"]](**meta["kwargs"]).eval()
    synth.load_recipe(m, meta["recipe_seed"], mild=meta["mild"]).to(DEV)
    check_full_size(m, meta, a, full_size_inputs(meta, DEV), TOL)


@pytest.mark.parametrize("name", ["jointnet_train", "masknet_train", "skinnet_train"])
def test_train_mode_forward(name):
    """SURVEY 8 f-4, forward half, on the HIP path: GEMMs and per-edge hidden layers on the MFMA kernels, batch statistics /
    affine / gather / segmented max in csrc/train_ops.hip; against the reference's own model.train() run (fp32 and fp64)."""
    from helpers import check_train_mode
    check_train_mode(name, DEV)


# ---- serving loop: deferred guard read and hipGraph capture (morig_amd/serving.py) ---------------------------
def _jointnet(seed=3):
    return synth.load_recipe(models.jointnet_motion(num_keyframes=5, chn_output=3, aggr_method="attn").eval(), seed, mild=True).to(DEV)


def test_forward_async_defers_the_guard_read():
    """forward_async = forward without the host read at its end: same bits, and the pending guard still reports what the
    synchronous one acts on -- ok / operand outside the split-fp16 range / edge index out of range -- even when it is read only
    after the NEXT forward was enqueued (the snapshot is per forward)."""
    from morig_amd import native
    if native.get_ops().precision == "f32":
        pytest.skip("MORIG_PRECISION=f32: the exact path has no range guard to defer (forward_async returns no pending read)")
    m = _jointnet()
    d = synth.collate([synth.make_mesh(5, n_side=16), synth.make_mesh(6, n_side=12)]).to(DEV)
    want = m(d, d.pred_flow)
    got, pend = m.forward_async(d, d.pred_flow)
    big = d.pred_flow * 1.0e7                                             # far outside the fp16 range: the fast path must flag it
    got_big, pend_big = m.forward_async(d, big)
    got2, pend2 = m.forward_async(d, d.pred_flow)                         # enqueued before any guard was read
    assert pend.result() is True and pend_big.result() is False and pend2.result() is True
    for a, b, c in zip(want, got, got2):
        assert torch.equal(a, b) and torch.equal(a, c)
    ref_big = m(d, big)                                                   # the synchronous forward re-runs on the fp32 kernels
    assert all(bool(torch.isfinite(t).all()) for t in ref_big)
    bad = synth.collate([synth.make_mesh(5, n_side=16)]).to(DEV)
    bad.geo_edge_index = bad.geo_edge_index.clone()
    bad.geo_edge_index[1, 3] = -7
    _, pend_bad = m.forward_async(bad, bad.pred_flow)
    with pytest.raises(native.MorigNativeError):
        pend_bad.result()


@pytest.mark.parametrize("sides", [(16, 12), (64, 64)])
def test_captured_forward_replays_bit_identically(sides):
    """one eval forward (CSR builds, every kernel, the guard snapshot) captured into a HIP graph: every replay equals the eager
    forward bit for bit -- also after the allocator handed out and took back other memory in between (what killed replays
    before the CSR build's memset nodes became a kernel), and after the inputs were refreshed in place."""
    from morig_amd.serving import CapturedForward
    m = _jointnet(4)
    d = synth.collate([synth.make_mesh(20 + i, n_side=s) for i, s in enumerate(sides)]).to(DEV)
    want = [t.clone() for t in m(d, d.pred_flow)]
    cf = CapturedForward(m, d, d.pred_flow)
    for r in range(4):
        out = cf.replay()
        assert cf.check() is True
        for a, b in zip(want, out):
            assert torch.equal(a, b), f"replay {r}"
        junk = [torch.full((1 << 22,), float(r), device=DEV) for _ in range(6)]
        torch.cuda.synchronize()
        del junk
    # new inputs of the same sizes, written in place (a captured graph serves one set of tensor sizes: the same graphs here)
    g = torch.Generator().manual_seed(9)
    new_pos = (d.pos.cpu() + 0.01 * torch.randn(d.pos.shape, generator=g)).to(DEV)
    new_flow = (0.05 * torch.randn(d.pred_flow.shape, generator=g)).to(DEV)
    other = synth.MeshData(**{k: v for k, v in d.__dict__.items()})
    other.pos, other.pred_flow = new_pos.clone(), new_flow.clone()
    want2 = [t.clone() for t in m(other, other.pred_flow)]
    d.pos.copy_(new_pos); d.pred_flow.copy_(new_flow)
    out = cf.replay()
    assert cf.check() is True
    for a, b in zip(want2, out):
        assert torch.equal(a, b)
    # an overflowing input is reported by the replay's own snapshot (MORIG_PRECISION=f32 has no range guard: nothing to report)
    from morig_amd import native
    d.pred_flow.mul_(1.0e7)
    cf.replay()
    assert cf.check() is (native.get_ops().precision == "f32")


def test_forward_server_serves_small_batches_from_captured_graphs():
    """morig_amd.serving.ForwardServer: batches of up to N vertices are served from a HIP graph captured per
    SHAPE -- a second batch of the same shape but different content is copied into the graph's static inputs and replayed -- larger
    ones run eagerly; results equal the eager forward bit for bit; an out-of-range input falls back to the eager (fp32) forward."""
    from morig_amd import native
    from morig_amd.serving import ForwardServer
    m = _jointnet(4)
    srv = ForwardServer(m, graph_max_vertices=2 * 24 * 24)
    a = synth.collate([synth.make_mesh(31, n_side=24)]).to(DEV)
    b = synth.collate([synth.make_mesh(32, n_side=24)]).to(DEV)           # same vertex / tpl-edge counts; the geo edge count may differ
    big = synth.collate([synth.make_mesh(33 + i, n_side=24) for i in range(3)]).to(DEV)
    for d in (a, b, a, big, b):
        want = [t.clone() for t in m(d, d.pred_flow)]
        got = srv(d, d.pred_flow)
        for x, y in zip(want, got):
            assert torch.equal(x, y)
    same_shape = a.geo_edge_index.shape == b.geo_edge_index.shape
    assert srv.stats["eager"] == 1 and srv.stats["captures"] == (1 if same_shape else 2) and srv.stats["replays"] == 4, srv.stats
    if native.get_ops().precision != "f32":
        hot = synth.MeshData(**{k: v for k, v in a.__dict__.items()})
        hot.pred_flow = a.pred_flow * 1.0e7
        want = m(hot, hot.pred_flow)
        got = srv(hot, hot.pred_flow)
        assert srv.stats["fallbacks"] == 1 and torch.equal(want[2], got[2])


def test_mixed_quad_edgeconv_network_equals_the_golden(monkeypatch):
    """MORIG_EDGE_MIX=1 (opt-in): the 256-wide edge layers on MORIG_CSR_MIN4 CSRs -- no padded rows, quads that straddle two
    segments split in the epilogue (edge_ws.hip <256, true, true>) -- give the reference's outputs on the 4096-vertex golden mesh and on a
    ragged batch, like the default 4-aligned form."""
    monkeypatch.setenv("MORIG_EDGE_MIX", "1")
    meta, a = load_golden("jointnet_4k")
    mesh = synth.collate([synth.make_mesh(meta["mesh_seed"], n_side=meta["n_side"])])
    m = models.jointnet_motion(**meta["kwargs"]).eval()
    synth.load_recipe(m, meta["recipe_seed"], mild=meta["mild"]).to(DEV)
    d = mesh.to(DEV)
    _, aggr, shift = m(d, d.pred_flow)
    assert rel_excess(aggr, a["motion_aggr"], TOL) <= 0 and rel_excess(shift, a["pred_shift"], TOL) <= 0
    monkeypatch.setenv("MORIG_EDGE_MIX", "0")
    _, aggr0, shift0 = m(d, d.pred_flow)
    assert maxdiff(shift, shift0) <= 2e-5 and maxdiff(aggr, aggr0) <= 2e-5
    monkeypatch.setenv("MORIG_EDGE_MIX", "1")
    meta, a = load_golden("jointnet_ragged")
    m = models.__dict__[meta["arch"]](**meta["kwargs"]).eval()
    synth.load_recipe(m, meta["recipe_seed"]).to(DEV)
    d = data_from(a, DEV)
    res = m(d, d.pred_flow)
    assert rel_excess(res[2], a["pred_shift"], TOL) <= 0


def test_range_shift_keeps_large_activations_on_the_fast_path():
    """[r06] activations beyond the split-fp16 range (|x| ~ 1e6 inside the stacks) no longer send every forward to the exact path: the
    GCNRig stacks are positively homogeneous in (inputs, additive constants), so the forward that overflows raises the model's range
    shift k (sticky), the stacks run at 2^-k -- exact powers of two in, the inverse out -- and the results stay within 1e-4 (of scale) of
    the CPU oracle WITHOUT the exact-path re-run; the next forward starts at that k and runs once."""
    from morig_amd import native
    from oracle import nets
    o = native.get_ops()
    if o.precision == "f32":
        pytest.skip("no range guard on the exact path")
    kw = dict(num_keyframes=5, chn_output=3, aggr_method="attn")
    batch = synth.make_batch([3, 4], n_side=12)
    ours = synth.load_recipe(models.jointnet_motion(**kw).eval(), 3, mild=True)
    ref = synth.load_recipe(nets.jointnet_motion(**kw).eval(), 3, mild=True)
    # blow the activations of the head up with weights that stay ordinary: the BatchNorm affine behind its first unit x 3e5, so that x_1 --
    # and, the stack being homogeneous, everything behind it up to the outputs -- is 1e5 ... 1e7 (the weights of every contraction keep
    # their magnitudes: the split-fp16 image of a weight needs the fp16 range too, in both directions)
    for net in (ours, ref):
        sd = net.state_dict()
        for k_ in ("jointnet.gcu_1.mlp.0.2.weight", "jointnet.gcu_1.mlp.0.2.bias"):
            sd[k_] = sd[k_] * 3.0e5
        net.load_state_dict(sd)
    ours = ours.to(DEV)
    with torch.no_grad():
        want = ref(batch, batch.pred_flow)
    d = batch.to(DEV)
    assert ours.range_shift == 0
    got = ours(d, d.pred_flow)
    assert ours.range_shift > 0, "the forward did not overflow: the test's weights are too tame"
    k1 = ours.range_shift
    got2 = ours(d, d.pred_flow)                                    # starts at the sticky shift: one pass, same bits
    assert ours.range_shift == k1
    for a, b in zip(got, got2):
        assert torch.equal(a, b)
    for name, g_, w_ in zip(("motion_all", "motion_aggr", "pred_shift"), got, want):
        assert rel_excess(g_, w_, TOL) <= 0, name
    # and it really is the fast path: at the sticky shift a forward with its guard read deferred reports no overflow ...
    out_async, pending = ours.forward_async(d, d.pred_flow)
    assert pending.result() is True and torch.equal(out_async[2], got[2])
    # ... while at shift 0 the same forward does
    ours.range_shift = 0
    _, pending0 = ours.forward_async(d, d.pred_flow)
    assert pending0.result() is False
    ours.range_shift = k1


def test_row_normalised_weights_keep_odd_magnitudes_accurate_and_on_the_fast_path():
    """[r06] packing.py's row normalisation. (1) A small BatchNorm scale behind the first edge Linear (gamma x 1e-5, undone behind the
    second: the network's function is unchanged) makes the folded W2 diag(s1) ~ 1e-6, an fp16 SUBNORMAL: its split image used to carry
    4-5 bits and the fast path answered with percent-level errors without any overflow to report. (2) A Linear whose weights are ~ 2e5
    (its BatchNorm statistics scaled to match) used to have no fp16 image at all: every forward re-ran on the exact path. Both are now
    ordinary fast-path layers: every row is scaled by a power of two into [4096, 8192) and the factor is undone through bias / scale."""
    from morig_amd import native, packing
    from oracle import nets
    o = native.get_ops()
    if o.precision == "f32":
        pytest.skip("the split-fp16 images are not used on the exact path")
    if packing.PACK_NORMALISE == "0":
        pytest.skip("MORIG_PACK_NORMALISE=0")
    kw = dict(num_keyframes=5, chn_output=3, aggr_method="attn")
    batch = synth.make_batch([3, 4], n_side=12)
    ours = synth.load_recipe(models.jointnet_motion(**kw).eval(), 3, mild=True)
    ref = synth.load_recipe(nets.jointnet_motion(**kw).eval(), 3, mild=True)
    with torch.no_grad():
        want0 = ref(batch, batch.pred_flow)
    c, big = 1.0e-5, 2.0e6
    for net in (ours, ref):
        sd = net.state_dict()
        for conv in ("edge_conv_tpl", "edge_conv_geo"):
            p = "jointnet.gcu_2.%s.nn_x." % conv
            for k_ in ("0.2.weight", "0.2.bias", "1.0.bias", "1.2.running_mean"):
                sd[p + k_] = sd[p + k_] * c
            sd[p + "1.2.weight"] = sd[p + "1.2.weight"] / c
        p = "jointnet.gcu_2.mlp.0."
        for k_ in ("0.weight", "0.bias", "2.running_mean"):
            sd[p + k_] = sd[p + k_] * big
        sd[p + "2.running_var"] = sd[p + "2.running_var"] * big * big
        net.load_state_dict(sd)
    with torch.no_grad():
        want = ref(batch, batch.pred_flow)
    for a, b in zip(want, want0):                                   # the rescaling left the function alone (the oracle says so)
        assert float((a - b).abs().max()) <= 1e-4 * max(1.0, float(b.abs().max()))
    ours = ours.to(DEV)
    d = batch.to(DEV)
    got, pending = ours.forward_async(d, d.pred_flow)
    assert pending.result() is True, "a layer fell off the fast path (no fp16 image of its weights?)"
    assert ours.range_shift == 0
    for name, g_, w_ in zip(("motion_all", "motion_aggr", "pred_shift"), got, want):
        assert rel_excess(g_, w_, TOL) <= 0, name
