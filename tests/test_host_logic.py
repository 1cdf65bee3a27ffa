"""CPU: the product's HOST logic (packing, column placement, replica handling, plan wiring) run on a
torch emulation of the op layer (tests/emulate.py) and held against golden vectors from the
reference. This does NOT exercise the HIP kernels (tests/test_gpu_*.py do, on an MI355X)."""
import pytest
import torch

import morig_amd.runtime as runtime
from conftest import load_golden
from emulate import EmuOps
from helpers import data_from, maxdiff, rel_excess
from morig_amd import models, synth
from morig_amd.models import basic_modules as bm, rignet as rn

TOL = 2e-5


@pytest.fixture(autouse=True, params=["fp32-activations", "split-activations"])
def emulated_ops(request):
    """every host-logic test runs twice: plain fp32 hand-offs, and with the split-fp16 activation layout requested
    (the emulation then checks the chunk-alignment / zero-tail rules the HIP loader relies on)."""
    ops = EmuOps()
    ops.emulate_split = request.param == "split-activations"
    runtime._test_ops = ops
    yield
    runtime._test_ops = None


def test_state_dict_keys_match_reference_layout():
    from oracle import nets
    for arch, kw in [("jointnet_motion", dict(num_keyframes=5, chn_output=3, aggr_method="attn")),
                     ("masknet_motion", dict(num_keyframes=5, chn_output=1, aggr_method="attn")),
                     ("skinnet_motion", dict(nearest_bone=5, use_Dg=False, use_Lf=False, num_keyframes=5,
                                             use_motion=True, motion_dim=32))]:
        ours = models.__dict__[arch](**kw, extra_ignored=1) if False else models.__dict__[arch](**kw)
        ref = getattr(nets, arch)(**kw)
        a, b = ours.state_dict(), ref.state_dict()
        assert list(a.keys()) == list(b.keys())
        assert all(a[k].shape == b[k].shape for k in a)


def test_factories_ignore_extra_kwargs():
    # training/train_rig.py:83 passes motion_dim to jointnet_motion; train_skin.py:88 passes aggr_method
    models.jointnet_motion(num_keyframes=5, chn_output=3, aggr_method="attn", motion_dim=32)
    models.skinnet_motion(nearest_bone=5, use_Dg=False, use_Lf=False, num_keyframes=5, use_motion=True,
                          motion_dim=32, aggr_method="attn")


def test_train_mode_of_a_standalone_block_is_refused():
    """blocks without a train-mode path (only the rignet networks have one) refuse instead of silently using running stats"""
    m = bm.GCU(3, 32).train()
    _, a = load_golden("gcu_3_32")
    with pytest.raises(NotImplementedError):
        m(a["x"], a["tpl_edge_index"], a["geo_edge_index"])


def test_edgeconvmotion_layer():
    meta, a = load_golden("edgeconvmotion_c64_h128")
    m = bm.EdgeConvMotion(nn_x=bm.MLP([128, 128, 128]), nn_pos=bm.MLP([6, 16, 16])).eval()
    synth.load_recipe(m, meta["recipe_seed"])
    assert maxdiff(m(a["pos"], a["x"], a["edge_index"]), a["out"]) <= TOL


def test_edgeconvmotion_1d_feature():
    meta, a = load_golden("edgeconvmotion_x1d")
    m = bm.EdgeConvMotion(nn_x=bm.MLP([2, 32, 32]), nn_pos=bm.MLP([6, 16, 16])).eval()
    synth.load_recipe(m, meta["recipe_seed"])
    assert maxdiff(m(a["pos"], a["x"], a["edge_index"]), a["out"]) <= TOL


def test_gcumotion_layer():
    meta, a = load_golden("gcumotion_256_512")
    m = synth.load_recipe(bm.GCUMotion(256, 512).eval(), meta["recipe_seed"])
    assert maxdiff(m(a["pos"], a["x"], a["tpl_edge_index"], a["geo_edge_index"]), a["out"]) <= TOL


def test_gcu_layer():
    meta, a = load_golden("gcu_3_32")
    m = synth.load_recipe(bm.GCU(3, 32).eval(), meta["recipe_seed"])
    assert maxdiff(m(a["x"], a["tpl_edge_index"], a["geo_edge_index"]), a["out"]) <= TOL


@pytest.mark.parametrize("name", ["gcnrig_f3_o32", "gcnrig_f64_o3"])
def test_gcnrig(name):
    meta, a = load_golden(name)
    m = synth.load_recipe(rn.GCNRig(meta["chn_feature"], meta["chn_output"]).eval(), meta["recipe_seed"])
    out = m(a["pos"], a["feature"], a["tpl_edge_index"], a["geo_edge_index"], a["batch"])
    assert maxdiff(out, a["out"]) <= TOL


def test_temporal_attention_cls_only():
    meta, a = load_golden("temporalattn_32_64")
    m = synth.load_recipe(rn.TemporalAttn(32, 2, 64, 512, 64).eval(), meta["recipe_seed"])
    assert maxdiff(m(a["x"]), a["out"]) <= TOL


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
def test_full_networks(name, outs):
    meta, a = load_golden(name)
    m = models.__dict__[meta["arch"]](**meta["kwargs"]).eval()
    synth.load_recipe(m, meta["recipe_seed"])
    d = data_from(a)
    res = m(d, d.pred_flow)
    for r, key in zip(res, outs):
        if key is not None:
            assert r.shape == a[key].shape and r.is_contiguous()
            assert rel_excess(r, a[key], TOL) <= 0, key     # skin logits reach |30|: normalise by output scale


def test_position_blocks_enter_the_unit_mlps_as_k_tails(emulated_ops):
    """[r06] under the split-activation plan the replica-invariant [pos_tpl | pos_geo] block of the wide units (gcu_2, gcu_3 of motionNet
    and of the head) is NOT copied into the 5 replicas' rows: ONE pack_tails launch per forward, the units' MLPs read it as the K tail
    (morig_gemm_args.X_tail). The fp32 plan keeps the copies. Results equal the golden either way (test_full_networks)."""
    ops = runtime._test_ops
    calls = dict(tail_gemms=0, pack=0, rep_split=0)
    gemm, pack, rep = ops.gemm, ops.pack_tails, ops.copy2d_rep

    def gemm_(*a, **k):
        calls["tail_gemms"] += k.get("x_tail") is not None
        return gemm(*a, **k)

    def pack_(*a, **k):
        calls["pack"] += 1
        return pack(*a, **k)

    def rep_(src, dst, replicas, step, src_col_step=0, split=False):
        calls["rep_split"] += bool(split and dst.cols == 32 and dst.ld in (288, 544))      # a [pos_tpl | pos_geo] chunk behind [x_tpl | x_geo]
        return rep(src, dst, replicas, step, src_col_step=src_col_step, split=split)
    ops.gemm, ops.pack_tails, ops.copy2d_rep = gemm_, pack_, rep_
    meta, a = load_golden("jointnet_ragged")
    m = synth.load_recipe(models.__dict__[meta["arch"]](**meta["kwargs"]).eval(), meta["recipe_seed"])
    d = data_from(a)
    res = m(d, d.pred_flow)
    assert rel_excess(res[2], a["pred_shift"], TOL) <= 0
    if ops.emulate_split:
        assert calls == dict(tail_gemms=4, pack=1, rep_split=0), calls
    else:
        assert calls["tail_gemms"] == 0 and calls["pack"] == 0, calls


@pytest.mark.parametrize("name,outs", [("jointnet_ragged", ("motion_all", "motion_aggr", "pred_shift")),
                                       ("masknet_ragged", ("motion_all", "motion_aggr", "pred_mask")),
                                       ("skinnet_ragged", ("motion_all", "motion_aggr", "skin_cls_pred")),
                                       ("skinnet_dg1_lf1", (None, None, "skin_cls_pred"))])
def test_range_shift_is_an_exact_rescaling_of_the_stacks(name, outs):
    """[r06] NativeModule's range shift: with range_shift = k the GCNRig / SkinNet stacks run at 2^-k -- inputs and every additive constant of
    their packs scaled down (packing.scale_additive), outputs scaled back -- and the networks still reproduce the reference's goldens:
    the bookkeeping of which tensors are scaled (positions, flows, the aggregated motion feature, biases, BN shifts, the first-layer
    bias of the x3 form; NOT the attention block between the stacks) on the op-layer emulation, where a power of two is exact."""
    meta, a = load_golden(name)
    d = data_from(a)
    m = synth.load_recipe(models.__dict__[meta["arch"]](**meta["kwargs"]).eval(), meta["recipe_seed"])
    base = m(d, d.pred_flow)
    m.range_shift = 6
    res = m(d, d.pred_flow)
    assert m.range_shift == 6
    for r, b, key in zip(res, base, outs):
        if key is not None:
            assert rel_excess(r, a[key], TOL) <= 0, key
            assert maxdiff(r, b) <= 2e-6 * max(1.0, float(b.abs().max())), key      # fp32 emulation: only rounding positions move


def test_packed_cache_invalidation():
    meta, a = load_golden("gcu_3_32")
    m = bm.GCU(3, 32).eval()
    synth.load_recipe(m, 1)
    o1 = m(a["x"], a["tpl_edge_index"], a["geo_edge_index"])
    synth.load_recipe(m, meta["recipe_seed"])           # load_state_dict must drop the packed cache
    o2 = m(a["x"], a["tpl_edge_index"], a["geo_edge_index"])
    assert maxdiff(o2, a["out"]) <= TOL and maxdiff(o1, o2) > 1e-3


def test_corrnet_state_dict_and_forward():
    from oracle import nets
    meta, a = load_golden("corrnet_ragged")
    m = models.corrnet(**meta["kwargs"]).eval()
    ref = nets.corrnet(**meta["kwargs"])
    assert list(m.state_dict().keys()) == list(ref.state_dict().keys())
    synth.load_recipe(m, meta["recipe_seed"])
    d = data_from(a)
    ov, op, vis, tau = m(d, True, False)
    assert rel_excess(ov, a["out_vtx"], TOL) <= 0
    assert rel_excess(op, a["out_pts"], TOL) <= 0
    assert rel_excess(vis, a["out_vismask"], 1e-4) <= 0
    assert tau is m.temprature
    assert m(d, False, False)[2] is None
    m(d, False, True)       # random FPS start runs


@pytest.mark.parametrize("name", ["deformnet_ragged", "deformnet_three"])
def test_deformnet_state_dict_and_forward(name):
    """the DeformNet plan (CorrNet -> mask normalisation -> k-NN votes -> GCNDeform) on the emulated op
    layer against the reference-generated fixture; FPS starts come from the recorded torch seed (deformnet.py:41)."""
    from oracle import nets
    meta, a = load_golden(name)
    m = models.deformnet(**meta["kwargs"]).eval()
    ref = nets.deformnet(**meta["kwargs"])
    assert list(m.state_dict().keys()) == list(ref.state_dict().keys())
    assert [tuple(v.shape) for v in m.state_dict().values()] == [tuple(v.shape) for v in ref.state_dict().values()]
    synth.load_recipe(m, meta["recipe_seed"], mild=meta["mild"])
    d = data_from(a)
    torch.manual_seed(meta["rng_seed"])
    pf, vf, ptf, vis, tau = m(d)
    assert rel_excess(vf, a["vtx_feature"], TOL) <= 0 and rel_excess(ptf, a["pts_feature"], TOL) <= 0
    assert rel_excess(vis, a["pred_vismask"], 1e-4) <= 0
    assert torch.equal(vis >= 0.5, a["pred_vismask"] >= 0.5)
    assert pf.shape == a["out_pred_flow"].shape and rel_excess(pf, a["out_pred_flow"], 1e-4) <= 0
    assert tau is m.corr_extractor.temprature


def test_gcndeform_argument_order():
    """GCNDeform.forward takes (pos, feature, geo_edge_index, tpl_edge_index, batch) -- geo first (deformnet.py:23)."""
    from oracle import nets
    from morig_amd.models.deformnet import GCNDeform
    _, a = load_golden("gcnrig_f3_o32")
    m = GCNDeform(chn_in=3, chn_output=3).eval()
    ref = nets.DeformGCN(3, 3).eval()
    synth.load_recipe(m, 77); synth.load_recipe(ref, 77)
    with torch.no_grad():
        want = ref(a["pos"], a["feature"], a["geo_edge_index"], a["tpl_edge_index"], a["batch"])
    got = m(a["pos"], a["feature"], a["geo_edge_index"], a["tpl_edge_index"], a["batch"])
    assert rel_excess(got, want, TOL) <= 0


def test_packed_cache_follows_in_place_edits_and_child_reloads(emulated_ops):
    """the kernel-layout weight cache must not survive (a) an in-place edit of a parameter, (b) load_state_dict
    on a plain Sequential child, (c) load_state_dict on a child NativeModule whose tensors the PARENT packs."""
    from morig_amd.models import basic_modules as bm
    torch.manual_seed(0)
    n = 40
    pos, x = torch.randn(n, 3), torch.randn(n, 8)
    ei = torch.randint(0, n, (2, 200))
    g = bm.GCUMotion(in_channels=8, out_channels=32, dim_pos_feat=16).eval()
    run = lambda: g(pos, x, ei, ei).clone()
    y0 = run()
    assert torch.equal(run(), y0)                                        # cache hit: same bits
    g.mlp[0][0].weight.mul_(2.0)                                         # (a) (the autouse fixture holds no_grad)
    y1 = run()
    assert not torch.allclose(y1, y0)
    sd = {k: v * 0.5 for k, v in g.mlp.state_dict().items()}            # (b) plain Sequential child
    g.mlp.load_state_dict(sd)
    y2 = run()
    assert not torch.allclose(y2, y1)
    sd = {k: (v * 1.5 if v.dtype.is_floating_point else v) for k, v in g.edge_conv_tpl.state_dict().items()}     # (c)
    g.edge_conv_tpl.load_state_dict(sd)
    y3 = run()
    assert not torch.allclose(y3, y2)
    fresh = bm.GCUMotion(in_channels=8, out_channels=32, dim_pos_feat=16).eval()
    fresh.load_state_dict(g.state_dict())
    assert torch.allclose(fresh(pos, x, ei, ei), y3, atol=1e-6)


@pytest.mark.parametrize("kind,ratio,r", __import__("helpers").POINT_MODULE_CASES)
def test_point_modules_standalone_forward(emulated_ops, kind, ratio, r):
    """SAModule / GlobalSAModule / FPModule are callable on their own with the reference's signatures and
    return tuples (models/basic_modules.py:74-86,121-125,133-138); host wiring on the emulated op layer vs the oracle."""
    from helpers import check_point_module
    check_point_module(kind, ratio, r, "cpu")


def test_radius_cpu_host_wiring(emulated_ops):
    from helpers import check_radius_cpu
    check_radius_cpu("cpu")


@pytest.mark.parametrize("name", ["jointnet_train", "masknet_train", "skinnet_train"])
def test_train_mode_forward_host_wiring(emulated_ops, name):
    """SURVEY 8 f-4 forward half: model.train() forward (batch-statistics BatchNorm over vertices and edges, running buffers
    updated) on the emulated op layer against the reference's own train-mode run."""
    from helpers import check_train_mode
    check_train_mode(name, "cpu")


def test_train_mode_modules_without_a_train_path_say_so(emulated_ops):
    m = bm.SAModule(0.5, 0.2, bm.MLP([3, 8, 8]), 16).train()
    with pytest.raises(NotImplementedError, match="BACKWARD|train-mode"):
        m(None, torch.zeros(4, 3), torch.zeros(4, dtype=torch.long))


@pytest.mark.parametrize("vismask", [True, False])
def test_corrnet_training_step_host_wiring(emulated_ops, vismask):
    """SURVEY 8 f-4 / VERDICT r2 missing #4: CorrNet in model.train() through the module API on the emulated op layer
    (tests/helpers.py::check_corrnet_training; the same check runs on the HIP operators in tests/test_gpu_backward.py)"""
    from helpers import check_corrnet_training
    check_corrnet_training("cpu", vismask)


def test_deformnet_training_step_host_wiring(emulated_ops):
    from helpers import check_deformnet_training
    check_deformnet_training("cpu")


def test_training_pack_cache_follows_parameter_updates(emulated_ops):
    """the per-module cache of kernel-layout weights (train_backward.packs_of) is keyed on the parameters' version counters: an
    in-place update (optimizer.step(), load_state_dict) must repack, a repeated forward must not"""
    from morig_amd import train_backward as TB
    layer = bm.MLP([6, 8])[0].train()
    x = torch.randn(20, 6, requires_grad=True)
    with torch.enable_grad():
        a = TB.mlp_layer(x, layer)
        cache = TB.packs_of(layer)
        first = cache.d["fwd"][1]
        b = TB.mlp_layer(x, layer)
        assert cache.d["fwd"][1] is first and torch.equal(a, b)
        opt = torch.optim.SGD(layer.parameters(), lr=0.5)
        b.sum().backward()
        with torch.no_grad():
            layer[0].weight.grad = torch.ones_like(layer[0].weight)
        opt.step()
        assert cache.d["wT"][1] is not None                  # the backward's transposed image sits beside the forward's
        c = TB.mlp_layer(x, layer)
        assert cache.d["fwd"][1] is not first and not torch.equal(b, c)
        fresh = copy_module = bm.MLP([6, 8])[0].train()
        fresh.load_state_dict(layer.state_dict())
        fresh[2].running_mean.copy_(layer[2].running_mean); fresh[2].running_var.copy_(layer[2].running_var)
        assert torch.allclose(TB.mlp_layer(x, fresh), TB.mlp_layer(x, layer), atol=1e-6)


def test_modules_pickle_and_deepcopy_without_device_caches(emulated_ops):
    """whole-model torch.save / deepcopy after a forward must not drag the kernel-layout cache, HIP streams or the
    last host plan along."""
    import copy, io
    meta, a = load_golden("corrnet_ragged")
    m = synth.load_recipe(models.corrnet(**meta["kwargs"]).eval(), meta["recipe_seed"])
    d = data_from(a)
    want = m(d, False, False)[0]
    m2 = copy.deepcopy(m)
    assert m2._packed is None and m2.last_plan is None and m2._streams == {}
    buf = io.BytesIO()
    torch.save(m, buf)
    buf.seek(0)
    m3 = torch.load(buf, weights_only=False)
    assert torch.equal(m3(d, False, False)[0], want) and torch.equal(m2(d, False, False)[0], want)


def test_train_backward_host_wiring(emulated_ops):
    """SURVEY 8 f-4 backward half: the autograd blocks of morig_amd/train_backward.py on the emulated op layer -- a GCNRig
    training step (12 edge MLPs, pooling, dense layers) against torch.autograd on the oracle. The operators' formulas are
    emulated in torch here; the HIP kernels themselves are held to the same references in tests/test_gpu_backward.py."""
    import copy
    from morig_amd import train_backward as TB
    from oracle import nets
    kw = dict(chn_feature=3, chn_output=32)
    ref = nets.RigGCN(**kw).train().double()
    g = torch.Generator().manual_seed(3)
    with torch.no_grad():
        for m in ref.modules():
            if isinstance(m, torch.nn.BatchNorm1d):
                m.weight.copy_(torch.rand(m.num_features, generator=g) + 0.5)
                m.weight[::3] *= -1.0
                m.bias.copy_(torch.randn(m.num_features, generator=g) * 0.2)
    mine = rn.GCNRig(**kw).train()
    mine.load_state_dict(copy.deepcopy(ref.float().state_dict()))
    ref.double()
    b = synth.make_batch([5, 6], n_side=7, with_skin=False)
    feat = torch.randn(b.pos.shape[0], 3, generator=g) * 0.05
    w = torch.randn(b.pos.shape[0], 32, generator=g)
    with torch.enable_grad():
        o = ref(b.pos.double(), feat.double(), b.tpl_edge_index, b.geo_edge_index, b.batch)
        (o * w.double()).sum().backward()
        st = TB.graph_state(b)
        om = TB.gcnrig(mine, b.pos.float(), feat, st["csr_tpl"], st["csr_geo"], st["batch"], st["mesh_ptr"], st["ng"])
        (om * w).sum().backward()
    assert rel_excess(om, o, 5e-4, strict=False) <= 0          # fp32 train-mode forward against the float64 run (98 vertices)
    med = []
    for (k, p), (_, q) in zip(mine.named_parameters(), ref.named_parameters()):
        assert p.grad is not None and p.grad.shape == q.grad.shape, k
        a, r = p.grad.double().flatten(), q.grad.flatten()
        med.append(float((a - r).abs().max()) / max(float(r.abs().max()), 1e-12))
        assert float(torch.dot(a, r) / (a.norm() * r.norm() + 1e-300)) >= 0.97, k       # fp32 conditioning: see test_gpu_backward.py
    assert sorted(med)[len(med) // 2] <= 2e-2
    for (k, v), (_, r) in zip(mine.state_dict().items(), ref.state_dict().items()):
        if k.endswith("running_mean"):
            assert maxdiff(v, r.float()) <= 1e-4, k


def _gcnrig_step(monkeypatch, stacked=True, sums=None):
    """one GCNRig training step on the emulated op layer -> (output, {parameter: gradient}, {buffer: value})"""
    import copy
    from morig_amd import train_backward as TB
    if not stacked:
        monkeypatch.setattr(TB, "stacked_pos_branches", lambda *a, **k: None)
    if sums is not None:
        monkeypatch.setenv("MORIG_TRAIN_EDGE_SUMS", sums)
    torch.manual_seed(11)
    net = rn.GCNRig(chn_feature=3, chn_output=8).train()
    g = torch.Generator().manual_seed(4)
    with torch.no_grad():
        for m in net.modules():
            if isinstance(m, torch.nn.BatchNorm1d):
                m.weight.copy_(torch.rand(m.num_features, generator=g) + 0.5)
                m.bias.copy_(torch.randn(m.num_features, generator=g) * 0.2)
    b = synth.make_batch([2, 9], n_side=6, with_skin=False)
    feat = torch.randn(b.pos.shape[0], 3, generator=g) * 0.05
    w = torch.randn(b.pos.shape[0], 8, generator=g)
    with torch.enable_grad():
        st = TB.graph_state(b)
        out = TB.gcnrig(net, b.pos.float(), feat, st["csr_tpl"], st["csr_geo"], st["batch"], st["mesh_ptr"], st["ng"])
        (out * w).sum().backward()
    return out.detach(), {k: p.grad.clone() for k, p in net.named_parameters()}, {k: v.clone() for k, v in net.named_buffers()}


def test_stacked_position_branches_are_the_three_layers(emulated_ops, monkeypatch):
    """round 4: the three 16-wide position branches of a GCNRig on one graph run as ONE edge MLP (rows of Linear1 stacked, Linear2
    block-diagonal, BatchNorm columns side by side): same outputs, same gradients for every parameter, same running buffers and
    batch counters as the three layers evaluated one by one"""
    o1, g1, b1 = _gcnrig_step(monkeypatch, stacked=True)
    o2, g2, b2 = _gcnrig_step(monkeypatch, stacked=False)
    assert maxdiff(o1, o2) <= 2e-6 * float(o2.abs().max())
    for k in g2:
        assert maxdiff(g1[k], g2[k]) <= 1e-5 * max(float(g2[k].abs().max()), 1e-6), k
    for k in b2:
        assert maxdiff(b1[k].float(), b2[k].float()) <= 1e-6 * max(float(b2[k].float().abs().max()), 1.0), k
    assert any(k.endswith("nn_pos.0.2.num_batches_tracked") and int(v) == 1 for k, v in b1.items())


def test_edge_sums_from_products_equal_the_pass(emulated_ops, monkeypatch):
    """round 4: the first edge layer's BatchNorm sums from M = du2^T Z1, db2 and W2 against the pass over dh and Z1"""
    o1, g1, _ = _gcnrig_step(monkeypatch, sums="products")
    o2, g2, _ = _gcnrig_step(monkeypatch, sums="pass")
    assert torch.equal(o1, o2)
    for k in g2:
        assert maxdiff(g1[k], g2[k]) <= 1e-5 * max(float(g2[k].abs().max()), 1e-6), k


def test_csr_transposed_walks_every_edge_once_by_source():
    from morig_amd.native import CSR
    g = torch.Generator().manual_seed(8)
    n, E, cap = 37, 200, 260
    dst = torch.sort(torch.randint(0, n, (E,), generator=g))[0]
    src = torch.randint(0, n, (E,), generator=g)
    rowptr = torch.searchsorted(dst, torch.arange(n + 1)).int()
    junk = torch.randint(-5, 1000, (cap - E,), generator=g).int()                  # rows past the live count hold anything
    csr = CSR(rowptr, torch.cat([src.int(), junk]), torch.cat([dst.int(), junk]), n, cap, torch.zeros(1, dtype=torch.int32))
    rt, pt = csr.transposed()
    assert rt.shape == (n + 1,) and int(rt[0]) == 0 and int(rt[-1]) == E and pt.shape == (cap,)
    assert torch.equal(torch.sort(pt[:E].long())[0], torch.arange(E))
    for u in range(n):
        rows = pt[int(rt[u]):int(rt[u + 1])].long()
        assert bool((src[rows] == u).all()) and bool((rows[1:] > rows[:-1]).all())   # its edges, in ascending row order
    assert csr.transposed()[1] is pt                                                 # built once


def test_canonical_csr_does_not_depend_on_the_slot_order_of_the_build():
    """round 4: `canonical_csr` (training graphs) -- whatever order the build left a target's edges in, the segments come out sorted
    by source, rows past the live count stay behind; the result is a NEW CSR (no cached transpose), the argument is left as built"""
    from morig_amd.native import CSR
    from morig_amd.train_backward import canonical_csr
    g = torch.Generator().manual_seed(3)
    n, E, cap = 29, 150, 190
    dst = torch.sort(torch.randint(0, n, (E,), generator=g))[0]
    src = torch.randint(0, n, (E,), generator=g)
    rowptr = torch.searchsorted(dst, torch.arange(n + 1)).int()
    outs = []
    for seed in (0, 1, 2):
        gp = torch.Generator().manual_seed(seed)
        s2 = src.clone()
        for v in range(n):                                                   # shuffle inside every segment: what an atomic cursor does
            a, b = int(rowptr[v]), int(rowptr[v + 1])
            s2[a:b] = s2[a:b][torch.randperm(b - a, generator=gp)]
        junk = torch.randint(0, n, (cap - E,), generator=gp).int()
        csr = CSR(rowptr, torch.cat([s2.int(), junk]), torch.cat([dst.int(), junk]), n, cap, torch.zeros(1, dtype=torch.int32))
        built = csr.src.clone()
        csr.transposed()
        raw, csr = csr, canonical_csr(csr)
        assert csr is not raw and torch.equal(raw.src, built) and raw._transposed is not None
        assert csr._transposed is None and torch.equal(csr.dst[:E].long(), dst)
        for v in range(n):
            seg = csr.src[int(rowptr[v]):int(rowptr[v + 1])]
            assert bool((seg[1:] >= seg[:-1]).all())
        outs.append(csr.src[:E].clone())
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2])


def test_row_normalisation_is_an_exact_refactoring_of_the_layer():
    """packing.py [r06]: every output row of a split-packed Linear is multiplied by a power of two (largest entry into [4096, 8192)) and
    the factor is undone through bias / scale, so s act(W x + b) + t is unchanged -- bit for bit in exact arithmetic; rows of any
    magnitude (1e-7 ... 1e6) then have a full-precision split-fp16 image, and a row-bias producer is coupled to its consumer's factors"""
    import torch
    from morig_amd import packing
    if packing.PACK_NORMALISE == "0":
        pytest.skip("MORIG_PACK_NORMALISE=0")
    torch.manual_seed(5)
    lin = torch.nn.Linear(40, 24)
    bn = torch.nn.BatchNorm1d(24).eval()
    with torch.no_grad():
        lin.weight.mul_(torch.logspace(-7, 6, 24)[:, None])
        lin.bias.mul_(torch.logspace(-7, 6, 24))
        bn.running_var.uniform_(0.5, 2.0), bn.running_mean.normal_(), bn.weight.normal_(), bn.bias.normal_()
    pk = packing.pack_linear(lin.weight, lin.bias, bn)
    f = pk.row_factor
    assert f is not None and torch.equal(torch.exp2(torch.log2(f).round()), f)                    # powers of two
    m = pk.W[:24].abs().amax(1)
    assert bool(((m >= 4096) & (m < 8192)).all()) and bool((f[24:] == 1).all())
    assert pk.Wsplit is not None                                                                   # 1e6-scale rows: an image all the same
    img = packing.unsplit_f16(pk.Wsplit, pk.W.shape[1])
    assert float((img - pk.W).abs().max()) <= 8192 * 2.0 ** -21                                  # 22 bits of the row maximum, every row
    x = torch.randn(64, 40, dtype=torch.float64)
    want = bn.double()(torch.relu(lin.double()(x)))
    got = pk.scale[:24].double() * torch.relu(x @ pk.W[:24, :40].double().T + pk.bias[:24].double()) + pk.shift[:24].double()
    assert torch.allclose(got, want.detach(), rtol=1e-6, atol=1e-9 * float(want.abs().max()))
    # the row-bias producer of a layer arrives in that layer's row units
    g = packing.couple_rowbias(packing.pack_linear(torch.randn(24, 16)), pk)
    own = g.row_factor[:24] if g.row_factor is not None else 1.0
    assert torch.allclose(g.scale[:24] * own, f[:24])
    # a layer with ordinary rows is left alone ("auto") ...
    assert packing.PACK_NORMALISE != "auto" or packing.pack_linear(torch.randn(24, 16) * 0.1, torch.randn(24)).scale is None
    # ... but not one whose rows the fp16 split would truncate, or could not hold at all
    for mag in (1e-5, 3e5):
        q = packing.pack_linear(torch.randn(24, 16) * mag, torch.randn(24))
        assert q.row_factor is not None and q.Wsplit is not None and q.scale is not None
    # the second edge Linear: sign(s2) (which the kernels' max trick reads) is untouched
    W2, b2, s2 = packing._normalise_edge(torch.randn(32, 32) * 1e-5, torch.randn(32), -torch.rand(32) - 0.1)
    assert bool((s2 < 0).all()) and 4096 <= float(W2.abs().amax(1).min())
