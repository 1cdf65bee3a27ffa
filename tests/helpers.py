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


NOTES = []                 # free-text lines for the parity report (tests/conftest.py)
PARITY_LOG = []            # (test id, abs err, |ref|inf, err / max(1, |ref|inf)) of every parity check; printed by conftest at the end


def rel_excess(a, b, tol, strict=None):
    """|a-b|_inf relative to max(1, |b|_inf), minus tol: the parity criterion used for network outputs
    ("within 1e-4 fp32" of BASELINE.json, normalised by the output scale when that exceeds 1).
    The STRICT reading -- absolute |a-b|_inf <= tol whatever the scale -- is asserted as well wherever it is known to hold:
    by default for every reference whose |ref|_inf <= 4 (all committed jointnet / masknet / skinnet / corrnet goldens); ``strict=False``
    switches it off for a check (large-magnitude logits), ``strict=True`` forces it. Both numbers go to the parity report."""
    a = a.detach().cpu().double()
    b = b.detach().cpu().double()
    err, scale = (a - b).abs().max().item(), b.abs().max().item()
    PARITY_LOG.append((os.environ.get("PYTEST_CURRENT_TEST", "?").split(" ")[0], err, scale, err / max(1.0, scale)))
    if strict is None:
        strict = scale <= 4.0
    if strict and err > tol:
        return err - tol
    return err / max(1.0, scale) - tol


# ---- standalone PointNet++ blocks (SAModule / GlobalSAModule / FPModule): one case builder for the CPU and GPU suites ----
def point_module_case(kind, ratio=0.5, r=0.12, seed=0, device="cpu"):
    """-> (module under test, oracle module with the same parameters, forward args tuple) for the standalone
    forwards of models/basic_modules.py:66-86,115-138 on two ragged clouds."""
    from morig_amd import synth
    from morig_amd.models import basic_modules as bm
    from oracle import nets
    g = torch.Generator().manual_seed(seed)
    n0, n1 = 300, 211
    pos = torch.rand(n0 + n1, 3, generator=g) * 0.5
    batch = torch.cat([torch.zeros(n0, dtype=torch.long), torch.ones(n1, dtype=torch.long)])
    if kind == "sa":
        cx = 16
        mod = bm.SAModule(ratio, r, bm.MLP([cx + 3, 32, 32, 64]), max_num_neighbors=64)
        ref = nets.SetAbstraction(ratio, r, nets.mlp_stack([cx + 3, 32, 32, 64]), 64)
        args = (torch.randn(n0 + n1, cx, generator=g), pos, batch, False)
    elif kind == "sa_nox":
        mod = bm.SAModule(ratio, r, bm.MLP([3, 32, 32, 64]), max_num_neighbors=64)
        ref = nets.SetAbstraction(ratio, r, nets.mlp_stack([3, 32, 32, 64]), 64)
        args = (None, pos, batch, False)
    elif kind == "gsa":
        cx = 29
        mod = bm.GlobalSAModule(bm.MLP([cx + 3, 64, 64, 128]))
        ref = nets.GlobalSetAbstraction(nets.mlp_stack([cx + 3, 64, 64, 128]))
        args = (torch.randn(n0 + n1, cx, generator=g), pos, batch)
    else:                                                   # "fp1" / "fp3": k = 1 / 3, coarse level = every third point
        k = int(kind[2:])
        cf, cs = 24, 12
        mod = bm.FPModule(k, bm.MLP([cf + cs, 64, 32]))
        ref = nets.FeaturePropagation(k, nets.mlp_stack([cf + cs, 64, 32]))
        sel = torch.arange(0, n0 + n1, 3)
        args = (torch.randn(sel.numel(), cf, generator=g), pos[sel], batch[sel], torch.randn(n0 + n1, cs, generator=g), pos, batch)
    synth.load_recipe(mod.eval(), seed + 3)                 # randomised BN statistics, negative gamma included
    ref.eval().load_state_dict(mod.state_dict())
    mod = mod.to(device)
    args = tuple(a.to(device) if torch.is_tensor(a) else a for a in args)
    return mod, ref, args


POINT_MODULE_CASES = [("sa", 0.5, 0.12), ("sa", 0.25, 0.25), ("sa", 0.25, 0.5), ("sa_nox", 0.5, 0.12), ("gsa", 0, 0), ("fp1", 0, 0),
                      ("fp3", 0, 0)]


def check_point_module(kind, ratio, r, device):
    mod, ref, args = point_module_case(kind, ratio, r, device=device)
    cpu_args = tuple(a.cpu() if torch.is_tensor(a) else a for a in args)
    with torch.no_grad():
        got = mod(*args)
        want = ref(*cpu_args)
    assert len(got) == len(want) == 3
    assert rel_excess(got[0], want[0], 2e-5) <= 0, (kind, maxdiff(got[0], want[0]))
    assert got[1].shape == want[1].shape and maxdiff(got[1], want[1]) == 0          # positions: gathered / passed through
    assert torch.equal(got[2].cpu().long(), want[2].long())                          # batch ids: index work, bit exact


# ---- full-size harsh-recipe goldens (tests/golden/*_4k_harsh.npz, corrnet_4k_8k_harsh.npz): shared by the CPU and GPU suites ----
FULL_SIZE_GOLDENS = ["jointnet_4k_harsh", "masknet_4k_harsh", "skinnet_4k_harsh", "corrnet_4k_8k_harsh"]


def full_size_inputs(meta, device="cpu"):
    """the inputs of a full-size fixture are a pure function of its mesh seed (only outputs are stored)"""
    from morig_amd import synth
    mesh = synth.make_mesh(meta["mesh_seed"], n_side=meta["n_side"], with_skin=meta.get("with_skin", False))
    clouds = [synth.make_point_cloud(mesh, int(mesh.name), meta["n_pts"])] if "n_pts" in meta else None
    return synth.collate([mesh], clouds).to(device)


def check_full_size(model, meta, a, data, tol):
    """run ``model`` on the regenerated inputs and compare with the reference's stored outputs"""
    step = meta["row_step"]
    assert maxdiff(data.pos[:8], a["pos_check"]) == 0 and torch.equal(data.geo_edge_index[:, :32].cpu(), a["geo_check"])
    if meta["arch"] == "corrnet":
        assert maxdiff(data.pts[:8], a["pts_check"]) == 0
        ov, op, vis, _ = model(data, True, False)
        assert rel_excess(ov[::step], a["out_vtx_rows"], tol) <= 0
        assert rel_excess(op[::step], a["out_pts_rows"], tol) <= 0
        # The visibility head reads out_pts[argmax cosine similarity] (models/corrnet.py:62-65): where a vertex's two best
        # similarities tie to within the feature tolerance the choice is decided by the last bit of out_vtx / out_pts, on the
        # reference as much as here, and the other candidate gives a different (equally valid) row. Such vertices are identified
        # from OUR features and excused; every other row must match, and the excused set must stay a sliver of the mesh.
        ref_vis = a["out_vismask"].to(vis.device)
        off = (vis - ref_vis).abs().flatten() > tol
        if bool(off.any()):
            top2 = (ov.double() @ op.double().t()).topk(2, dim=1).values          # one mesh, one cloud in these fixtures
            near_tie = (top2[:, 0] - top2[:, 1]) <= 4 * tol
            assert bool(near_tie[off].all()), "a visibility row differs although its nearest point is well separated"
            # measured on MI355X (r03j): 1 of 4096 rows (fraction 0.00024); the bound leaves a factor of four
            assert float(off.float().mean()) <= 0.001, float(off.float().mean())
        NOTES.append("corrnet full-size visibility head: %d of %d rows excused as arg-max near-ties (fraction %.5f; bound 0.001)"
                     % (int(off.sum()), off.numel(), float(off.float().mean())))
        assert rel_excess(vis[~off], ref_vis[~off], tol) <= 0
    else:
        _, aggr, last = model(data, data.pred_flow)
        key = [k for k in ("pred_shift", "pred_mask", "skin_cls_pred") if k in a][0]
        assert rel_excess(aggr[::step], a["motion_aggr_rows"], tol) <= 0
        assert rel_excess(last, a[key], tol) <= 0, key


def check_radius_cpu(device):
    """radius_cpu (models/basic_modules.py:9-29) against a fixture produced by the reference's own function: the deterministic
    case bit for bit (index work), the over-full case through the properties torch.multinomial guarantees."""
    from conftest import load_golden
    from morig_amd.models.basic_modules import radius_cpu
    meta, a = load_golden("radius_cpu_kat")
    x, y = a["x"].to(device), a["y"].to(device)
    e = radius_cpu(x, y, meta["r_exact"], meta["max_exact"])
    assert e.dtype == torch.int64 and torch.equal(e.cpu(), a["edges_exact"])
    mx = meta["max_over"]
    torch.manual_seed(1)
    e1 = radius_cpu(x, y, meta["r_over"], mx).cpu()
    n_res = a["edges_over_reserved"].shape[1]
    assert torch.equal(e1[:, :n_res], a["edges_over_reserved"])                       # rows within the cap: all hits, in order
    cnt = a["counts_over"]
    over_rows = torch.nonzero(cnt > mx).flatten()
    tail = e1[:, n_res:]
    assert tail.shape[1] == over_rows.numel() * mx
    assert torch.equal(tail[1], torch.repeat_interleave(over_rows, mx))               # exactly max per over-full row, rows ascending
    d = torch.cdist(a["y"].double(), a["x"].double())
    assert bool((d[tail[1], tail[0]] <= meta["r_over"] + 1e-6).all())                 # every kept neighbour is inside the ball
    per_row = tail[0].view(-1, mx)
    assert all(len(set(r.tolist())) == mx for r in per_row)                           # without replacement
    torch.manual_seed(2)
    e2 = radius_cpu(x, y, meta["r_over"], mx).cpu()
    assert not torch.equal(e1, e2)                                                     # a new draw per call, as torch.multinomial
    torch.manual_seed(1)
    assert torch.equal(radius_cpu(x, y, meta["r_over"], mx).cpu(), e1)                # reproducible under torch.manual_seed


# ---- train-mode forward (forward half): fixtures hold the reference's own run in float32 AND float64 ----
def check_train_mode(name, device):
    """batch-statistics forward against the reference's model.train() run. Criterion: as close to the reference's float64 run as
    the reference's own float32 run is, or within 1e-4 of the output scale, whichever is larger (batch-statistics BatchNorm
    amplifies fp32 rounding through 1/sqrt(var + eps) of nearly constant channels: see oracle/make_golden.py::train_mode_fixture)."""
    from conftest import load_golden
    from morig_amd import models, synth
    meta, a = load_golden(name)
    m = models.__dict__[meta["arch"]](**meta["kwargs"])
    synth.load_recipe(m, meta["recipe_seed"], mild=meta["mild"]).to(device).train()
    d = data_from(a, device)
    got = m(d, d.pred_flow)
    assert m.training and all(not g.requires_grad for g in got)
    tid = os.environ.get("PYTEST_CURRENT_TEST", "?").split(" ")[0]
    for g, key in zip(got, ("motion_all", "motion_aggr", "head")):
        if key + "_f64" not in a:
            continue
        r64, r32 = a[key + "_f64"].double(), a[key + "_f32"].double()
        assert tuple(g.shape) == tuple(r64.shape), key
        err = (g.detach().cpu().double() - r64).abs().max().item()
        own = (r32 - r64).abs().max().item()
        scale = r64.abs().max().item()
        PARITY_LOG.append((tid + ":" + key, err, scale, err / max(1.0, scale)))
        assert err <= max(1e-4 * max(1.0, scale), own), (key, err, own, scale)
    sd = m.state_dict()
    ours = torch.cat([sd[k].detach().cpu().flatten().double() for k in meta["bn_keys"]])
    b64, b32 = a["bn_f64"].double(), a["bn_f32"].double()
    err, own, scale = (ours - b64).abs().max().item(), (b32 - b64).abs().max().item(), b64.abs().max().item()
    PARITY_LOG.append((tid + ":running_stats", err, scale, err / max(1.0, scale)))
    assert err <= max(1e-4 * max(1.0, scale), own), ("running buffers", err, own)
    assert [int(sd[k]) for k in sd if k.endswith("num_batches_tracked")] == meta["num_batches_tracked"]
    # the updated buffers feed the next eval forward: the packed-weight cache must follow the in-place update
    m.eval()
    e1 = m(d, d.pred_flow)[2]
    assert bool(torch.isfinite(e1).all())


def _models():
    from morig_amd import models
    return models


def _synth():
    from morig_amd import synth
    return synth


# ---- training steps of CorrNet / DeformNet, shared by the emulated-op CPU tests and the HIP tests ----
def _mild_bn(ref, seed):
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for m in ref.modules():
            if isinstance(m, torch.nn.BatchNorm1d):
                m.weight.copy_(torch.rand(m.num_features, generator=g) + 0.5)
                m.weight[::3] *= -1.0
                m.bias.copy_(torch.randn(m.num_features, generator=g) * 0.2)
    return g


def _grad_agreement(mine, ref, ref32=None, cos_min=0.97, median_max=2e-2):
    """gradients against the float64 oracle's. Whole-network fp32 gradients of these nets are ill-conditioned (arg-max near-ties
    route a gradient to another edge: tests/test_gpu_backward.py::_grad_report); with ``ref32`` -- the oracle's own float32
    autograd -- the criterion is that yardstick's: every tensor points the way the float64 gradient does about as well as torch's
    float32 one, the median relative error is within 5x torch's."""
    med, med32, cat_a, cat_r, cat_b = [], [], [], [], []
    r32s = dict(ref32.named_parameters()) if ref32 is not None else {}
    cosine = lambda u, v: float(torch.dot(u, v) / (u.norm() * v.norm() + 1e-300))
    for (k, p), (_, q) in zip(mine.named_parameters(), ref.named_parameters()):
        if q.grad is None:
            assert p.grad is None or float(p.grad.abs().max()) == 0.0, k
            continue
        assert p.grad is not None and p.grad.shape == q.grad.shape, k
        a, r = p.grad.detach().cpu().double().flatten(), q.grad.flatten()
        if float(r.abs().max()) == 0.0:
            continue
        med.append(float((a - r).abs().max()) / float(r.abs().max()))
        cat_a.append(a); cat_r.append(r)
        floor = cos_min
        if k in r32s and r32s[k].grad is not None:
            b = r32s[k].grad.double().flatten()
            cat_b.append(b)
            med32.append(float((b - r).abs().max()) / float(r.abs().max()))
            c32 = cosine(b, r)
            if c32 < 0.9 or a.numel() < 256:                      # float32 itself does not resolve this tensor's gradient; one flipped
                continue                                          # arg-max moves a 16-element BatchNorm gradient visibly (whole-vector check below)
            floor = min(cos_min, c32 - 0.1)
        assert cosine(a, r) >= floor, (k, floor)
    m = sorted(med)[len(med) // 2]
    assert m <= (max(median_max, 5.0 * sorted(med32)[len(med32) // 2]) if med32 else median_max), m
    whole = cosine(torch.cat(cat_a), torch.cat(cat_r))            # all parameters as one vector
    assert whole >= (min(0.97, cosine(torch.cat(cat_b), torch.cat(cat_r)) - 0.02) if cat_b else 0.97), whole


def check_corrnet_training(device, vismask):
    """SURVEY 8 f-4 / VERDICT r2 missing #4: CorrNet in model.train() through the module API (training/train_corr_pose.py:61-70) --
    batch-statistics forward, loss.backward() -- against torch.autograd on the float64 oracle: vertex branch (4 GCUs), point
    branch (FPS, ball-query PointConvs, global pooling, four feature propagations), cosine matching and the visibility head."""
    import copy
    from oracle import nets
    kw = dict(input_feature=3, output_feature=64, temprature=0.07)
    torch.manual_seed(2)
    ref = nets.CorrNet(**kw).train().double()
    g = _mild_bn(ref, 11)
    ref32 = copy.deepcopy(ref).float()                            # torch's own float32 run: the yardstick (batch-statistics BatchNorm
    mine = _models().corrnet(**kw).train()                           # through ~40 layers amplifies fp32 rounding to 1e-4 .. 1e-2)
    mine.load_state_dict(copy.deepcopy(ref32.state_dict()))
    mine.to(device)
    b = _synth().make_batch([7, 8], n_side=7, with_skin=False, n_pts=160)
    wv = torch.randn(b.vtx.shape[0], 64, generator=g)
    wp = torch.randn(b.pts.shape[0], 64, generator=g)
    wm = torch.randn(b.vtx.shape[0], 1, generator=g)              # (an unweighted sum behind a BatchNorm has zero gradient)
    bd = copy.deepcopy(b)
    bm_ = copy.deepcopy(b).to(device)
    bd.vtx, bd.pts = b.vtx.double(), b.pts.double()
    with torch.enable_grad():
        ov, op, vis, tau = ref(bd, vismask, False)
        loss = (ov * wv.double()).sum() + (op * wp.double()).sum() + ((vis * wm.double()).sum() if vismask else 0.0)
        loss.backward()
        mv, mp, mvis, mtau = mine(bm_, vismask, False)
        assert mine.training and mv.requires_grad and mtau is mine.temprature
        lm = (mv * wv.to(device)).sum() + (mp * wp.to(device)).sum() + ((mvis * wm.to(device)).sum() if vismask else 0.0)
        lm.backward()
    with torch.enable_grad():
        o32 = ref32(b, vismask, False)
        ((o32[0] * wv).sum() + (o32[1] * wp).sum() + ((o32[2] * wm).sum() if vismask else 0.0)).backward()
    for got, want, own, name in ((mv, ov, o32[0], "out_vtx"), (mp, op, o32[1], "out_pts"), (mvis, vis, o32[2], "out_vismask")):
        if want is None:
            assert got is None
            continue
        err = float((got.detach().cpu().double() - want).abs().max())
        yard = float((own.double() - want).abs().max())
        PARITY_LOG.append((os.environ.get("PYTEST_CURRENT_TEST", "?").split(" ")[0] + ":" + name, err, float(want.abs().max()),
                           err / max(1.0, float(want.abs().max()))))
        assert err <= max(5e-4 * max(1.0, float(want.abs().max())), 3.0 * yard), (name, err, yard)
    _grad_agreement(mine, ref, ref32)
    for (k, v), (_, r) in zip(mine.state_dict().items(), ref.state_dict().items()):
        if k.endswith("running_mean"):
            assert float((v.detach().cpu() - r.float()).abs().max()) <= 1e-4, k
        if k.endswith("num_batches_tracked"):
            assert int(v) == int(r), k
    # no_grad + train(): same forward, buffers move again, nothing requires grad
    with torch.no_grad():
        out = mine(bm_, vismask, False)
    assert not out[0].requires_grad and int(mine.vtx_mlp_glb[0][2].num_batches_tracked) == 2


def check_deformnet_training(device):
    """DeformNet in model.train() as training/train_deform_pose.py:149-153 runs it -- the correspondence extractor's parameters
    frozen (its BatchNorm layers still in batch-statistics mode), ``completing`` trained -- against torch.autograd on the
    float64 oracle; then with nothing frozen (the reference's in-place mask normalisation cannot backpropagate there; the
    product's out-of-place form can): every parameter group receives a finite gradient."""
    import copy
    from oracle import nets
    kw = dict(tau_nce=0.07, num_interp=5)
    torch.manual_seed(4)                                          # (an initialisation whose visibility mask keeps 1e-2 away from the
    ref = nets.DeformNet(**kw).train().double()                   # 0.5 threshold: the float32 yardstick run must split the vertices alike)
    g = _mild_bn(ref, 13)
    mine = _models().deformnet(**kw).train()
    sd32 = copy.deepcopy(ref.float().state_dict())
    mine.load_state_dict(copy.deepcopy(sd32))
    mine.to(device)
    ref.double()
    for net in (mine, ref):
        for p in net.corr_extractor.parameters():
            p.requires_grad = False
    b = _synth().make_batch([9, 10], n_side=7, with_skin=False, n_pts=160)
    bd = copy.deepcopy(b)
    bd.vtx, bd.pts = b.vtx.double(), b.pts.double()
    bm_ = copy.deepcopy(b).to(device)
    w = torch.randn(b.vtx.shape[0], 3, generator=g)
    torch.manual_seed(5)                                          # FPS start draws (random_start defaults to True, deformnet.py:41)
    with torch.enable_grad():
        pred, vf, pf, vis, tau = mine(bm_)
        (pred * w.to(device)).sum().backward()
    assert float((vis - 0.5).abs().min()) > 5e-3
    torch.manual_seed(5)
    with torch.enable_grad():                                     # the oracle votes over the product's neighbour tables (tie-prone choice)
        rpred, rvf, rpf, rvis, _ = ref(bd, neighbours=tuple(t.long().cpu() for t in mine.last_neighbours))
        (rpred * w.double()).sum().backward()
    ref32 = nets.DeformNet(**kw).train()
    ref32.load_state_dict(sd32)
    for p in ref32.corr_extractor.parameters():
        p.requires_grad = False
    torch.manual_seed(5)
    with torch.enable_grad():
        own = ref32(b, neighbours=tuple(t.long().cpu() for t in mine.last_neighbours))
        (own[0] * w).sum().backward()
    for got, want, o32, name in ((vf, rvf, own[1], "vtx_feature"), (vis, rvis, own[3], "pred_vismask"), (pred, rpred, own[0], "pred_flow")):
        err = float((got.detach().cpu().double() - want).abs().max())
        yard = float((o32.double() - want).abs().max())         # torch's own float32 run against its float64 run
        PARITY_LOG.append((os.environ.get("PYTEST_CURRENT_TEST", "?").split(" ")[0] + ":" + name, err, float(want.abs().max()),
                           err / max(1.0, float(want.abs().max()))))
        assert err <= max(5e-4 * max(1.0, float(want.abs().max())), 3.0 * yard), (name, err, yard)
    _grad_agreement(mine.completing, ref.completing, ref32.completing)
    assert all(p.grad is None for p in mine.corr_extractor.parameters())
    assert int(mine.completing.mlp_glb[0][2].num_batches_tracked) == 1 and int(mine.corr_extractor.vtx_mlp_glb[0][2].num_batches_tracked) == 1
    # nothing frozen
    for p in mine.parameters():
        p.requires_grad = True
    with torch.enable_grad():
        pred, vf, pf, vis, tau = mine(bm_)
        (pred.abs().sum() + vf.sum() + pf.sum()).backward()
    got = [k for k, p in mine.named_parameters() if p.grad is not None and float(p.grad.abs().max()) > 0]
    for prefix in ("completing.gcu_1", "corr_extractor.vtx_gcu_1", "corr_extractor.pts_sa1_module", "corr_extractor.pts_fp1_module",
                   "corr_extractor.lin_vismask"):
        assert any(k.startswith(prefix) for k in got), prefix
    assert all(bool(torch.isfinite(p.grad).all()) for _, p in mine.named_parameters() if p.grad is not None)


