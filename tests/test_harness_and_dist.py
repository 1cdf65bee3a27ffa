"""CPU: caller-side harness (post-ops, writers) against fixtures captured from the reference, the
C-ABI export list against include/morig_hip.h, and the multi-rank path on gloo (world_size 2)."""
import os
import re
import subprocess
import sys
import tempfile

import numpy as np
import pytest
import torch

from conftest import ROOT, load_golden
from helpers import maxdiff
from morig_amd import harness


def test_ply_writer_bytes_match_reference():
    _, a = load_golden("ply_writer_kat")
    assert harness.ply_bytes(a["pts"]) == bytes(a["ply_bytes"].numpy().tobytes())


def test_post_ops_match_reference_captures():
    _, j = load_golden("jointnet_ragged")
    assert maxdiff(harness.joint_positions(j["pred_shift"], j["pos"]), j["y_pred"]) == 0
    _, m = load_golden("masknet_ragged")
    assert maxdiff(harness.attention_probability(m["pred_mask"]), m["attn"]) == 0
    _, s = load_golden("skinnet_ragged")
    assert maxdiff(harness.skin_probability(s["skin_cls_pred"]), s["skin_softmax"]) == 0


def test_eval_output_files_roundtrip():
    _, j = load_golden("jointnet_ragged")
    with tempfile.TemporaryDirectory() as td:
        harness.write_eval_outputs(td, [7, 9], j["batch"], y_pred=j["y_pred"], attn=torch.sigmoid(j["pred_shift"][:, :1]))
        # evaluate/eval_rigging.py reads the ply by skipping 7 header lines (utils/io_utils.py:18-26)
        lines = open(os.path.join(td, "9.ply")).read().splitlines()
        pts = np.array([[float(v) for v in ln.split()] for ln in lines[7:]])
        sel = (j["batch"] == 1).numpy()
        assert pts.shape == (int(sel.sum()), 3)
        assert np.abs(pts - j["y_pred"].numpy()[sel]).max() < 1e-6
        assert np.load(os.path.join(td, "7_attn.npy")).shape == (int((~sel).sum()), 1)


def test_c_abi_exports_every_declared_symbol():
    """the shared library loads without a GPU and exports exactly what include/morig_hip.h declares."""
    from morig_amd import native
    hdr = open(os.path.join(ROOT, "include", "morig_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(morig_[a-z0-9_]+)\s*\(", hdr))
    assert declared == set(native.EXPORTS), declared ^ set(native.EXPORTS)
    lib = native.load_library()
    for name in declared:
        assert hasattr(lib, name)
    assert lib.morig_abi_version() == 3 == native.ABI_VERSION
    assert lib.morig_strerror(-2).decode().startswith("unsupported")


def test_c_abi_argument_structs_carry_their_size():
    """ABI 3 (include/morig_hip.h): every argument struct starts with struct_size. The ctypes mirrors have the header's sizes; a struct
    whose struct_size is missing (0), shorter than the version-3 layout, or a version-2 struct (no struct_size member: its first word
    is M / H) is refused with MORIG_E_INVALID before anything is read or launched -- so this runs without a GPU."""
    import ctypes as C
    from morig_amd import native
    hdr = open(os.path.join(ROOT, "include", "morig_hip.h")).read()
    sizes = {k: int(v) for k, v in re.findall(r"#define MORIG_([A-Z0-9_]+)_ARGS_V3_SIZE\s+(\d+)u", hdr)}
    mirrors = dict(GEMM=native.GemmArgs, EDGECONV=native.EdgeConvArgs, EDGECONV_X3=native.EdgeConvX3Args, SEGMAX=native.SegmaxArgs,
                   POINTCONV=native.PointConvArgs)
    assert set(sizes) == set(mirrors)
    current = {k: int(v) for k, v in re.findall(r"#define MORIG_([A-Z0-9_]+)_ARGS_SIZE\s+(\d+)u", hdr)}     # structs that grew since
    assert set(current) == {"EDGECONV", "EDGECONV_X3"} and all(current[k] > sizes[k] and current[k] % 8 == 0 for k in current)
    for k, cls in mirrors.items():
        assert C.sizeof(cls) == current.get(k, sizes[k]), (k, C.sizeof(cls), sizes[k])     # the mirrors have the header's CURRENT sizes
        assert cls._fields_[0] == ("struct_size", C.c_uint32)
        assert native._args(cls).struct_size == C.sizeof(cls)
    lib = native.load_library()
    calls = dict(GEMM=lib.morig_gemm, EDGECONV=lib.morig_edgeconv, EDGECONV_X3=lib.morig_edgeconv_x3, SEGMAX=lib.morig_segmax_gemm,
                 POINTCONV=lib.morig_pointconv_fused)
    for k, cls in mirrors.items():
        for bad in (0, 8, sizes[k] - 8, sizes[k] + 4, 4096):
            a = cls()
            a.struct_size = bad
            assert calls[k](C.byref(a), None) == -1, (k, bad)                    # MORIG_E_INVALID
    assert lib.morig_edgeconv_can_split_out(C.byref(native.EdgeConvArgs())) == 0
    # a caller built against the 192-byte round-5/6 struct (no pair members) is still taken: same answer as the full struct with NULL pointers
    a = native._args(native.EdgeConvArgs)
    a.struct_size = sizes["EDGECONV"]
    assert lib.morig_edgeconv(C.byref(a), None) == -1 and lib.morig_edgeconv_can_split_out(C.byref(a)) == 0

    class GemmArgsV2(C.Structure):             # the round-5 layout: M first, no struct_size, no K tail
        _fields_ = [f for f in native.GemmArgs._fields_ if f[0] not in ("struct_size", "X_tail", "ld_tail", "tail_rows", "tail_cols")]
    old = GemmArgsV2()
    old.M, old.N, old.K = 1310720, 512, 544
    buf = (C.c_char * 256)()                   # (room behind it: the call must not depend on what follows a short struct)
    C.memmove(buf, C.byref(old), C.sizeof(old))
    assert lib.morig_gemm(C.cast(buf, C.POINTER(native.GemmArgs)), None) == -1
    # a LONGER struct from a newer caller is fine: the members this build knows are read, the rest ignored (here: all pointers NULL -> invalid)
    a = native._args(native.GemmArgs)
    assert lib.morig_gemm(C.byref(a), None) == -1


def test_product_refuses_to_run_without_gpu():
    if torch.cuda.is_available():
        return
    from morig_amd import models, native
    import pytest
    _, a = load_golden("gcu_3_32")
    m = models.basic_modules.GCU(3, 32).eval() if hasattr(models, "basic_modules") else None
    from morig_amd.models import basic_modules as bm
    with pytest.raises(native.MorigNativeError):
        bm.GCU(3, 32).eval()(a["x"], a["tpl_edge_index"], a["geo_edge_index"])


_WORKER = r"""
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, os.path.join(sys.argv[1], 'tests'))
from morig_amd import dist as mdist, models, synth
import morig_amd.runtime as runtime
from emulate import EmuOps
runtime._test_ops = EmuOps()          # host logic on CPU; the collective path is what is under test
dist.init_process_group('gloo', init_method='tcp://127.0.0.1:%s' % sys.argv[2], rank=int(sys.argv[3]), world_size=2)
rank = dist.get_rank()
seeds_sides = [(1, 8), (2, 10), (3, 8), (4, 6), (5, 10)]
meshes = [synth.make_mesh(s, n_side=n) for s, n in seeds_sides]
mine = mdist.shard_items(meshes, rank, 2)
m = synth.load_recipe(models.jointnet_motion(num_keyframes=5, chn_output=3, aggr_method='attn').eval(), 5, mild=True)
with torch.no_grad():
    b = synth.collate(mine)
    shift = m(b, b.pred_flow)[2]
    allshift = mdist.all_gather_rows(shift)                      # ragged: count exchange + padded gather
    eq = mdist.all_gather_rows(torch.full((4, 3), float(rank)), equal_rows=True)
    assert eq.shape == (8, 3) and eq[:4].eq(0).all() and eq[4:].eq(1).all()
    # single-process reference over the rank-concatenated order
    order = mdist.unshard_order(len(meshes), 2)
    full = synth.collate([meshes[i] for i in order])
    want = m(full, full.pred_flow)[2]
    err = (allshift - want).abs().max().item()
    assert allshift.shape == want.shape and err < 2e-5, err
dist.barrier()                                   # nobody tears its sockets down while a peer is still inside a collective:
dist.destroy_process_group()                     # without this a rank's exit raced the others' and aborted now and then (-6 after 'ok')
print('rank', rank, 'ok')
"""


def test_two_rank_gloo_shard_and_all_gather():
    import socket
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    with tempfile.NamedTemporaryFile("w", suffix=".py", delete=False) as f:
        f.write(_WORKER)
    try:
        procs = [subprocess.Popen([sys.executable, f.name, ROOT, str(port), str(r)], stdout=subprocess.PIPE,
                                  stderr=subprocess.STDOUT, text=True) for r in range(2)]
        outs = [p.communicate(timeout=300)[0] for p in procs]
        for r, (p, o) in enumerate(zip(procs, outs)):
            assert p.returncode == 0 and f"rank {r} ok" in o, o
    finally:
        os.unlink(f.name)


_TRAIN_WORKER = r"""
import copy, os, sys, torch, torch.distributed as dist
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, os.path.join(sys.argv[1], 'tests'))
from morig_amd import dist as mdist, models, synth, train_backward as TB
import morig_amd.runtime as runtime
from emulate import EmuOps
runtime._test_ops = EmuOps()
dist.init_process_group('gloo', init_method='tcp://127.0.0.1:%s' % sys.argv[2], rank=int(sys.argv[3]), world_size=2)
rank = dist.get_rank()
torch.manual_seed(7)                                              # the same initial weights on both ranks
net = models.rignet.GCNRig(chn_feature=3, chn_output=8).train()
g = torch.Generator().manual_seed(3)
for m in net.modules():
    if isinstance(m, torch.nn.BatchNorm1d):
        with torch.no_grad():
            m.weight.copy_(torch.rand(m.num_features, generator=g) + 0.5); m.weight[::3] *= -1.0
meshes = [synth.make_mesh(s, n_side=n, with_skin=False) for s, n in [(1, 7), (2, 8), (3, 6), (4, 7)]]


def step(model, batch_meshes, sync):
    TB.set_batchnorm_sync(True if sync else None)
    b = synth.collate(batch_meshes)
    st = TB.graph_state(b)
    feat = b.pred_flow[:, :3].float()
    out = TB.gcnrig(model, b.pos.float(), feat, st['csr_tpl'], st['csr_geo'], st['batch'], st['mesh_ptr'], st['ng'])
    w = torch.sin(b.pos[:, :1].float() * 50.0 + b.pos[:, 1:2].float() * 31.0 + torch.arange(out.shape[1], dtype=torch.float32))   # a function of the VERTEX, not of its row in the shard
    (out * w).sum().backward()
    return out.detach()


# sharded: each rank its meshes, BatchNorm moments and backward sums all-reduced; parameter gradients summed afterwards
mine = mdist.shard_items(meshes, rank, 2)
sharded = copy.deepcopy(net)
out_local = step(sharded, mine, sync=True)
for p in sharded.parameters():
    dist.all_reduce(p.grad)
# single process over the whole batch, in the rank-concatenated order
order = mdist.unshard_order(len(meshes), 2)
full = copy.deepcopy(net)
out_full = step(full, [meshes[i] for i in order], sync=False)
out_all = mdist.all_gather_rows(out_local)
assert out_all.shape == out_full.shape
e_out = float((out_all - out_full).abs().max()) / max(1.0, float(out_full.abs().max()))
assert e_out < 2e-4, e_out
# two fp32 runs that differ only in the summation order of the statistics: a 1e-6 perturbation of the input moves these gradients
# by 6e-4 (median over tensors) to 7e-3 (worst) of a tensor's scale (measured), so that is the band; the blocks are held tighter below
errs, worst, low = [], 0.0, 0
for (k, p), (_, q) in zip(sharded.named_parameters(), full.named_parameters()):
    a, r = p.grad.double().flatten(), q.grad.double().flatten()
    cos = float(torch.dot(a, r) / (a.norm() * r.norm() + 1e-300))
    worst = max(worst, 1.0 - cos)
    errs.append(float((a - r).abs().max()) / max(float(r.abs().max()), 1e-9))
    assert cos >= 0.999, (k, cos)
assert sorted(errs)[len(errs) // 2] <= 5e-3 and max(errs) <= 5e-2, (sorted(errs)[len(errs) // 2], max(errs))
for (k, v), (_, r) in zip(sharded.state_dict().items(), full.state_dict().items()):
    if k.endswith('running_var') or k.endswith('running_mean'):
        assert float((v - r).abs().max()) <= 1e-4 * max(1.0, float(r.abs().max())), k
# and without the synchronisation the shards really do disagree with the full batch
unsynced = copy.deepcopy(net)
out_u = mdist.all_gather_rows(step(unsynced, mine, sync=False))
assert float((out_u - out_full).abs().max()) > 10 * float((out_all - out_full).abs().max())
# ---- the two blocks on their own: well conditioned, so sharded == full tightly ----
from oracle import nets
torch.manual_seed(11)
layer = nets.mlp_stack([12, 16])[0].train()
conv = nets.EdgeMaxConv(6, 16).train()
for m in list(layer.modules()) + list(conv.modules()):
    if isinstance(m, torch.nn.BatchNorm1d):
        with torch.no_grad():
            m.weight.copy_(torch.rand(m.num_features, generator=g) + 0.5); m.weight[::3] *= -1.0
x = torch.randn(400, 12, generator=g); wd = torch.randn(400, 16, generator=g)
ei = torch.stack([torch.randint(0, 200, (1500,), generator=g), torch.randint(0, 200, (1500,), generator=g)])
xe = torch.randn(400, 6, generator=g); we = torch.randn(400, 16, generator=g)
ops = runtime.get_ops()


def dense(model, rows, sync):
    TB.set_batchnorm_sync(True if sync else None)
    xi = x[rows].clone().requires_grad_(True)
    (TB.mlp_layer(xi, model) * wd[rows]).sum().backward()
    return xi.grad


def edge(model, half, sync):                     # two disjoint 200-vertex graphs: rank r owns vertices [200 r, 200 r + 200)
    TB.set_batchnorm_sync(True if sync else None)
    parts = [0, 1] if half is None else [half]
    xi = torch.cat([xe[200 * h:200 * h + 200] for h in parts]).clone().requires_grad_(True)
    e = torch.cat([ei + 200 * i for i in range(len(parts))], 1)
    csr = ops.csr_build(e, 200 * len(parts))
    wsel = torch.cat([we[200 * h:200 * h + 200] for h in parts])
    (TB.edge_mlp(xi, csr, model.nn_pos) * wsel).sum().backward()
    return xi.grad


for name, fn, mod, shard_arg in (('dense', dense, layer, slice(200 * rank, 200 * rank + 200)), ('edge', edge, conv, rank)):
    a, b = copy.deepcopy(mod), copy.deepcopy(mod)
    gx_s = fn(a, shard_arg, True)
    for p in a.parameters():
        dist.all_reduce(p.grad)
    gx_f = fn(b, slice(0, 400) if name == 'dense' else None, False)
    ref_x = gx_f[200 * rank:200 * rank + 200]
    assert float((gx_s - ref_x).abs().max()) <= 1e-3 * float(gx_f.abs().max()), (name, 'dx')
    for (k, p), (_, q) in zip(a.named_parameters(), b.named_parameters()):
        assert float((p.grad - q.grad).abs().max()) <= 1e-3 * max(float(q.grad.abs().max()), 1e-9), (name, k)
TB.set_batchnorm_sync(None)
dist.barrier()
dist.destroy_process_group()
print('rank', rank, 'ok', e_out, worst)
"""


def test_two_rank_gloo_sharded_training_step_with_synchronised_batchnorm():
    """SURVEY 8(e) caveat / f-4: a train-mode step sharded over two ranks with the cross-rank BatchNorm statistics of
    morig_amd/train_forward.py (forward moments and the backward's two sums all-reduced) reproduces the single-process step
    over the whole batch: outputs, summed parameter gradients, running buffers."""
    import socket
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    with tempfile.NamedTemporaryFile("w", suffix=".py", delete=False) as f:
        f.write(_TRAIN_WORKER)
    try:
        procs = [subprocess.Popen([sys.executable, f.name, ROOT, str(port), str(r)], stdout=subprocess.PIPE,
                                  stderr=subprocess.STDOUT, text=True) for r in range(2)]
        outs = [p.communicate(timeout=600)[0] for p in procs]
        for r, (p, o) in enumerate(zip(procs, outs)):
            assert p.returncode == 0 and f"rank {r} ok" in o, o[-3000:]
    finally:
        os.unlink(f.name)


def _bench_line(extra, env_extra=None):
    import json
    env = dict(os.environ, OMP_NUM_THREADS="2")
    env.pop("WORLD_SIZE", None); env.pop("RANK", None); env.pop("LOCAL_RANK", None)
    env.update(env_extra or {})
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "bench_plumbing.py")] + extra, env=env, stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    out = p.stdout.splitlines()
    lines = [ln for ln in out if ln.startswith("{")]
    assert len(lines) == 1, p.stdout            # rank 0 prints ONE JSON line ...
    # ... and it is the LAST line of stdout, compact enough for the driver's stdout tail (a 21 KB line lost the record);
    # the detail goes out before it on '[bench-detail]' lines
    assert out[-1] == lines[0] and len(lines[0]) < 4096, (len(lines[0]), out[-1][:200])
    assert any(ln.startswith("[bench-detail] ") for ln in out[:-1])
    r = json.loads(lines[0])
    assert isinstance(r["roofline"], dict) and "cpu_baseline" in r and "frac" in r["roofline"] and "traffic" in r["roofline"]
    return r


def test_bench_self_launches_two_ranks_from_a_bare_shell():
    """`python bench.py --gpus 2` with no launcher around it (how the driver invoked --gpus 1 in round 1) must spawn its
    own ranks; plumbing mode = gloo + CPU emulation on tiny meshes, so this checks launch, sharding, the collective and the
    JSON contract, not speed."""
    r = _bench_line(["--gpus", "2", "--steps", "2", "--warmup", "1", "--batch", "2", "--cpu-seconds", "0"])
    assert r["n_gpus"] == 2 and r["rccl_ranks"] == 2 and r["backend"] == "gloo"
    assert r["scaling"] == "weak" and r["config"]["global_batch"] == 4 and r["config"]["meshes_per_gpu"] == 2
    assert len(r["per_rank_ms_per_step"]) == 2 and r["steps"] == 2 and r["warmup"] == 1
    assert r["ms_per_step_p10"] <= r["ms_per_step_median"] <= r["ms_per_step_p90"]
    assert abs(r["value"] - 4 * 2 / (r["ms_per_step"] * 2e-3)) / r["value"] < 0.02      # whole-job units / max-rank time
    assert "PLUMBING" in r["data"] and r["higher_is_better"] is True and r["vs_baseline"] is None


def test_bench_strong_scaling_splits_one_batch():
    r = _bench_line(["--gpus", "2", "--steps", "1", "--warmup", "0", "--batch", "4", "--scaling", "strong", "--cpu-seconds", "0"])
    assert r["scaling"] == "strong" and r["config"]["global_batch"] == 4 and r["config"]["meshes_per_gpu"] == 2


def test_bench_single_rank_line_keeps_the_contract():
    r = _bench_line(["--steps", "2", "--warmup", "1", "--batch", "2", "--cpu-seconds", "0"])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in r, k
    assert r["n_gpus"] == 1 and r["rccl_ranks"] == 1 and r["unit"] == "meshes/s"
    assert "workload" in r["config"] and "model" not in r["config"]


# ---- world size 8: the node the north star names has eight ranks; nothing here can measure it, so the plumbing
# ---- (launch, sharding, ragged and equal-row gathers, the line's contract) is run at that world size on gloo
_WORKER8 = r"""
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, os.path.join(sys.argv[1], 'tests'))
torch.set_num_threads(1)
from morig_amd import dist as mdist, models, synth
import morig_amd.runtime as runtime
from emulate import EmuOps
runtime._test_ops = EmuOps()
W = 8
dist.init_process_group('gloo', init_method='tcp://127.0.0.1:%s' % sys.argv[2], rank=int(sys.argv[3]), world_size=W)
rank = dist.get_rank()
# 11 meshes of three sizes over 8 ranks: ranks 0..2 hold two meshes, the others one -- ragged row counts on every rank
sides = [6, 8, 5, 7, 6, 8, 5, 7, 6, 8, 5]
meshes = [synth.make_mesh(20 + i, n_side=n) for i, n in enumerate(sides)]
mine = mdist.shard_items(meshes, rank, W)
assert len(mine) == (2 if rank < 3 else 1)
m = synth.load_recipe(models.jointnet_motion(num_keyframes=5, chn_output=3, aggr_method='attn').eval(), 5, mild=True)
with torch.no_grad():
    b = synth.collate(mine)
    shift = m(b, b.pred_flow)[2]
    allshift = mdist.all_gather_rows(shift)                      # ragged: count exchange + padded gather
    order = mdist.unshard_order(len(meshes), W)
    assert sorted(order) == list(range(len(meshes)))
    full = synth.collate([meshes[i] for i in order])
    want = m(full, full.pred_flow)[2]
    err = (allshift - want).abs().max().item()
    assert allshift.shape == want.shape and err < 2e-5, err
    # equal rows: ONE all_gather_into_tensor, rank-major
    eq = mdist.all_gather_rows(torch.full((3, 2), float(rank)), equal_rows=True)
    assert eq.shape == (3 * W, 2) and all(bool(eq[3 * r: 3 * r + 3].eq(float(r)).all()) for r in range(W))
    # a rank with NO rows (fewer meshes than ranks) still takes part in the ragged gather
    part = torch.full((0 if rank == 5 else rank + 1, 4), float(rank))
    rg = mdist.all_gather_rows(part)
    assert rg.shape[0] == sum(0 if r == 5 else r + 1 for r in range(W))
    off = 0
    for r in range(W):
        n = 0 if r == 5 else r + 1
        assert bool(rg[off: off + n].eq(float(r)).all())
        off += n
dist.barrier()                                   # nobody tears its sockets down while a peer is still inside a collective:
dist.destroy_process_group()                     # without this a rank's exit raced the others' and aborted now and then (-6 after 'ok')
print('rank', rank, 'ok')
"""


def test_eight_rank_gloo_ragged_and_equal_gathers():
    import socket
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    with tempfile.NamedTemporaryFile("w", suffix=".py", delete=False) as f:
        f.write(_WORKER8)
    try:
        env = dict(os.environ, OMP_NUM_THREADS="1")
        procs = [subprocess.Popen([sys.executable, f.name, ROOT, str(port), str(r)], stdout=subprocess.PIPE,
                                  stderr=subprocess.STDOUT, text=True, env=env) for r in range(8)]
        outs = [p.communicate(timeout=600)[0] for p in procs]
        for r, (p, o) in enumerate(zip(procs, outs)):
            assert p.returncode == 0 and f"rank {r} ok" in o, o[-3000:]
    finally:
        os.unlink(f.name)


def _check_n8_line(r, scaling, per_gpu, total, unit):
    assert r["n_gpus"] == 8 and r["rccl_ranks"] == 8 and r["backend"] == "gloo" and r["scaling"] == scaling
    assert r["config"]["meshes_per_gpu"] == per_gpu and r["config"]["global_batch"] == total and r["unit"] == unit
    assert len(r["per_rank_ms_per_step"]) == 8 and all(x > 0 for x in r["per_rank_ms_per_step"])
    assert abs(max(r["per_rank_ms_per_step"]) - r["ms_per_step"]) < 1e-6                  # max over ranks is what `value` uses
    assert abs(r["value"] - total * r["steps"] / (r["ms_per_step"] * r["steps"] * 1e-3)) / r["value"] < 0.02
    assert r["allgather_ms_per_step"] is not None and r["allgather_ms_per_step"] > 0 and r["allgather_calls_per_step"] >= 1
    # the keys an N = 8 line is judged on are all there (roofline / cpu_baseline carry their plumbing notes)
    assert isinstance(r["roofline"], dict) and r["roofline"]["bound"] in ("mfma", "hbm")
    cb = r["cpu_baseline"]
    assert isinstance(cb, dict) and cb["kind"] == "port" and cb["value"] > 0 and cb["cores"] >= 1 and cb["unit"] == unit
    assert "PLUMBING" in r["data"] and r["vs_baseline"] is None


def test_bench_eight_ranks_weak():
    r = _bench_line(["--gpus", "8", "--steps", "2", "--warmup", "1", "--batch", "2", "--cpu-seconds", "1"], dict(OMP_NUM_THREADS="1"))
    _check_n8_line(r, "weak", 2, 16, "meshes/s")


def test_bench_eight_ranks_strong():
    r = _bench_line(["--gpus", "8", "--steps", "2", "--warmup", "0", "--batch", "8", "--scaling", "strong", "--cpu-seconds", "1"],
                    dict(OMP_NUM_THREADS="1"))
    _check_n8_line(r, "strong", 1, 8, "meshes/s")


def test_bench_eight_ranks_corrnet_pairs():
    """BASELINE.json configs[3]: cloud pairs sharded over eight ranks, the three outputs gathered (bench.py's pair workloads)"""
    r = _bench_line(["--gpus", "8", "--steps", "2", "--warmup", "1", "--batch", "2", "--cpu-seconds", "1", "--workload", "corrnet"],
                    dict(OMP_NUM_THREADS="1"))
    _check_n8_line(r, "weak", 2, 16, "pairs/s")
    assert "configs[3]" in r["config"]["workload"] and r["allgather_calls_per_step"] == 3 and r["allgather_bytes_per_rank"] > 0


def test_bench_eight_ranks_ragged_batches():
    """the config-5 stand-in (`--workload jointnet_ragged`) at world size 8: every rank holds meshes of different sizes, so
    the outputs go through the count exchange + padded gather (no two ranks hold the same number of rows)"""
    r = _bench_line(["--gpus", "8", "--steps", "2", "--warmup", "1", "--batch", "2", "--cpu-seconds", "1", "--workload", "jointnet_ragged"],
                    dict(OMP_NUM_THREADS="1"))
    _check_n8_line(r, "weak", 2, 16, "meshes/s")
    assert "ragged" in r["config"]["workload"] and len(r["config"]["workload"]) < 120
    lo, hi = r["config"]["vertices_per_mesh"]
    assert 16 <= lo < hi <= 64 and r["vertices_per_s"] > 0                     # (plumbing meshes: 4..8-side grids)


# ---- RCCL on hardware: the one-GPU box can only hold a world of one rank, but every collective the multi-GPU paths issue
# ---- (process-group creation on the device, all-reduce, all-gather, barrier) goes through RCCL all the same
@pytest.mark.gpu
def test_rccl_path_on_one_gpu():
    """the driver's own launch line at --nproc-per-node 1, with MORIG_BENCH_FORCE_DIST=1 so that bench.py takes its N > 1
    branch: `init_process_group('nccl', device_id=...)`, the rank-count all-reduce, the RCCL all-gather of the outputs,
    the barriers around the timed region and the max-over-ranks reduction all run on the GPU."""
    import json, socket
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    env = dict(os.environ, MORIG_BENCH_FORCE_DIST="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MORIG_BENCH_PLUMBING"):
        env.pop(k, None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "3", "--warmup", "1",
           "--batch", "8", "--cpu-seconds", "0", "--secondary", "0", "--prof-steps", "0"]
    p = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout
    r = json.loads(lines[0])
    assert r["backend"] == "nccl" and r["rccl_ranks"] == 1 and r["n_gpus"] == 1
    assert "RCCL all-gather" in r["config"]["step"] and len(r["config"]["workload"]) < 120 and r["value"] > 0 and len(r["per_rank_ms_per_step"]) == 1


_RCCL_WORKER = r"""
import copy, os, sys, torch, torch.distributed as dist
sys.path.insert(0, sys.argv[1])
from morig_amd import dist as mdist, models, synth, train_backward as TB
dev = torch.device('cuda', 0)
torch.cuda.set_device(dev)
dist.init_process_group('nccl', init_method='tcp://127.0.0.1:%s' % sys.argv[2], rank=0, world_size=1, device_id=dev)
# ragged gather: count exchange + padded all_gather_into_tensor
t = torch.arange(35, dtype=torch.float32, device=dev).reshape(7, 5)
assert torch.equal(mdist.all_gather_rows(t, even_alone=True), t)
assert torch.equal(mdist.all_gather_rows(t, equal_rows=True, even_alone=True), t)
# one training step with the BatchNorm sums all-reduced over RCCL (fp64 buffers) == the same step without the hook
torch.manual_seed(7)
net = models.rignet.GCNRig(chn_feature=3, chn_output=8).train().to(dev)
b = synth.collate([synth.make_mesh(s, n_side=12, with_skin=False) for s in (1, 2, 3)]).to(dev)


def step(model, sync):
    TB.set_batchnorm_sync(True if sync else None)
    st = TB.graph_state(b)
    out = TB.gcnrig(model, b.pos.float(), b.pred_flow[:, :3].float(), st['csr_tpl'], st['csr_geo'], st['batch'], st['mesh_ptr'], st['ng'])
    w = torch.sin(b.pos[:, :1].float() * 50.0 + torch.arange(out.shape[1], dtype=torch.float32, device=dev))
    (out * w).sum().backward()
    return out.detach()


a, c = copy.deepcopy(net), copy.deepcopy(net)
with torch.enable_grad():
    oa, oc = step(a, True), step(c, False)
TB.set_batchnorm_sync(None)
assert float((oa - oc).abs().max()) <= 1e-5 * max(1.0, float(oc.abs().max()))
for (k, p), (_, q) in zip(a.named_parameters(), c.named_parameters()):
    assert p.grad is not None and torch.isfinite(p.grad).all(), k
    x, y = p.grad.double().flatten(), q.grad.double().flatten()
    assert float(torch.dot(x, y) / (x.norm() * y.norm() + 1e-300)) >= 0.9999, k
dist.barrier()
dist.destroy_process_group()
print('rccl ok')
"""


@pytest.mark.gpu
def test_rccl_ragged_gather_and_batchnorm_sums_on_one_gpu():
    import socket, tempfile
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    with tempfile.NamedTemporaryFile("w", suffix=".py", delete=False) as f:
        f.write(_RCCL_WORKER)
    try:
        env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
        p = subprocess.run([sys.executable, f.name, ROOT, str(port)], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT,
                           text=True, timeout=600)
        assert p.returncode == 0 and "rccl ok" in p.stdout, p.stdout[-3000:]
    finally:
        os.unlink(f.name)


@pytest.mark.gpu
def test_c_abi_rccl_all_gather_on_one_gpu():
    """the C-ABI's own RCCL wrapper (morig_rccl_unique_id / _comm_init / morig_allgather_rows) -- a communicator of
    one rank on the one GPU a box has: the collective runs through librccl on the device stream and returns the rows"""
    import torch
    from morig_amd import dist as mdist
    comm = mdist.RcclComm(1, 0, mdist.RcclComm.unique_id())
    try:
        t = torch.arange(5000 * 3, dtype=torch.float32, device="cuda").reshape(5000, 3)
        out = comm.all_gather_rows(t)
        torch.cuda.synchronize()
        assert out.shape == (5000, 3) and torch.equal(out, t)
    finally:
        comm.close()
