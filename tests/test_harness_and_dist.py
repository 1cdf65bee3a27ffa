"""CPU: caller-side harness (post-ops, writers) against fixtures captured from the reference, the
C-ABI export list against include/morig_hip.h, and the multi-rank path on gloo (world_size 2)."""
import os
import re
import subprocess
import sys
import tempfile

import numpy as np
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
    assert lib.morig_abi_version() == 1
    assert lib.morig_strerror(-2).decode().startswith("unsupported")


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


def _bench_line(extra, env_extra=None):
    import json
    env = dict(os.environ, MORIG_BENCH_PLUMBING="1", OMP_NUM_THREADS="2")
    env.pop("WORLD_SIZE", None); env.pop("RANK", None); env.pop("LOCAL_RANK", None)
    env.update(env_extra or {})
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + extra, env=env, stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout            # rank 0 prints ONE JSON line
    return json.loads(lines[0])


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
