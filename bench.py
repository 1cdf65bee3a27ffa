#!/usr/bin/env python
"""bench.py -- meshes/sec of the jointnet_motion eval-mode forward on synthetic 4 k-vertex meshes.

    python bench.py --gpus N --steps K --warmup W [--scaling weak|strong]

N > 1 from a bare shell re-executes itself under `python -m torch.distributed.run` (one rank per GPU, RCCL); when the
driver already launched it that way (WORLD_SIZE set) it just joins. One step = one forward of
`jointnet_motion(num_keyframes=5, chn_output=3, aggr_method='attn')` over a device-resident batch of 64 synthetic
4096-vertex meshes PER GPU (BASELINE.json configs[1]; weak scaling; `--scaling strong` splits ONE 64-mesh batch),
including the COO->CSR graph preparation and, for N > 1, the RCCL all-gather of pred_shift. Rank 0 prints ONE JSON line.

Timing: W warm-up steps, then K steps bracketed by barrier + synchronize (wall clock -> `value`, max over ranks) with one
HIP event per step boundary (median / p10 / p90); the per-kernel breakdown and the roofline come from a SECOND pass of a
few steps with HIP events around every launch (morig_prof_*), so the timed region itself carries no per-launch events.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import torch  # noqa: E402

PEAK_F32_MFMA_TFLOPS = 157.3          # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense
PEAK_F16_MFMA_TFLOPS = 2500.0         # MI355X_MICROARCH.md: BF16/FP16 MFMA, dense
PEAK_SCLK_MHZ = 2400.0                # the clock the peaks above are quoted at; what the chip SUSTAINS is sampled in the run (ClockSampler)
PEAK_HBM_GBS = 8000.0
# MORIG_BENCH_PLUMBING=1 (set by tests/ only): gloo + CPU tensors + the torch emulation of the op layer on tiny meshes.
# It exercises launch / sharding / all-gather / JSON assembly; its numbers are not measurements and the line says so.
PLUMBING = os.environ.get("MORIG_BENCH_PLUMBING") == "1"

def lib_sha256():
    """sha256 of the libmorig_hip.so this process loads: the counter passes under profiles/ carry the hash of the library they
    were collected on, and their numbers enter the line only when it is the same build."""
    import hashlib
    from morig_amd import native
    try:
        return hashlib.sha256(open(native.library_path(), "rb").read()).hexdigest()
    except Exception:
        return None


def _profile_json(name):
    """-> (parsed counter pass or None, reason it is not usable or None)"""
    path = os.path.join(ROOT, "profiles", name)
    try:
        prof = json.load(open(path))
    except Exception:
        return None, f"profiles/{name} not found"
    want, have = prof.get("lib_sha256"), lib_sha256()
    if want is None:
        return None, f"profiles/{name} carries no lib_sha256 stamp"
    if want != have:
        return None, f"profiles/{name} was collected on library {want[:12]}, this run loaded {str(have)[:12]}"
    return prof, None


def _kernels_of(prof, symbol):
    return [(n, v) for n, v in prof.get("kernels", {}).items() if symbol in n]


def measured_traffic(symbol):
    """HBM bytes per launch of a kernel symbol from the committed rocprofv3 PMC pass (profiles/traffic_latest.json, produced
    by tools/gpu_pmc_bench.sh on the same command and the same library build), corrected as MI355X_MICROARCH.md prescribes:
    (2 x FETCH_SIZE + WRITE_SIZE) KiB. -> (bytes or None, reason or None)"""
    prof, why = _profile_json("traffic_latest.json")
    if prof is None or symbol is None:
        return None, why or "no single kernel symbol"
    tot, n = 0.0, 0            # a symbol prefix may cover several instantiations: dispatch-weighted mean
    for name, v in _kernels_of(prof, symbol):
        if "FETCH_SIZE_KiB_per_dispatch" in v and "WRITE_SIZE_KiB_per_dispatch" in v:
            d = v.get("dispatches", 1)
            tot += (2.0 * v["FETCH_SIZE_KiB_per_dispatch"] + v["WRITE_SIZE_KiB_per_dispatch"]) * 1024.0 * d
            n += d
    return (round(tot / n), None) if n else (None, f"{symbol} not in profiles/traffic_latest.json")


def measured_mfma_util(symbol):
    """MFMA busy fraction of a kernel symbol from the committed counter pass (profiles/mfma_pmc_latest.json, produced by
    tools/gpu_pmc_mfma.sh: SQ_VALU_MFMA_BUSY_CYCLES / (4 SIMDs x SQ_BUSY_CU_CYCLES)). -> (fraction or None, reason or None)"""
    prof, why = _profile_json("mfma_pmc_latest.json")
    if prof is None or symbol is None:
        return None, why or "no single kernel symbol"
    busy = cu = 0.0
    for name, v in _kernels_of(prof, symbol):
        if v.get("SQ_BUSY_CU_CYCLES"):
            d = v.get("dispatches", 1)
            busy += v.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) * d
            cu += v["SQ_BUSY_CU_CYCLES"] * d
    return (round(busy / (4.0 * cu), 4), None) if cu else (None, f"{symbol} not in profiles/mfma_pmc_latest.json")


def measured_mfma_issued(symbol):
    """v_mfma instructions one launch of a kernel symbol ISSUES (SQ_INSTS_MFMA of the committed counter pass, per-dispatch average over the
    symbol's launches) -> (count or None, reason or None). Against 3 x the algorithmic products it prices what the launch spends on CSR
    padding and tile quantisation (VERDICT r5 #1c)."""
    prof, why = _profile_json("mfma_pmc_latest.json")
    if prof is None or symbol is None:
        return None, why or "no single kernel symbol"
    n = d_all = 0.0
    for name, v in _kernels_of(prof, symbol):
        if v.get("SQ_INSTS_MFMA"):
            d = v.get("dispatches", 1)
            n += v["SQ_INSTS_MFMA"] * d
            d_all += d
    return (n / d_all, None) if d_all else (None, f"{symbol} not in profiles/mfma_pmc_latest.json")


class ClockSampler:
    """Samples the GPU's shader clock and socket power in a background thread while a timed loop runs (VERDICT r3 #1c: the
    clock in the line is the clock of THIS run). Sources, first one that answers: the amdsmi Python binding, the amdgpu sysfs
    nodes (pp_dpm_sclk / hwmon power1_average), `rocm-smi --json`. Never raises: a box without any of them yields no samples
    and the line says so."""

    def __init__(self, index=0, period=0.05):
        import threading
        self.index, self.period = index, period
        self.samples = []                              # (t, sclk MHz or None, watts or None)
        self._stop = threading.Event()
        self._thread = None
        self.source = None
        self._read = self._pick_source()

    # -- sources -------------------------------------------------------------------------------------------------
    def _pick_source(self):
        for name, make in (("amdsmi", self._make_amdsmi), ("sysfs", self._make_sysfs), ("rocm-smi", self._make_rocm_smi)):
            try:
                fn = make()
                if fn is None:
                    continue
                s = fn()
                if s and (s[0] or s[1]):
                    self.source = name
                    return fn
            except Exception:
                continue
        return None

    def _make_amdsmi(self):
        import amdsmi
        amdsmi.amdsmi_init()
        hs = amdsmi.amdsmi_get_processor_handles()
        if not hs:
            return None
        h = hs[min(self.index, len(hs) - 1)]

        def num(v):
            try:
                return float(v)
            except Exception:
                return None

        def read():
            clk = pw = None
            try:
                ci = amdsmi.amdsmi_get_clock_info(h, amdsmi.AmdSmiClkType.GFX)
                clk = num(ci.get("clk", ci.get("cur_clk")))
            except Exception:
                pass
            try:
                pi = amdsmi.amdsmi_get_power_info(h)
                for k in ("current_socket_power", "average_socket_power", "socket_power"):
                    v = num(pi.get(k))
                    if v:
                        pw = v
                        break
            except Exception:
                pass
            return clk, pw
        return read

    def _make_sysfs(self):
        import glob
        cards = sorted(glob.glob("/sys/class/drm/card[0-9]*/device/pp_dpm_sclk"))
        if not cards:
            return None
        dpm = cards[min(self.index, len(cards) - 1)]
        hw = sorted(glob.glob(os.path.join(os.path.dirname(dpm), "hwmon/hwmon*/power1_average")) +
                    glob.glob(os.path.join(os.path.dirname(dpm), "hwmon/hwmon*/power1_input")))

        def read():
            clk = pw = None
            for line in open(dpm):
                if "*" in line:
                    clk = float(line.split(":")[1].lower().replace("mhz", "").replace("*", "").strip())
            if hw:
                pw = float(open(hw[0]).read().strip()) / 1e6
            return clk, pw
        return read

    def _make_rocm_smi(self):
        import shutil
        exe = shutil.which("rocm-smi") or "/opt/rocm/bin/rocm-smi"
        if not os.path.exists(exe):
            return None
        idx = self.index

        def read():
            out = subprocess.run([exe, "--showclocks", "--showpower", "--json"], capture_output=True, text=True, timeout=5).stdout
            card = json.loads(out).get(f"card{idx}", {})
            clk = pw = None
            for k, v in card.items():
                kl = k.lower()
                if "sclk" in kl and "level" in kl and "(" in str(v):
                    clk = float(str(v).split("(")[1].lower().replace("mhz)", "").strip())
                elif "power" in kl and "(w)" in kl:
                    try:
                        pw = float(v)
                    except Exception:
                        pass
            return clk, pw
        return read

    # -- control --------------------------------------------------------------------------------------------------
    def _run(self):
        while not self._stop.is_set():
            try:
                clk, pw = self._read()
                self.samples.append((time.perf_counter(), clk, pw))
            except Exception:
                pass
            self._stop.wait(self.period)

    def start(self):
        import threading
        self.samples = []
        self._stop.clear()
        if self._read is not None:
            self._thread = threading.Thread(target=self._run, daemon=True)
            self._thread.start()
        return self

    def stop(self):
        self._stop.set()
        if self._thread is not None:
            self._thread.join(timeout=6)
            self._thread = None
        return self

    def summary(self, t0=None, t1=None):
        """medians over the samples taken inside [t0, t1] (the timed region), skipping the first quarter of them: the power
        controller needs about a second to settle after the load starts"""
        ss = [s for s in self.samples if (t0 is None or s[0] >= t0) and (t1 is None or s[0] <= t1)]
        settled = ss[len(ss) // 4:] if len(ss) >= 8 else ss
        clk = [s[1] for s in settled if s[1]]
        pw = [s[2] for s in settled if s[2]]
        return dict(sclk_under_load_mhz=round(pct(clk, 0.5), 1) if clk else None,
                    sclk_p10_mhz=round(pct(clk, 0.1), 1) if clk else None, sclk_p90_mhz=round(pct(clk, 0.9), 1) if clk else None,
                    socket_power_w=round(pct(pw, 0.5), 1) if pw else None,
                    samples=len(ss), samples_used=len(settled), source=self.source,
                    period_s=self.period,
                    note="sampled in a background thread during the timed steps of THIS run (medians; first quarter of the samples dropped as ramp)"
                         if ss else "no clock source answered on this host (amdsmi / sysfs / rocm-smi)")


def _issued_fields(dom, per_product):
    """mfma_issued_over_algorithmic = v_mfma_f32_32x32x16 instructions the dominant kernel issues per launch (counter pass) / (per_product x
    algorithmic FLOPs per launch / 32 768 FLOPs per instruction): 1.0 = no padded rows, no tile quantisation"""
    issued, why = measured_mfma_issued(dom["symbol"])
    if not issued or not dom["launches"] or dom["flops"] <= 0:
        return dict(mfma_issued_per_launch=None, mfma_issued_over_algorithmic=None, mfma_issued_unavailable=why)
    alg = per_product * dom["flops"] / dom["launches"] / 32768.0
    return dict(mfma_issued_per_launch=round(issued), mfma_issued_over_algorithmic=round(issued / alg, 4))


def roofline_of(prof, psteps, clocks=None):
    """-> (roofline dict of the dominant kernel symbol, per-kind breakdown, total GPU ms) from one morig_prof_* pass.
    Launch kinds are grouped by the ONE kernel symbol they run (morig_prof_symbol), so the dominant entry is the object
    `rocprofv3 --kernel-trace --stats` ranks first and frac follows from profiles/ + this line alone. A dominant kind that
    declares FLOPs is priced against the dense MFMA peak of its arithmetic; one that declares only bytes against HBM."""
    from morig_amd import native
    total_ms = sum(v["ms"] for v in prof.values())
    by_sym = {}
    for k, v in prof.items():
        key = v.get("symbol") or ("kind:" + k)
        g = by_sym.setdefault(key, dict(ms=0.0, flops=0.0, bytes=0.0, launches=0, kinds=[], symbol=v.get("symbol")))
        g["ms"] += v["ms"]; g["flops"] += v["flops"]; g["bytes"] += v["bytes"]; g["launches"] += v["launches"]
        g["kinds"].append(k)
    dom_key, dom = max(by_sym.items(), key=lambda kv: kv[1]["ms"])
    chain = None
    if dom["flops"] <= 0 and dom["bytes"] <= 0:
        # the largest entry declares neither FLOPs nor bytes: a dependent-latency chain (farthest-point sampling: m arg-max steps
        # per cloud on ONE CU each, 32 of 256 CUs busy, on its own stream beside the other branch) -- no roof applies to it. It is
        # reported as such, and the roofline object describes the largest kernel that HAS a roof.
        chain = dict(kind=dom_key, kernel_is="launch kind (latency chain: neither the HBM nor the MFMA roof applies)",
                     ms_per_step=round(dom["ms"] / psteps, 3), launches_per_step=dom["launches"] / psteps,
                     share_of_gpu_time=round(dom["ms"] / total_ms, 4),
                     note="event time of a kernel that runs CONCURRENTLY with the other stream's kernels: its share of the summed "
                          "event times overstates its share of the step")
        rest = {k: v for k, v in by_sym.items() if v["flops"] > 0 or v["bytes"] > 0}
        if rest:
            dom_key, dom = max(rest.items(), key=lambda kv: kv[1]["ms"])
    sclk = (clocks or {}).get("sclk_under_load_mhz")
    common = dict(kernel=dom["symbol"] or dom_key, kernel_is="rocprofv3 kernel symbol (prefix)" if dom["symbol"] else "launch kind (several symbols)",
                  launch_kinds=sorted(dom["kinds"]), lib_sha256=lib_sha256(),
                  launches_per_step=dom["launches"] / psteps, avg_launch_ms=round(dom["ms"] / dom["launches"], 4),
                  share_of_gpu_time=round(dom["ms"] / total_ms, 4),
                  timing="HIP events around every launch in a separate pass of %d steps" % psteps,
                  sclk_under_load_mhz=sclk, socket_power_w=(clocks or {}).get("socket_power_w"),
                  sclk_source=("%s, %d samples inside the timed region of this run" % ((clocks or {}).get("source"), (clocks or {}).get("samples_used", 0)))
                              if sclk else "not sampled", peak_quoted_at_mhz=PEAK_SCLK_MHZ)
    alg_bytes = round(dom["bytes"] / dom["launches"]) if dom["launches"] else None
    traffic, traffic_why = measured_traffic(dom["symbol"])
    tr = dict(algorithmic_bytes=alg_bytes,
              algorithmic_bytes_unit="compulsory bytes per launch as the launcher declares them: every operand-table row and every result row ONCE "
                                     "(fp32 elements; the per-edge re-reads of gathered rows are L2 hits and are not counted)",
              traffic=traffic, traffic_unavailable=traffic_why,
              traffic_over_algorithmic=round(traffic / alg_bytes, 3) if traffic and alg_bytes else None,
              traffic_unit="HBM bytes per launch ((2 x FETCH_SIZE + WRITE_SIZE) KiB, rocprofv3 --pmc, profiles/traffic_latest.json)")
    if dom["flops"] > 0:
        achieved = dom["flops"] / (dom["ms"] * 1e-3) / 1e12 if dom["ms"] > 0 else 0.0
        # split-fp16 kernels issue 3 f16 MFMAs per algorithmic product: `achieved` stays ALGORITHMIC flops/s,
        # `peak` is the dense f16 MFMA peak, so frac <= 1/3 by construction (frac_of_3x_split_peak rescales)
        split = any("f16x3" in k for k in dom["kinds"]) or (native.get_ops().precision == "f16x3" and any(k in ("cosine_knn",) for k in dom["kinds"]))
        peak = PEAK_F16_MFMA_TFLOPS if split else PEAK_F32_MFMA_TFLOPS
        util, util_why = measured_mfma_util(dom["symbol"])
        roof = dict(bound="mfma", achieved=round(achieved, 2), peak=peak, unit="TFLOP/s", frac=round(achieved / peak, 4),
                    algorithmic_flops_per_launch=round(dom["flops"] / dom["launches"]) if dom["launches"] else None,
                    mfma_issued_per_product=3 if split else 1,
                    **_issued_fields(dom, 3 if split else 1),
                    frac_of_3x_split_peak=round(3 * achieved / peak, 4) if split else None,
                    frac_of_3x_split_peak_at_sclk=round(3 * achieved / peak * PEAK_SCLK_MHZ / sclk, 4) if (split and sclk) else None,
                    mfma_util_counter=util, mfma_util_counter_unavailable=util_why,
                    mfma_util_counter_source="profiles/mfma_pmc_latest.json (SQ_VALU_MFMA_BUSY_CYCLES / 4 SQ_BUSY_CU_CYCLES)")
    else:
        gbs = dom["bytes"] / (dom["ms"] * 1e-3) / 1e9 if dom["ms"] > 0 else 0.0
        roof = dict(bound="hbm", achieved=round(gbs, 1), peak=PEAK_HBM_GBS, unit="GB/s", frac=round(gbs / PEAK_HBM_GBS, 4),
                    note="the dominant launch kind declares no FLOPs: priced on the bytes its launcher declares against the HBM rate "
                         "(a latency chain such as farthest-point sampling sits far below either roof by construction)")
    roof.update(common)
    roof.update(tr)
    if chain is not None:
        roof["largest_entry_is_a_latency_chain"] = chain
    breakdown = {k: dict(ms_per_step=round(v["ms"] / psteps, 3),
                         tflops=round(v["flops"] / (v["ms"] * 1e-3) / 1e12, 2) if v["ms"] > 0 else 0.0,
                         launches_per_step=v["launches"] / psteps, symbol=v.get("symbol"))
                 for k, v in sorted(prof.items(), key=lambda kv: -kv[1]["ms"])}
    return roof, breakdown, total_ms


def mfma_power_probe(seconds=1.5, index=0):
    """The matrix pipes' rate under the socket POWER CAP (morig_ubench_mfma: every wave issues v_mfma_f32_32x32x16_f16 from registers,
    random fp16 operands): the data sheet's 2.5 PFLOP/s assumes 2.4 GHz, which the cap does not grant once operands toggle. Returns the
    dense f16 TFLOP/s this GPU sustains, with the clock and power sampled during the probe (peaks "verify on the box")."""
    import ctypes
    import torch
    from morig_amd import native
    lib = native.load_library()
    scratch = torch.zeros(16, device="cuda")
    fl = ctypes.c_double(0.0)
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    iters = 4000
    native.check(lib.morig_ubench_mfma(0, iters, 2, scratch.data_ptr(), ctypes.byref(fl), st), "morig_ubench_mfma")
    torch.cuda.synchronize()
    cs = ClockSampler(index=index, period=0.02).start()
    t0 = time.perf_counter()
    total = 0.0
    while time.perf_counter() - t0 < seconds:
        native.check(lib.morig_ubench_mfma(0, iters, 8, scratch.data_ptr(), ctypes.byref(fl), st), "morig_ubench_mfma")
        torch.cuda.synchronize()
        total += fl.value
    dt = time.perf_counter() - t0
    cs.stop()
    sm = cs.summary()
    return dict(tflops=round(total / dt / 1e12, 1), sclk_mhz=sm.get("sclk_under_load_mhz"), socket_power_w=sm.get("socket_power_w"),
                seconds=round(dt, 2), operands="random fp16", kernel="ubench_mfma_kernel<0>",
                note="every wave issues v_mfma_f32_32x32x16_f16 back to back from registers: what the socket power cap leaves of the 2.5 PFLOP/s")


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except Exception:
        pass
    return "unknown"


def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def self_launch(n):
    """bare `python bench.py --gpus N`: become the launcher of N ranks (the driver's own command line, verbatim)."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(free_port()), os.environ.get("MORIG_BENCH_ENTRY") or os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "8")
    return subprocess.call(cmd, env=env)


def build_batch(seeds, n_side, with_skin=False, n_pts=0, dev=None):
    """the synthetic batch of SURVEY 8(d). On the GPU its geodesic-ball graph is built on the device for all meshes at once
    (morig_geo_ball_graph, the reference's get_geo_edges: data_proc/common_ops.py:214-226); positions, the 1-ring graph and the
    flows are host numpy (no O(V^2) work). PLUMBING (CPU emulation, tiny meshes): the host recipe."""
    from morig_amd import synth
    if dev is None or dev.type != "cuda":
        b = synth.make_batch(seeds, n_side=n_side, n_pts=n_pts, with_skin=with_skin)
    else:
        b = synth.make_batch_device(seeds, dev, n_side=n_side, n_pts=n_pts, with_skin=with_skin, geo_seed=seeds[0])
    b.num_graphs = len(seeds)
    return b


RAGGED_SIDES = (32, 80)               # grid sides of the ragged batch: 1 024 ... 6 400 vertices per mesh
GEO_RADIUS = 0.06                     # data_proc/common_ops.py:214 get_geo_edges(radius=0.06): ONE radius for every mesh size


def ragged_sides(seeds, lo=RAGGED_SIDES[0], hi=RAGGED_SIDES[1]):
    """grid side of every mesh of the ragged batch: a pure function of its seed (uniform over [lo, hi])"""
    import numpy as np
    return [int(np.random.default_rng([0x52414747, int(sd)]).integers(lo, hi + 1)) for sd in seeds]


def build_batch_ragged(seeds, sides, dev=None):
    """The config-5 stand-in (BASELINE.json configs[4] is blocked offline): meshes of DIFFERENT sizes in one batch, as the rig datasets
    hold them (datasets/dataset_rig.py:85-138: ~1-5 k vertices per character), same recipe as build_batch per mesh, the reference's
    fixed ball radius 0.06 for all of them (small meshes get fewer than 15 ball members, large ones the random subset of 15)."""
    from morig_amd import synth
    if dev is None or dev.type != "cuda":
        b = synth.collate([synth.make_mesh(sd, n_side=ns, geo_radius=GEO_RADIUS, with_skin=False) for sd, ns in zip(seeds, sides)])
    else:
        from morig_amd import graph_build
        b = synth.collate([synth.make_mesh(sd, n_side=ns, geo="none", with_skin=False) for sd, ns in zip(seeds, sides)]).to(dev)
        b.geo_edge_index = graph_build.get_geo_edges(b.pos, b.batch, GEO_RADIUS, 15, seed=int(seeds[0]), self_loops=True,
                                                     num_graphs=len(seeds))
    b.num_graphs = len(seeds)
    return b


def cpu_baseline(seconds, n_side, rank0_batch_seed):
    """the CPU oracle (oracle/nets.py, 'port') timed on this host's cores on a bounded sample."""
    from morig_amd import synth
    from oracle import nets
    t_begin = time.perf_counter()
    cores = os.cpu_count() or 1
    try:
        import psutil
        cores = psutil.cpu_count(logical=False) or cores
    except Exception:
        pass
    kw = dict(num_keyframes=5, chn_output=3, aggr_method="attn")
    m = synth.load_recipe(nets.jointnet_motion(**kw).eval(), 0, mild=True)
    n_mesh = 2
    batch = synth.collate([synth.make_mesh(rank0_batch_seed + i, n_side=n_side, with_skin=False) for i in range(n_mesh)])
    with torch.no_grad():
        # the per-edge gathers do not scale to every core of a 2-socket host: probe a few thread counts
        # (one forward each) and time the best one; `cores` reports the count actually used
        best = None
        for th in sorted({min(cores, c) for c in (16, 32, 64, cores)}):
            torch.set_num_threads(th)
            m(batch, batch.pred_flow)
            t0 = time.perf_counter()
            m(batch, batch.pred_flow)
            d = time.perf_counter() - t0
            if best is None or d < best[1]:
                best = (th, d)
            if time.perf_counter() - t_begin > 0.6 * seconds:
                break
        cores = best[0]
        torch.set_num_threads(cores)
        t0 = time.perf_counter()
        reps = 0
        while True:
            m(batch, batch.pred_flow)
            reps += 1
            if time.perf_counter() - t0 >= 0.4 * seconds or reps >= 50:
                break
        dt = time.perf_counter() - t0
        # n = 1 thread: ONE forward of ONE mesh, only if the budget allows (~20 s on a current core)
        one = None
        if seconds >= 25:
            torch.set_num_threads(1)
            single = synth.collate([synth.make_mesh(rank0_batch_seed, n_side=n_side, with_skin=False)])
            t1 = time.perf_counter()
            m(single, single.pred_flow)
            d1 = time.perf_counter() - t1
            one = dict(value=round(1.0 / d1, 4), unit="meshes/s", sample=f"1 forward of 1 mesh, 1 thread, {d1:.1f} s")
            torch.set_num_threads(cores)
    return dict(value=round(n_mesh * reps / dt, 4), unit="meshes/s", cores=cores, kind="port", cpu_model=cpu_model(),
                host_cores=os.cpu_count(), threads_1=one,
                sample=f"{reps} forwards of a {n_mesh}-mesh batch ({n_side * n_side} vertices each), torch {torch.__version__} CPU, "
                       f"{cores} threads (best of a thread-count probe), {dt:.1f} s")


def cpu_baseline_other(workload, seconds, n_side, n_pts, seed):
    """the CPU oracle of the secondary workloads on ONE unit of the same size (one mesh / one mesh-cloud pair): whole
    forwards until the budget is used, at least one."""
    from morig_amd import synth
    from oracle import nets
    cores = min(os.cpu_count() or 1, 32)
    torch.set_num_threads(cores)
    with_pts = workload in ("corrnet", "deformnet")
    mesh = synth.make_mesh(seed, n_side=n_side, with_skin=workload == "mask_skin")
    clouds = [synth.make_point_cloud(mesh, int(mesh.name), n_pts)] if with_pts else None
    batch = synth.collate([mesh], clouds)
    if workload == "mask_skin":
        a = synth.load_recipe(nets.masknet_motion(num_keyframes=5, chn_output=1, aggr_method="attn").eval(), 0, mild=True)
        b = synth.load_recipe(nets.skinnet_motion(nearest_bone=5, use_Dg=False, use_Lf=False, num_keyframes=5, use_motion=True,
                                                  motion_dim=32).eval(), 1, mild=True)
        run = lambda: (a(batch, batch.pred_flow), b(batch, batch.pred_flow))
    elif workload == "corrnet":
        m = synth.load_recipe(nets.corrnet(input_feature=3, output_feature=64, temprature=0.07).eval(), 0, mild=True)
        run = lambda: m(batch, True, False)
    else:
        m = synth.load_recipe(nets.deformnet(tau_nce=0.07, num_interp=5).eval(), 0, mild=True)
        run = lambda: m(batch)
    with torch.no_grad():
        t0 = time.perf_counter()
        reps = 0
        while True:
            run()
            reps += 1
            if time.perf_counter() - t0 >= 0.8 * seconds or reps >= 20:
                break
        dt = time.perf_counter() - t0
    unit = "pairs/s" if with_pts else "meshes/s"
    what = f"one {n_side * n_side}-vertex mesh" + (f" + {n_pts}-point cloud" if with_pts else "")
    return dict(value=round(reps / dt, 4), unit=unit, cores=cores, kind="port", cpu_model=cpu_model(),
                sample=f"{reps} forward(s) of {what}, torch {torch.__version__} CPU, {cores} threads, {dt:.1f} s")


NAMES = {"jointnet": ("meshes/sec jointnet_motion forward, 4 k-vert synthetic",
                      "jointnet_motion(num_keyframes=5, attn) eval forward", "BASELINE.json configs[1]"),
         "jointnet_ragged": ("meshes/sec jointnet_motion forward, ragged synthetic batch (1 024 ... 6 400 vertices per mesh)",
                             "jointnet_motion(num_keyframes=5, attn) eval forward", "configs[4] stand-in"),
         "mask_skin": ("meshes/sec masknet_motion + skinnet_motion forward, 4 k-vert synthetic",
                       "masknet_motion + skinnet_motion(5 bones) eval fwds", "configs[2]"),
         "corrnet": ("pairs/sec corrnet forward, 4 k-vert mesh + 8 k-point cloud",
                     "corrnet(vismask, fixed FPS start) eval forward", "configs[3]"),
         "deformnet": ("pairs/sec deformnet forward, 4 k-vert mesh + 8 k-point cloud",
                       "deformnet(tau 0.07, 5 interp) eval fwd", "SURVEY 8(f-1), configs[3] pairs")}


def make_step(workload, data, dev, gather, model_only=False):
    """-> step() running ONE forward of `workload` over the resident batch (+ the all-gather of its outputs);
    model_only (jointnet): -> step(data) for any batch"""
    from morig_amd import models, synth
    if workload in ("jointnet", "jointnet_ragged"):
        model = models.jointnet_motion(num_keyframes=5, chn_output=3, aggr_method="attn").eval()
        synth.load_recipe(model, 0, mild=True).to(dev)
        if model_only:
            fn = lambda d: model(d, d.pred_flow)[2]
            fn.model = model
            return fn
        if PLUMBING or os.environ.get("MORIG_BENCH_GUARD", "deferred") == "sync":       # (the CPU plumbing run has no device-side guard)
            def step():
                motion_all, motion_aggr, pred_shift = model(data, data.pred_flow)
                return gather(pred_shift)
            return step
        # serving loop with the guard read deferred by one forward (NativeModule.forward_async): forward k + 1 is enqueued before
        # the host reads forward k's range flag / CSR status words, so the GPU does not idle on that read. EVERY forward's guard
        # is still read inside the timed region (the last one by drain(), in front of the closing fence).
        pending = []

        def check(p):
            if p is not None and not p.result():
                raise RuntimeError("bench input left the split-fp16 range: the timed forwards would have to be re-run in fp32")

        def step():
            (motion_all, motion_aggr, pred_shift), pend = model.forward_async(data, data.pred_flow)
            pending.append(pend)
            if len(pending) > 1:
                check(pending.pop(0))
            return gather(pred_shift)

        def drain():
            while pending:
                check(pending.pop(0))
        step.drain = drain
    elif workload == "mask_skin":
        model = models.masknet_motion(num_keyframes=5, chn_output=1, aggr_method="attn").eval()
        skin = models.skinnet_motion(nearest_bone=5, use_Dg=False, use_Lf=False, num_keyframes=5, use_motion=True,
                                     motion_dim=32).eval()
        synth.load_recipe(model, 0, mild=True).to(dev)
        synth.load_recipe(skin, 1, mild=True).to(dev)

        def step():
            mask = model(data, data.pred_flow)[2]
            sk = skin(data, data.pred_flow)[2]
            gather(mask)
            return gather(sk)
    elif workload == "deformnet":
        model = models.deformnet(tau_nce=0.07, num_interp=5).eval()
        synth.load_recipe(model, 0, mild=True).to(dev)

        def step():
            pred_flow, _, _, vis, _ = model(data)
            gather(vis)
            return gather(pred_flow)
    else:
        model = models.corrnet(input_feature=3, output_feature=64, temprature=0.07).eval()
        synth.load_recipe(model, 0, mild=True).to(dev)

        def step():
            out_vtx, out_pts, vis, _ = model(data, True, False)
            gather(out_pts)
            gather(vis)
            return gather(out_vtx)
    return step


def pct(xs, q):
    xs = sorted(xs)
    if not xs:
        return None
    i = (len(xs) - 1) * q
    lo, hi = int(i), min(int(i) + 1, len(xs) - 1)
    return xs[lo] + (xs[hi] - xs[lo]) * (i - lo)


LINE_LIMIT = 4096                     # the driver keeps ~8 KB of stdout tail: the LAST line must fit well inside it
ROOF_KEYS = ("bound", "kernel", "achieved", "peak", "unit", "frac", "algorithmic_flops_per_launch", "algorithmic_bytes", "traffic",
             "traffic_over_algorithmic", "mfma_util_counter", "mfma_issued_per_product", "mfma_issued_over_algorithmic", "frac_of_3x_split_peak", "avg_launch_ms",
             "launches_per_step", "share_of_gpu_time", "sclk_under_load_mhz", "socket_power_w", "mfma_power_capped_tflops",
             "frac_of_power_capped_peak", "lib_sha256")


def compact_roofline(roof):
    """numbers and identifiers only (no prose): what the judge recomputes frac from"""
    if not isinstance(roof, dict):
        return roof
    out = {k: roof.get(k) for k in ROOF_KEYS if k in roof}
    out.setdefault("traffic", None)
    return out


def compact_cpu(cb):
    if not isinstance(cb, dict):
        return cb
    out = {k: cb.get(k) for k in ("value", "unit", "cores", "kind", "cpu_model", "host_cores", "error") if k in cb}
    if isinstance(cb.get("threads_1"), dict):
        out["threads_1_value"] = cb["threads_1"].get("value")
    if "sample" in cb:
        out["sample"] = str(cb["sample"])[:160]
    return out


def compact_secondary(sec):
    """ONE number per secondary workload (+ its roofline fraction / kernel and its CPU number where it has them)"""
    if not isinstance(sec, dict):
        return sec
    out = {}
    for k, v in sec.items():
        if not isinstance(v, dict):
            continue
        if "error" in v:
            out[k] = dict(error=str(v["error"])[:80])
            continue
        if k == "small_batch":
            out[k] = {b: v[b].get("meshes_per_s") for b in ("B1", "B2", "B4", "B8") if isinstance(v.get(b), dict)}
            out[k].update({b + "_served": v[b].get("hipgraph_meshes_per_s") for b in ("B1", "B8") if isinstance(v.get(b), dict)})
            out[k]["B1_served_ms"] = (v.get("B1") or {}).get("hipgraph_ms_per_forward")
            out[k]["B8_over_B64"] = v.get("per_mesh_throughput_B8_over_B64")
            continue
        e = dict(value=v.get("value"), unit=v.get("unit"))
        if "ms_per_step" in v:
            e["ms_per_step"] = v["ms_per_step"]
        if "vertices_per_s" in v:                      # the ragged batch: vertices/s beside the uniform batch's
            e.update(vertices_per_s=v["vertices_per_s"], over_uniform=v.get("vertices_per_s_over_uniform"))
        r = v.get("roofline")
        if isinstance(r, dict):
            e.update(frac=r.get("frac"), bound=r.get("bound"), kernel=r.get("kernel"))
        c = v.get("cpu_baseline")
        if isinstance(c, dict):
            e["cpu"] = c.get("value")
            e["cpu_cores"] = c.get("cores")
        out[k] = e
    return out


def emit(res, detail_path=None):
    """stdout protocol. The per-workload detail (kernel tables, notes, small-batch sweep ...) goes out FIRST, one
    '[bench-detail] <json>' line per block (none starts with '{'), and into a side file; the LAST line is the ONE compact JSON
    object the driver parses: < LINE_LIMIT bytes, numbers and identifiers only."""
    full = dict(res)
    sec = full.get("secondary") or {}
    blocks = [("headline", {k: v for k, v in full.items() if k != "secondary"})]
    blocks += [("secondary." + k, v) for k, v in sec.items()]
    for name, blk in blocks:
        parts = [(name, blk)]
        if isinstance(blk, dict) and len(json.dumps(blk)) > 3500:       # keep every detail line short too: tables go out on their own
            rest = {k: v for k, v in blk.items() if k not in ("kernels", "roofline")}
            parts = [(name, rest)] + [(name + "." + k, blk[k]) for k in ("roofline", "kernels") if blk.get(k) is not None]
        for pname, part in parts:
            print("[bench-detail] " + json.dumps({"block": pname, "detail": part}, separators=(",", ":")), flush=True)
    path = detail_path or os.environ.get("MORIG_BENCH_DETAIL") or os.path.join(ROOT, "gpurun_out", "bench_detail.json")
    try:
        os.makedirs(os.path.dirname(path), exist_ok=True)
        with open(path, "w") as f:
            json.dump(full, f, indent=1)
    except Exception:
        path = None
    line = {k: full.get(k) for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "ms_per_step_median",
                                      "ms_per_step_p10", "ms_per_step_p90", "higher_is_better", "scaling", "vs_baseline", "dtype",
                                      "data", "config", "vertices_per_s", "rccl_ranks", "backend", "per_rank_ms_per_step", "allgather_ms_per_step",
                                      "allgather_calls_per_step", "allgather_bytes_per_rank")}
    line["roofline"] = compact_roofline(full.get("roofline"))
    line["whole_forward_tflops"] = full.get("whole_forward_tflops")
    line["hbm_bound_kernels"] = {k: v.get("hbm_frac") for k, v in (full.get("hbm_bound_kernels") or {}).items()}
    line["cpu_baseline"] = compact_cpu(full.get("cpu_baseline"))
    line["secondary"] = compact_secondary(full.get("secondary"))
    line["detail"] = "'[bench-detail]' stdout lines above" + (" + " + os.path.relpath(path, ROOT) if path else "")
    text = json.dumps(line, separators=(",", ":"))
    for drop in ("hbm_bound_kernels", "per_rank_ms_per_step", "detail", "secondary"):   # never reached at the default workloads
        if len(text) < LINE_LIMIT:
            break
        line.pop(drop, None)
        text = json.dumps(line, separators=(",", ":"))
    assert len(text) < LINE_LIMIT, len(text)
    print(text, flush=True)
    return line


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50, help="timed steps (SURVEY 8(d): >= 50)")
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--batch", type=int, default=64, help="meshes per GPU (weak) / in total (strong)")
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"])
    ap.add_argument("--n-side", type=int, default=64, help="mesh grid side (64 -> 4096 vertices)")
    ap.add_argument("--workload", default="jointnet", choices=["jointnet", "jointnet_ragged", "mask_skin", "corrnet", "deformnet"],
                    help="jointnet = BASELINE.json configs[1] (the headline metric); mask_skin = configs[2]; "
                         "corrnet = configs[3] (8192-point clouds, 32 pairs per GPU); deformnet = the producer of pred_flow "
                         "(SURVEY 8 f-1), same pairs as corrnet")
    ap.add_argument("--n-pts", type=int, default=8192)
    ap.add_argument("--power-probe-seconds", type=float, default=1.5,
                    help="length of the matrix-pipe power-cap probe behind roofline.mfma_power_capped_tflops (0 = skip)")
    ap.add_argument("--cpu-seconds", type=float, default=30.0, help="budget of the cpu_baseline leg (0 = skip)")
    ap.add_argument("--secondary-cpu-seconds", type=float, default=10.0,
                    help="budget of the cpu_baseline leg of each of the mask_skin / corrnet secondary entries (0 = skip)")
    ap.add_argument("--prof-steps", type=int, default=5, help="steps of the second, per-launch-evented pass")
    ap.add_argument("--secondary", type=int, default=-1,
                    help="steps of the short mask_skin / corrnet / deformnet runs added to the jointnet line at N = 1 "
                         "(-1 = 3; 0 switches them off: profiler passes and A/B scripts do)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args.gpus))
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world} (launch with --nproc-per-node {args.gpus})")

    B = args.batch
    pairs = args.workload in ("corrnet", "deformnet")
    if pairs and args.batch == 64:
        B = 32                                         # configs[3]: 256 pairs over 8 GPUs
    if args.scaling == "strong":
        if B % world:
            raise SystemExit(f"bench.py: --scaling strong needs --batch divisible by --gpus ({B} % {world})")
        B_local = B // world
        seeds = [1000 + i for i in range(B) if i % world == rank]            # morig_amd.dist.shard_items: round robin
    else:
        B_local = B
        seeds = [1000 + rank * B + i for i in range(B)]
    if PLUMBING:
        args.n_side, args.n_pts = min(args.n_side, 8), min(args.n_pts, 128)
    with_skin = args.workload == "mask_skin"
    import torch.distributed as dist
    if PLUMBING:
        from morig_amd import runtime
        # tests/bench_plumbing.py (the only entry that sets MORIG_BENCH_PLUMBING) installed its CPU emulation of the op layer
        assert runtime._test_ops is not None, "MORIG_BENCH_PLUMBING is for tests/bench_plumbing.py only"
        dev = torch.device("cpu")
        backend = "gloo"
    else:
        torch.cuda.set_device(local)
        dev = torch.device("cuda", local)
        backend = "nccl"                               # RCCL on ROCm
    rccl_ranks = 1
    # MORIG_BENCH_FORCE_DIST=1: take the collective path at world size 1 too (the 1-GPU box is the only hardware the
    # RCCL calls can be exercised on from this side: tests/test_harness_and_dist.py::test_rccl_path_on_one_gpu)
    use_dist = world > 1 or os.environ.get("MORIG_BENCH_FORCE_DIST") == "1"
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        if PLUMBING:
            dist.init_process_group(backend)
        else:
            dist.init_process_group(backend, device_id=dev)
        one = torch.ones(1, dtype=torch.int32, device=dev)
        dist.all_reduce(one)                           # a real collective over the backend: every rank must answer
        rccl_ranks = int(one.item())

    from morig_amd import dist as mdist, native

    ragged = args.workload == "jointnet_ragged"
    if ragged:
        sides = ragged_sides(seeds, *((4, 8) if PLUMBING else RAGGED_SIDES))
        data = build_batch_ragged(seeds, sides, dev=dev).to(dev)
    else:
        data = build_batch(seeds, args.n_side, with_skin=with_skin, n_pts=args.n_pts if pairs else 0, dev=dev).to(dev)
    n_vert = data.pos.shape[0]
    n_vert_all = n_vert * world
    if ragged and use_dist:
        nv_t = torch.tensor([n_vert], dtype=torch.int64, device=dev)
        dist.all_reduce(nv_t)
        n_vert_all = int(nv_t.item())
    gathered = []                                      # the tensors one step hands to the collective (refreshed every step)

    def gather(t):
        if not use_dist:
            return t
        if gather.record:
            gathered.append(t)
        # ragged batches: every rank holds a different number of rows -- the count-exchange + padded gather path
        return mdist.all_gather_rows(t, equal_rows=not ragged, even_alone=True)
    gather.record = False
    step = make_step(args.workload, data, dev, gather)

    def sync():
        if not PLUMBING:
            torch.cuda.synchronize()

    def fence():
        sync()
        if use_dist:
            dist.barrier()
            sync()

    def timed_run(step, steps, warmup, sampler=None):
        """-> (wall seconds for `steps` steps, per-step ms from HIP events, last output); `sampler` (ClockSampler) runs from the
        first warm-up step on and is summarised over the timed region only"""
        if sampler is not None:
            sampler.start()
        for _ in range(warmup):
            out = step()
        if hasattr(step, "drain"):
            step.drain()
        ev = None if PLUMBING else [torch.cuda.Event(enable_timing=True) for _ in range(steps + 1)]
        fence()
        t0 = time.perf_counter()
        if ev:
            ev[0].record()
        marks = [t0]
        for i in range(steps):
            out = step()
            if ev:
                ev[i + 1].record()
            else:
                marks.append(time.perf_counter())
        if hasattr(step, "drain"):
            step.drain()                                # the deferred guard read of the last forward(s) belongs to the timed region
        fence()
        t1 = time.perf_counter()
        dt = t1 - t0
        if sampler is not None:
            sampler.stop()
            sampler.window = (t0, t1)
        per = ([ev[i].elapsed_time(ev[i + 1]) for i in range(steps)] if ev
               else [(marks[i + 1] - marks[i]) * 1e3 for i in range(steps)])
        return dt, per, out

    def prof_pass(step, n):
        """second pass: HIP events around every launch, on the launch stream (outside the timed region)"""
        native.prof_reset()
        native.prof_enable(True)
        for _ in range(n):
            step()
        if hasattr(step, "drain"):
            step.drain()
        fence()
        native.prof_enable(False)
        return native.prof_collect()

    prof = {}
    with torch.no_grad():
        if not PLUMBING:
            ops = native.get_ops()
            ops.learn_edge_counts = True              # accounting only: exact E' per graph (one host read each),
            ops.csr_build(data.tpl_edge_index, n_vert)    # learned outside the steps so the FLOP counters use
            ops.csr_build(data.geo_edge_index, n_vert)    # algorithmic edges
            ops.learn_edge_counts = False
        sampler = None if PLUMBING else ClockSampler(index=local)
        dt, per_step, out = timed_run(step, args.steps, args.warmup, sampler)
        clocks = sampler.summary(*sampler.window) if sampler is not None else None
        if not PLUMBING and args.prof_steps > 0:
            prof = prof_pass(step, args.prof_steps)

    allgather_ms = None
    if use_dist:
        # the collective on its own: one more step records the tensors it hands to all_gather_rows, then the same
        # calls are repeated back to back between fences -- what the all-gather of one step costs when nothing hides it
        with torch.no_grad():
            gather.record = True
            step()
            if hasattr(step, "drain"):
                step.drain()
            gather.record = False
            reps = 20
            fence()
            tg = time.perf_counter()
            for _ in range(reps):
                for g_t in gathered:
                    mdist.all_gather_rows(g_t, equal_rows=not ragged, even_alone=True)
            fence()
            allgather_ms = (time.perf_counter() - tg) / reps * 1e3
        tg_t = torch.tensor([allgather_ms], dtype=torch.float64, device=dev)
        dist.all_reduce(tg_t, op=dist.ReduceOp.MAX)
        allgather_ms = float(tg_t.item())

    t = torch.tensor([dt], dtype=torch.float64, device=dev)
    per_rank = [dt]
    if use_dist:
        allt = [torch.zeros_like(t) for _ in range(world)]
        dist.all_gather(allt, t)
        per_rank = [float(x.item()) for x in allt]
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dt = float(t.item())
    assert out.shape[0] == n_vert_all and bool(torch.isfinite(out).all())

    secondary = None
    n_secondary = args.secondary if args.secondary >= 0 else (3 if args.workload == "jointnet" else 0)
    if rank == 0 and world == 1 and not PLUMBING and n_secondary > 0 and args.workload == "jointnet":
        # configs[2] / configs[3] (and the deformnet row of SURVEY 8(f)) as short driver-visible runs on the same GPU
        secondary = {}
        del step
        for wl in ("mask_skin", "corrnet", "deformnet"):
            wpairs = wl in ("corrnet", "deformnet")
            nb = 32 if wpairs else args.batch
            d2 = build_batch([1000 + i for i in range(nb)], args.n_side, with_skin=wl == "mask_skin", n_pts=args.n_pts if wpairs else 0, dev=dev)
            ns, nw = (4 * n_secondary, 3) if wpairs else (n_secondary, 1)        # the pair workloads are 12-18 ms steps: more of them
            with torch.no_grad():
                st = make_step(wl, d2, dev, lambda x: x)
                ssampler = ClockSampler(index=local)
                sdt, sper, _ = timed_run(st, ns, nw, ssampler)
                sroof = skern = None
                if args.prof_steps > 0:
                    # the same per-launch-evented pass as the headline workload: dominant rocprofv3 symbol, frac, bytes / FLOPs per launch
                    sp_steps = 4 if wpairs else 2
                    sroof, skern, _ = roofline_of(prof_pass(st, sp_steps), sp_steps, ssampler.summary(*ssampler.window))
            secondary[wl] = dict(metric=NAMES[wl][0], value=round(nb * ns / sdt, 2), unit="pairs/s" if wpairs else "meshes/s",
                                 ms_per_step=round(sdt / ns * 1e3, 3), steps=ns, warmup=nw, batch=nb,
                                 config=NAMES[wl][2], roofline=sroof, kernels=skern)
            del st, d2
            torch.cuda.empty_cache()
            if args.cpu_seconds > 0 and args.secondary_cpu_seconds > 0 and wl != "deformnet":
                # BASELINE configs[2] / configs[3] next to the CPU oracle on this host, like the headline line (VERDICT r3 #1a)
                try:
                    secondary[wl]["cpu_baseline"] = cpu_baseline_other(wl, args.secondary_cpu_seconds, args.n_side, args.n_pts, 1000)
                except Exception as e:
                    secondary[wl]["cpu_baseline"] = dict(error=repr(e)[:300])
        # config 5's stand-in: the same forward over a RAGGED batch -- 64 meshes of 1 024 ... 6 400 vertices -- in meshes/s AND
        # vertices/s beside the uniform headline batch (the tile-run / XCD partition of the EdgeConv kernels is otherwise only timed on
        # equal-size meshes)
        try:
            with torch.no_grad():
                rseeds = [1000 + i for i in range(args.batch)]
                rsides = ragged_sides(rseeds)
                d2 = build_batch_ragged(rseeds, rsides, dev=dev)
                st = make_step("jointnet", d2, dev, lambda x: x)
                rsteps = max(4 * n_secondary, 8)
                sdt, sper, _ = timed_run(st, rsteps, 3)
                nv2 = int(d2.pos.shape[0])
                uni_vps = n_vert * args.steps / dt
                secondary["jointnet_ragged"] = dict(
                    metric=NAMES["jointnet_ragged"][0], value=round(args.batch * rsteps / sdt, 2), unit="meshes/s",
                    ms_per_step=round(sdt / rsteps * 1e3, 3), steps=rsteps, warmup=3, batch=args.batch, vertices=nv2,
                    vertices_per_mesh_min_max=[min(rsides) ** 2, max(rsides) ** 2],
                    edges=dict(tpl=int(d2.tpl_edge_index.shape[1]), geo=int(d2.geo_edge_index.shape[1])),
                    vertices_per_s=round(nv2 * rsteps / sdt, 1), vertices_per_s_uniform_batch=round(uni_vps, 1),
                    vertices_per_s_over_uniform=round(nv2 * rsteps / sdt / uni_vps, 4), config=NAMES["jointnet_ragged"][2],
                    note="meshes of 32..80-side grids (seeded), the reference's fixed ball radius 0.06: small meshes have fewer geo edges "
                         "per vertex than the uniform batch, large ones the same 15; vertices/s is the comparable rate")
                del st, d2
                torch.cuda.empty_cache()
        except Exception as e:
            secondary["jointnet_ragged"] = dict(error=repr(e)[:300])
        # north_star's one-mesh-per-GPU operating point (and what `--scaling strong --gpus 8` gives each rank: 8 meshes): the same
        # forward at B = 1, 2, 4, 8 meshes per launch set
        try:
            sb = {}
            with torch.no_grad():
                st = make_step("jointnet", None, dev, lambda x: x, model_only=True)
                for nb in (1, 2, 4, 8):
                    d2 = build_batch([1000 + i for i in range(nb)], args.n_side, dev=dev)
                    sdt, sper, _ = timed_run(lambda: st(d2), 30, 5)
                    sb[f"B{nb}"] = dict(ms_per_forward=round(sdt / 30 * 1e3, 3), ms_median=round(pct(sper, 0.5), 3),
                                         meshes_per_s=round(nb * 30 / sdt, 1))
                    # the same batches through morig_amd.serving.ForwardServer: the serving default for <= 8 meshes -- per
                    # call the inputs are copied into the captured graph's static buffers, ONE replay, the guard read behind it
                    from morig_amd.serving import ForwardServer
                    srv = ForwardServer(st.model)
                    d3 = build_batch([2000 + i for i in range(nb)], args.n_side, dev=dev)       # a second batch of the same shape
                    if d3.geo_edge_index.shape != d2.geo_edge_index.shape:
                        d3 = d2
                    both = [d2, d3]
                    cnt = [0]

                    def served():
                        d_ = both[cnt[0] & 1]
                        cnt[0] += 1
                        return srv(d_, d_.pred_flow)[2]
                    gdt, gper, _ = timed_run(served, 30, 4)
                    sb[f"B{nb}"]["hipgraph_ms_per_forward"] = round(gdt / 30 * 1e3, 3)
                    sb[f"B{nb}"]["hipgraph_ms_median"] = round(pct(gper, 0.5), 3)
                    sb[f"B{nb}"]["hipgraph_meshes_per_s"] = round(nb * 30 / gdt, 1)
                    sb[f"B{nb}"]["server_stats"] = dict(srv.stats)
                    del srv, d3, both
                    del d2
            sb["per_mesh_throughput_B8_over_B64"] = round(sb["B8"]["meshes_per_s"] / (B_local * args.steps / dt), 3)
            secondary["small_batch"] = dict(metric="jointnet_motion forward at 1 / 2 / 4 / 8 meshes per GPU (4 k-vert synthetic)", **sb)
            del st
            torch.cuda.empty_cache()
        except Exception as e:
            secondary["small_batch"] = dict(error=repr(e)[:300])
        # a17: the batched geodesic-ball graph build that produces geo_edge_index (data_proc/common_ops.py:214-226) on the device
        try:
            from morig_amd import graph_build, synth
            gbatch = synth.collate([synth.make_mesh(1000 + i, n_side=args.n_side, geo="none", with_skin=False) for i in range(args.batch)]).to(dev)
            r_geo = 0.06 * 64.0 / args.n_side
            fn = lambda: graph_build.get_geo_edges(gbatch.pos, gbatch.batch, r_geo, 15, seed=1, self_loops=True, num_graphs=args.batch)
            sdt, _, ei = timed_run(fn, 10, 2)
            nv = gbatch.pos.shape[0]
            pairs_n = float(args.batch) * (nv / args.batch) ** 2
            secondary["geo_graph"] = dict(metric="batched geodesic-ball graph build (ball r = 0.06, <= 15 random members, + self loops), meshes/s",
                                          value=round(args.batch * 10 / sdt, 1), unit="meshes/s", ms_per_batch=round(sdt / 10 * 1e3, 3),
                                          batch=args.batch, edges=int(ei.shape[1]), pair_tests_per_s=round(pairs_n * 10 / sdt / 1e12, 3),
                                          pair_tests_unit="1e12 distance tests / s (V^2 per mesh: the kernel is bound by these, not by its "
                                                          "%.1f MB of compulsory HBM bytes)" % ((12.0 * nv + 16.0 * ei.shape[1]) / 1e6),
                                          includes="slot kernel + scan + one host read of the edge count + fill",
                                          config="VERDICT r2 a17; SURVEY 8(d) synthetic recipe")
            del gbatch, ei
        except Exception as e:
            secondary["geo_graph"] = dict(error=repr(e)[:300])
        # one TRAINING step of the headline network (train-mode forward with batch statistics, a stand-in L2 loss,
        # backward through the native backward operators; no optimizer) -- correctness-first kernels, reported for orientation
        try:
            from morig_amd import models as _models, synth
            nb = 8
            d2 = build_batch([2000 + i for i in range(nb)], args.n_side, with_skin=False, n_pts=0, dev=dev)
            tm = _models.jointnet_motion(num_keyframes=5, chn_output=3, aggr_method="attn").train()
            synth.load_recipe(tm, 0, mild=True).to(dev)

            def train_step():
                for p_ in tm.parameters():
                    p_.grad = None
                o = tm(d2, d2.pred_flow)
                loss = (o[2] ** 2).mean() + (o[1] ** 2).mean()
                loss.backward()
                return loss
            with torch.enable_grad():
                sdt, _, _ = timed_run(train_step, n_secondary, 1)
            secondary["train_step"] = dict(metric="meshes/sec jointnet_motion TRAINING step (train-mode forward + backward, no optimizer), 4 k-vert synthetic",
                                           value=round(nb * n_secondary / sdt, 2), unit="meshes/s", ms_per_step=round(sdt / n_secondary * 1e3, 3),
                                           steps=n_secondary, warmup=1, batch=nb, config="SURVEY 8(f-4); forward contractions on exact-fp32 MFMA (MORIG_TRAIN_PRECISION), gradient contractions on the bf16 x 3 split (MORIG_TRAIN_BWD)")
            del tm, d2
            torch.cuda.empty_cache()
        except Exception as e:                                   # the secondary lines never take the headline line down
            secondary["train_step"] = dict(error=repr(e)[:300])
        # CorrNet in model.train() (morig_amd/train_corr.py; training/train_corr_pose.py:61-70): forward with batch statistics over
        # vertices, mesh edges and ball-query edges, a stand-in loss on the three outputs, backward -- 8 pairs of a 4 k-vertex mesh
        # and an 8 k-point cloud
        try:
            from morig_amd import models as _models, synth
            nbc = 8
            dc = build_batch([3000 + i for i in range(nbc)], args.n_side, with_skin=False, n_pts=8192, dev=dev)
            cm = _models.corrnet(input_feature=3, output_feature=64, temprature=0.07).train()
            synth.load_recipe(cm, 0, mild=True).to(dev)

            def corr_train_step():
                for p_ in cm.parameters():
                    p_.grad = None
                ov, op, vis, _ = cm(dc, True)
                loss = (ov[:, ::7] ** 2).mean() + (op[:, ::5] ** 2).mean() + (vis ** 2).mean()
                loss.backward()
                return loss
            with torch.enable_grad():
                sdt, _, _ = timed_run(corr_train_step, n_secondary, 1)
            secondary["corrnet_train_step"] = dict(metric="pairs/sec corrnet TRAINING step (train-mode forward + backward, no optimizer), 4 k-vert mesh + 8 k-point cloud per pair",
                                                   value=round(nbc * n_secondary / sdt, 2), unit="pairs/s", ms_per_step=round(sdt / n_secondary * 1e3, 3),
                                                   steps=n_secondary, warmup=1, batch=nbc, config="SURVEY 8(f-4); forward contractions on exact-fp32 MFMA (MORIG_TRAIN_PRECISION), gradient contractions on the bf16 x 3 split (MORIG_TRAIN_BWD)")
            del cm, dc
            torch.cuda.empty_cache()
        except Exception as e:
            secondary["corrnet_train_step"] = dict(error=repr(e)[:300])
        # the joint extraction that follows the networks (evaluate/eval_rigging.py:72-95): per mesh 4096 shifted points
        # + their mirror images, bandwidth, 29 weighted mean-shift steps, NMS, flip -- for ALL meshes of the batch at once
        # (segment-aware kernels), next to the one-mesh-per-call form the reference has
        try:
            import numpy as np
            from morig_amd import joints as _joints
            rng = np.random.default_rng(3)
            nbj = args.batch
            halves, attns = [], []
            for _ in range(nbj):
                centres = rng.uniform(-0.4, 0.4, (20, 3)); centres[:, 0] = -np.abs(centres[:, 0])
                halves.append(centres[rng.integers(0, 20, 4096)] + rng.normal(0, 0.03, (4096, 3)))
                attns.append((rng.random((4096, 1)) ** 2).astype(np.float32))
            jp = torch.from_numpy(np.concatenate(halves)).to(dev)
            ja = torch.from_numpy(np.concatenate(attns)).to(dev)
            jb = torch.arange(nbj, device=dev).repeat_interleave(4096)
            n_found = []

            def joint_step():
                outs = _joints.extract_joints_batched(jp, ja, jb, None, 0.04, -1.0, 0.02, 30, num_graphs=nbj)
                n_found.append(sum(len(o["joints"]) for o in outs) / nbj)
                return jp
            reps = max(3, n_secondary)
            sdt, _, _ = timed_run(joint_step, reps, 1)
            jp1, ja1 = jp[:4096].contiguous(), ja[:4096].contiguous()
            sdt1, _, _ = timed_run(lambda: (_joints.extract_joints(jp1, ja1, None, 0.04, -1.0, 0.02, 30), jp1)[1], reps, 1)
            secondary["joint_extraction"] = dict(metric="meshes/sec joint extraction (mirror, bandwidth, mean-shift, NMS) from 4096 shifted points per mesh, batched over the meshes",
                                                 value=round(nbj * reps / sdt, 2), unit="meshes/s", ms_per_step=round(sdt / reps * 1e3, 3),
                                                 steps=reps, warmup=1, batch=nbj, joints_found_per_mesh=n_found[-1],
                                                 one_mesh_per_call=dict(value=round(reps / sdt1, 2), unit="meshes/s", ms_per_mesh=round(sdt1 / reps * 1e3, 3)),
                                                 config="SURVEY 8(f-2); float64 kernels")
        except Exception as e:
            secondary["joint_extraction"] = dict(error=repr(e)[:300])

    if rank == 0:
        names = NAMES[args.workload]
        roof, breakdown, hbm_kinds, all_flops = None, {}, {}, 0.0
        psteps = max(args.prof_steps, 1)
        if prof:
            roof, breakdown, total_ms = roofline_of(prof, psteps, clocks)
            if roof.get("bound") == "mfma" and args.power_probe_seconds > 0:
                try:
                    pp = mfma_power_probe(args.power_probe_seconds, local)
                    roof["mfma_power_capped"] = pp
                    roof["mfma_power_capped_tflops"] = pp["tflops"]
                    k = roof.get("mfma_issued_per_product") or 1
                    roof["frac_of_power_capped_peak"] = round(k * roof["achieved"] / pp["tflops"], 4) if pp["tflops"] else None
                except Exception as e:
                    roof["mfma_power_capped"] = dict(error=repr(e)[:200])
            # the index / copy kernels are the HBM-bound ones: bytes the launcher declares / event time, against 8 TB/s
            for k in ("csr_build", "copy", "rownorm"):
                v = prof.get(k)
                if v and v["ms"] > 0 and v.get("bytes", 0) > 0:
                    gbs = v["bytes"] / (v["ms"] * 1e-3) / 1e9
                    hbm_kinds[k] = dict(gb_per_s=round(gbs, 1), hbm_frac=round(gbs / PEAK_HBM_GBS, 4),
                                        ms_per_step=round(v["ms"] / psteps, 3))
            all_flops = sum(v["flops"] for v in prof.values()) / psteps
        n_units = world * B_local
        res = {
            "metric": names[0],
            "value": round(n_units * args.steps / dt, 2), "unit": "pairs/s" if pairs else "meshes/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 3),
            "ms_per_step_median": round(pct(per_step, 0.5), 3), "ms_per_step_p10": round(pct(per_step, 0.1), 3),
            "ms_per_step_p90": round(pct(per_step, 0.9), 3),
            "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None,
            "dtype": ("f32 (split-fp16: 3x f16 MFMA, f32 accumulate)" if (PLUMBING or native.get_ops().precision == "f16x3") else "f32"),
            "data": "synthetic" if not PLUMBING else "synthetic (PLUMBING RUN on CPU emulation + gloo: not a measurement)",
            # (workload stays under 120 characters: the driver's record cuts longer strings -- VERDICT r5 #10; what a step covers is `step`)
            "config": {"workload": (f"{names[1]}, B={B_local} x {args.n_side * args.n_side}-vertex synthetic meshes/GPU, {names[2]}" if not ragged else
                                    f"{names[1]}, B={B_local} ragged meshes/GPU ({n_vert} vertices), {names[2]}"),
                       "step": "COO->CSR prep + forward" + (" + RCCL all-gather of the outputs" if use_dist else ""),
                       "meshes_per_gpu": B_local, "global_batch": n_units,
                       "vertices_per_mesh": (args.n_side * args.n_side) if not ragged else [min(sides) ** 2, max(sides) ** 2],
                       "parallelism": f"mesh-sharded dp{world}",
                       # guard_read: "sync" = one host read of the range flag / CSR status at the end of every forward; "deferred" =
                       # read one forward later (forward_async), every forward's still inside the timed region
                       "guard_read": "sync" if (args.workload not in ("jointnet", "jointnet_ragged") or os.environ.get("MORIG_BENCH_GUARD") == "sync") else "deferred",
                       "geo_graph": "device" if not PLUMBING else "host"},
            "rccl_ranks": rccl_ranks, "backend": backend if use_dist else None,
            "per_rank_ms_per_step": [round(x / args.steps * 1e3, 3) for x in per_rank],
            "allgather_ms_per_step": round(allgather_ms, 4) if allgather_ms is not None else None,
            "allgather_calls_per_step": len(gathered) if allgather_ms is not None else None,
            "allgather_bytes_per_rank": sum(g_t.numel() * g_t.element_size() for g_t in gathered) if allgather_ms is not None else None,
            "allgather_note": ("the step's %d all_gather_into_tensor call(s) (%s bytes per rank in total) repeated 20x back to back between "
                               "fences in a separate pass, max over ranks; inside the timed steps the same calls are part of ms_per_step"
                               % (len(gathered), sum(g_t.numel() * g_t.element_size() for g_t in gathered))) if allgather_ms is not None else None,
            "vertices_per_s": round(n_vert_all * args.steps / dt, 1),
            "roofline": roof,
            "hbm_bound_kernels": hbm_kinds,
            "whole_forward_tflops": round(all_flops / (dt / args.steps) / 1e12, 2) if prof else None,
            "kernels": breakdown,
            "secondary": secondary,
        }
        res["clocks_under_load"] = clocks
        if PLUMBING:
            res["roofline"] = dict(bound="mfma", achieved=None, peak=PEAK_F16_MFMA_TFLOPS, unit="TFLOP/s", frac=None, traffic=None,
                                   note="PLUMBING RUN: no HIP kernel ran (CPU emulation of the op layer); the key is here so that the "
                                        "line's schema can be checked at any world size")
        # the CPU oracle on this host's cores, on rank 0, AFTER the timed region (the other ranks wait at the closing barrier), at
        # every world size: an N > 1 line without it would be unmeasured by rule (VERDICT r3 #1b)
        if args.cpu_seconds > 0 and not PLUMBING and args.workload in ("jointnet", "jointnet_ragged"):
            res["cpu_baseline"] = cpu_baseline(args.cpu_seconds, args.n_side, 1000)
        elif args.cpu_seconds > 0 and not PLUMBING:
            res["cpu_baseline"] = cpu_baseline_other(args.workload, args.cpu_seconds, args.n_side, args.n_pts, 1000)
        elif PLUMBING and args.cpu_seconds > 0 and args.workload in ("jointnet", "jointnet_ragged"):
            res["cpu_baseline"] = cpu_baseline(args.cpu_seconds, args.n_side, 1000)      # the real oracle, on the plumbing run's tiny meshes
        elif PLUMBING and args.cpu_seconds > 0:
            res["cpu_baseline"] = cpu_baseline_other(args.workload, args.cpu_seconds, args.n_side, args.n_pts, 1000)
        else:
            res["cpu_baseline"] = None
        emit(res)
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
