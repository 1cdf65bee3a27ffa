"""f-4, one whole-network check that is NOT statistical. On a batch where every max-aggregation of a GCNRig forward is
decided by a clear margin -- the top-2 gap of every (target, channel) maximum exceeds 3e-5 of the aggregated tensor's scale in the
float64 oracle (found by tools/no_tie_search.py; a float32 forward is off by ~1e-6) -- a float32 implementation routes every gradient
through the SAME edge as the oracle, and the whole-network gradients are held to a criterion that is NOT statistical: for EVERY
parameter tensor the cosine with the float64 gradient is >= 1 - 1e-5 (the statistical tests accept 0.95-0.99; measured: >= 1 - 3e-6 on MI355X, a 16-entry bias next to one ReLU kink), 99 % of its entries are
within 1e-3 of the tensor's scale (or four times what torch's own float32 autograd shows on the same tensor): the block criterion
is 2e-4, but these batches are tiny so that a tie-free one can be found at all -- 16-18 vertices -- and BatchNorm over 16 rows costs
float32 a few 1e-4 on the first layers' tensors (measured worst: 3-6e-4 here, 0.8-1.8e-4 for torch's own float32 run; cosine >= 0.9999998). A routing bug confined to near-tie edges moves a whole gradient column and
cannot pass. Runs on the CPU emulation of the op layer and, marked gpu, on the HIP kernels."""
import copy
import os
import sys

import pytest
import torch

from conftest import ROOT
from helpers import PARITY_LOG
from morig_amd import runtime, synth
from morig_amd.models import rignet as rn

sys.path.insert(0, os.path.join(ROOT, "tools"))

# (n_side, meshes, seed) found by `python tools/no_tie_search.py <n_side> <meshes> 4000`: min top-2 gaps 4.6e-5 / 3.3e-5 / 3.4e-5 of scale. (A case
# whose margin is smaller than the float32 forward noise of its own tiny batch -- (3, 2, 494): gap 4.2e-5, forward error 8e-5 -- does flip
# one route and fails the cosine bound at 0.999999: the criterion sees what it is meant to see.)
CASES = [(4, 1, 676), (4, 1, 621), (3, 2, 280)]
MIN_GAP = 3e-5


def _case(n_side, n_mesh, seed):
    import no_tie_search as S
    from oracle import nets
    torch.manual_seed(seed)
    ref = S.randomise(nets.RigGCN(chn_feature=3, chn_output=8), seed).train().double()
    b = synth.make_batch(range(100 + seed, 100 + seed + n_mesh), n_side=n_side, with_skin=False)
    g = torch.Generator().manual_seed(seed)
    feat = torch.randn(b.pos.shape[0], 3, generator=g) * 0.05
    gap, _ = S.min_gap(copy.deepcopy(ref), b, feat)
    return ref, b, feat, gap


def _run(n_side, n_mesh, seed, dev):
    from morig_amd import train_backward as TB
    ref, b, feat, gap = _case(n_side, n_mesh, seed)
    if gap < MIN_GAP:
        pytest.skip(f"this torch build initialises the layers differently: min top-2 gap {gap:.2e} (re-run tools/no_tie_search.py)")
    mine = rn.GCNRig(chn_feature=3, chn_output=8).train()
    mine.load_state_dict(copy.deepcopy(ref).float().state_dict())
    mine.to(dev)
    g = torch.Generator().manual_seed(seed + 1)
    w = torch.randn(b.pos.shape[0], 8, generator=g)
    ref32 = copy.deepcopy(ref).float()
    with torch.enable_grad():
        o = ref(b.pos.double(), feat.double(), b.tpl_edge_index, b.geo_edge_index, b.batch)
        (o * w.double()).sum().backward()
        o32 = ref32(b.pos.float(), feat.float(), b.tpl_edge_index, b.geo_edge_index, b.batch)
        (o32 * w).sum().backward()
        bd = b.to(dev)
        st = TB.graph_state(bd)
        om = TB.gcnrig(mine, bd.pos.float(), feat.to(dev), st["csr_tpl"], st["csr_geo"], st["batch"], st["mesh_ptr"], st["ng"])
        (om * w.to(dev)).sum().backward()
    scale = float(o.abs().max())
    err = float((om.detach().cpu().double() - o.detach()).abs().max()) / scale
    assert err <= 5e-4, err                                  # the forward first (tiny batches: BatchNorm over 16-18 rows amplifies float32 rounding)
    worst = (0.0, "")
    for (k, p), (_, q), (_, q32) in zip(mine.named_parameters(), ref.named_parameters(), ref32.named_parameters()):
        assert p.grad is not None and p.grad.shape == q.grad.shape, k
        a, r, r32 = p.grad.detach().cpu().double().flatten(), q.grad.flatten(), q32.grad.double().flatten()
        s = max(float(r.abs().max()), 1e-12)
        ev = (a - r).abs() / s
        e, e32 = float(ev.max()), float((r32 - r).abs().max()) / s
        q99 = float(torch.quantile(ev, 0.99)) if ev.numel() > 100 else e
        cos = float(torch.dot(a, r) / (a.norm() * r.norm() + 1e-300))
        PARITY_LOG.append((f"backward:no_ties_{n_side}x{n_mesh}:{k}", e * s, s, e))
        worst = max(worst, (e, k))
        assert cos >= 1.0 - 1e-5, (k, cos, gap)
        # (an isolated entry next to a ReLU kink is decided by the forward's last bit on any float32 implementation: q99 / max as in the
        # block tests)
        assert q99 <= max(1e-3, 4.0 * e32) and e <= 2e-2, (k, q99, e, e32, cos, gap)
    return worst, gap


@pytest.mark.parametrize("n_side,n_mesh,seed", CASES)
def test_whole_network_gradients_meet_the_block_criterion_without_near_ties_emulated(n_side, n_mesh, seed):
    from emulate import EmuOps
    runtime._test_ops = EmuOps()
    try:
        _run(n_side, n_mesh, seed, "cpu")
    finally:
        runtime._test_ops = None


@pytest.mark.gpu
@pytest.mark.parametrize("n_side,n_mesh,seed", CASES)
def test_whole_network_gradients_meet_the_block_criterion_without_near_ties(n_side, n_mesh, seed, monkeypatch):
    monkeypatch.setenv("MORIG_TRAIN_PRECISION", "f32")
    _run(n_side, n_mesh, seed, "cuda")
