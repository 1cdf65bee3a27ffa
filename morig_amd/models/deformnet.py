"""Drop-in mirror of the reference's ``models/deformnet.py`` (/root/reference/models/deformnet.py:13-104): same
class / factory names, ``forward(data)`` signature, returned tuple and state_dict keys (``corr_extractor.*``,
``completing.gcu_{1,2,3}.*``, ``completing.mlp_glb.*``, ``completing.mlp_tramsform.*`` (sic)); eval-mode arithmetic on
the MI355X-native op layer. DeformNet produces the ``pred_flow`` the rig networks consume
(datasets/dataset_rig.py:111-115) -- SURVEY.md 8(f-1).

Restructurings (exact up to fp32 rounding unless noted):
  * ``GCNDeform`` is ``GCNRig``'s wiring at widths 128/256/512: one wide activation buffer, pooled GEMM epilogue,
    x_global as a per-mesh row bias (morig_amd/models/rignet.py);
  * the visible / invisible split (:57-63) is not compacted: the k-NN kernel takes the normalised mask and lets rows
    with mask < 0.5 query rows with mask >= 0.5 of the same cloud (``morig_cosine_knn`` split mode); neighbour
    order, tie rule (lowest index) and the vote's summation order are those of the compacted formulation;
  * ``knn(..., cosine=True)`` re-normalises its inputs; CorrNet's features are already unit rows, so the kernel
    ranks by the plain dot product (orders can differ only between candidates whose similarities agree to ~1e-7).
"""
from __future__ import annotations

import torch

from ..native import Mat
from ..runtime import get_ops
from .basic_modules import NativeModule
from .corrnet import CorrNet
from .rignet import GCNRig

__all__ = ["deformnet"]


class GCNDeform(GCNRig):
    """models/deformnet.py:13-32. ``forward`` takes geo before tpl (:23)."""

    WIDTHS = (128, 256, 512)
    TRANSFORM = "mlp_tramsform"

    def __init__(self, chn_in, chn_output, aggr="max"):
        super().__init__(chn_in, chn_output, aggr=aggr)

    def _forward(self, pos, feature, geo_edge_index, tpl_edge_index, batch):
        return super()._forward(pos, feature, tpl_edge_index, geo_edge_index, batch)


class DeformNet(NativeModule):
    """models/deformnet.py:35-99."""

    def __init__(self, tau_nce, num_interp):
        super().__init__()
        self.corr_extractor = CorrNet(3, 64, temprature=tau_nce)
        self.completing = GCNDeform(chn_in=4, chn_output=3)
        self.num_interp = num_interp
        self.last_neighbours = None

    def _pack(self):
        return {}

    def _forward(self, data):
        ops = get_ops()
        dev = data.vtx.device
        k = int(self.num_interp)
        # CorrNet with its defaults: train_vismask=True, random_start=True (:41)
        vtx_f, pts_f, logit, tau = self.corr_extractor._forward(data, True)
        n = vtx_f.shape[0]
        plan = self.corr_extractor.last_plan              # offsets / counts of this very forward: no second count, no sync
        B, vcounts = plan.B, plan.vcounts
        vb = data.vtx_batch
        ptr_v, ptr_p = plan.ptr_v, plan.ptr[0]

        vis = torch.empty((n, 1), dtype=torch.float32, device=dev)
        ops.sigmoid_minmax(Mat.of(logit), ptr_v, B, Mat.of(vis))                      # :42-46
        vtx4 = torch.zeros((n, 4), dtype=torch.float32, device=dev)
        ops.copy2d(Mat.of(data.vtx.float().contiguous()), Mat.of(vtx4, 0, 3))
        pts = data.pts.float().contiguous()
        l1 = torch.empty((n, 4), dtype=torch.float32, device=dev)                     # [flow_init | pred_vismask]  (:97)
        # visible part: every vertex votes from its k most similar points (:49-54)
        idx = ops.cosine_knn(Mat.of(vtx_f), ptr_v, Mat.of(pts_f), ptr_p, B, max(vcounts), k)
        ops.flow_vote(0, idx, Mat.of(vtx_f), Mat.of(pts_f), Mat.of(vtx4, 0, 3), Mat.of(pts), Mat.of(vis), Mat.of(l1))
        # invisible part: vertices with mask < 0.5 vote from their k most similar visible vertices (:57-95)
        idx2 = ops.cosine_knn(Mat.of(vtx_f), ptr_v, Mat.of(vtx_f), ptr_v, B, max(vcounts), k, vis=Mat.of(vis), split=True)
        ops.flow_vote(1, idx2, Mat.of(vtx_f), Mat.of(vtx_f), None, None, Mat.of(vis), Mat.of(l1))

        self.last_neighbours = (idx, idx2)            # [n, k] int32 tables (-1 padded); inspection / tie-aware parity tests
        gd = self.completing
        pred_flow = torch.empty((n, gd.chn_output), dtype=torch.float32, device=dev)
        seg = ops.make_seg(vb, B, 1)
        # the 4-aligned CSRs of the two graphs were built by the CorrNet forward above; the 16-wide position layers run on them
        # too (padding repeats an edge: harmless under max, ~20 % more rows on layers that cost 0.1 ms; four CSR builds saved)
        csr_tpl4, csr_geo4 = self.corr_extractor.last_csr or (ops.csr_build(data.tpl_edge_index, n, pad4=True),
                                                              ops.csr_build(data.geo_edge_index, n, pad4=True))
        self.corr_extractor.last_csr = None               # consumed: the device buffers are not pinned between forwards
        gd.run(ops, vtx4, lambda w, sp: ops.copy2d_pad(Mat.of(l1), w, split=sp), csr_tpl4, csr_geo4, seg, B, 1, Mat.of(pred_flow),
               csr_geo_wide=csr_geo4, csr_tpl_wide=csr_tpl4)
        return pred_flow, vtx_f, pts_f, vis, tau


    # ---- model.train() (training/train_deform_pose.py:29-40) ----
    def _forward_train_grad(self, data):
        from .. import train_corr
        return train_corr.deformnet_step(self, data)

    def _forward_train(self, data):
        from .. import train_corr
        return train_corr.deformnet_step(self, data)


def deformnet(**kwargs):
    return DeformNet(tau_nce=kwargs["tau_nce"], num_interp=kwargs["num_interp"])
