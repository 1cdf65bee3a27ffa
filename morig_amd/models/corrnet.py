"""Drop-in mirror of the reference's ``models/corrnet.py`` (/root/reference/models/corrnet.py:10-82)."""
from __future__ import annotations

import torch
from torch.nn import Linear as Lin, Parameter, Sequential as Seq

from .. import packing
from ..native import Mat
from ..runtime import get_ops
from .basic_modules import GCU, MLP, NativeModule

__all__ = ["corrnet"]


class CorrNet(NativeModule):
    def __init__(self, input_feature, output_feature, temprature, aggr="max"):
        super().__init__()
        self.input_feature = input_feature
        self.output_feature = output_feature
        self.temprature = Parameter(torch.Tensor([temprature]))
        self.vtx_gcu_1 = GCU(in_channels=3, out_channels=32, aggr=aggr)
        self.vtx_gcu_2 = GCU(in_channels=32, out_channels=64, aggr=aggr)
        self.vtx_gcu_3 = GCU(in_channels=64, out_channels=256, aggr=aggr)
        self.vtx_gcu_4 = GCU(in_channels=256, out_channels=512, aggr=aggr)
        self.vtx_mlp_glb = MLP([(32 + 64 + 256 + 512), 1024])
        self.vtx_mlp = Seq(MLP([1024 + 3 + 32 + 64 + 256 + 512, 1024, 256]), Lin(256, output_feature))

    def forward(self, data, train_vismask, random_start=True):
        raise NotImplementedError("CorrNet native path: under construction")


def corrnet(**kwargs):
    return CorrNet(input_feature=kwargs["input_feature"], output_feature=kwargs["output_feature"],
                   temprature=kwargs["temprature"])
