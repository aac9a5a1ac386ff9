"""SharedMLP: a stack of 1x1 conv + BN + ReLU blocks shared over the
point (ndim=1) or pixel (ndim=2) axis; an ``nn.ModuleList`` like the reference (nn/mlp.py:45-81) so the
state-dict keys are ``<i>.conv.weight`` / ``<i>.bn.*``."""
from torch import nn

from .conv import Conv1d, Conv2d


class SharedMLP(nn.ModuleList):
    def __init__(self, in_channels, mlp_channels, ndim=1, bn=True, bn_momentum=0.1):
        super(SharedMLP, self).__init__()
        if ndim not in (1, 2):
            raise ValueError()
        block = Conv1d if ndim == 1 else Conv2d
        self.in_channels = in_channels
        width = in_channels
        for out_channels in mlp_channels:
            self.append(block(width, out_channels, 1, relu=True, bn=bn, bn_momentum=bn_momentum))
            width = out_channels
        self.out_channels = width

    def forward(self, x):
        if x.dim() == 3 and len(self) > 0:
            from .. import pointflow
            if pointflow.hip_inference(x, self):           # no autograd graph: the chain on the HIP GEMM kernels
                blocks = list(self)
                if pointflow.shared_mlp_supported(blocks):
                    from .. import graph
                    return graph.module_forward(self, lambda v: pointflow.shared_mlp_forward(blocks, v), x)
        for layer in self:
            x = layer(x)
        return x
