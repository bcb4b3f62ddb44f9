"""(2+1)D factorised convolution on HIP kernels; same class name, constructor signature, attribute names and
state_dict keys as the reference's models/spatiotempconv.py:7-65 (acts as a drop-in nn.Conv3d)."""
import math

import torch.nn as tnn
from torch.nn.modules.utils import _triple

from .. import _lib
from .. import functional as F
from .. import nn as hnn
from ..functional import ClTensor


class SpatioTemporalConv(tnn.Module):
    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, bias=True):
        super(SpatioTemporalConv, self).__init__()
        kernel_size = _triple(kernel_size)
        stride = _triple(stride)
        padding = _triple(padding)

        spatial_kernel_size = [1, kernel_size[1], kernel_size[2]]
        spatial_stride = [1, stride[1], stride[2]]
        spatial_padding = [0, padding[1], padding[2]]
        temporal_kernel_size = [kernel_size[0], 1, 1]
        temporal_stride = [stride[0], 1, 1]
        temporal_padding = [padding[0], 0, 0]

        # M of the R(2+1)D paper, section 3.5 (reference :44-45)
        intermed_channels = int(math.floor((kernel_size[0] * kernel_size[1] * kernel_size[2] * in_channels * out_channels) /
                                           (kernel_size[1] * kernel_size[2] * in_channels + kernel_size[0] * out_channels)))

        self.spatial_conv = hnn.Conv3d(in_channels, intermed_channels, spatial_kernel_size,
                                       stride=spatial_stride, padding=spatial_padding, bias=bias)
        self.bn = hnn.BatchNorm3d(intermed_channels)
        self.relu = hnn.ReLU()
        self.temporal_conv = hnn.Conv3d(intermed_channels, out_channels, temporal_kernel_size,
                                        stride=temporal_stride, padding=temporal_padding, bias=bias)

    def forward(self, x, stats=None, bias_token=None):
        """spatial conv -> BatchNorm+ReLU (one fused pass; bf16: statistics from the conv epilogue) -> temporal conv (bf16: its
        data gradient carries the BatchNorm+ReLU backward reduce, nn.run_fused).
        `stats` (optional [2*Cp] float32 zeros) receives the temporal conv's per-channel sum / sum of squares for the
        BatchNorm the callers apply next (models/mygannet.py:24-26,113-115); `bias_token`: that BatchNorm's backward also
        produces the temporal conv's bias gradient (functional.bn_act)."""
        plain = not isinstance(x, ClTensor)
        if plain:
            x = F.to_cl(x)
        x = hnn.run_fused([self.spatial_conv, self.bn, self.relu, self.temporal_conv], x, last_stats=stats, last_bias_token=bias_token)
        return x.to_torch() if plain else x
