"""(2+1)D factorised convolution, restatement of reference models/spatiotempconv.py:7-65."""
import math

import torch.nn as nn
from torch.nn.modules.utils import _triple


def intermed_channels(in_channels, out_channels, kernel_size):
    """:44-45  M = floor(kt*kh*kw*in*out / (kh*kw*in + kt*out))."""
    kt, kh, kw = _triple(kernel_size)
    return int(math.floor((kt * kh * kw * in_channels * out_channels) / (kh * kw * in_channels + kt * out_channels)))


class SpatioTemporalConv(nn.Module):
    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, bias=True):   # :22
        super().__init__()
        kernel_size, stride, padding = _triple(kernel_size), _triple(stride), _triple(padding)
        m = intermed_channels(in_channels, out_channels, kernel_size)
        self.spatial_conv = nn.Conv3d(in_channels, m, [1, kernel_size[1], kernel_size[2]],                 # :49-50
                                      stride=[1, stride[1], stride[2]], padding=[0, padding[1], padding[2]], bias=bias)
        self.bn = nn.BatchNorm3d(m)                                                                         # :51
        self.relu = nn.ReLU()                                                                               # :52
        self.temporal_conv = nn.Conv3d(m, out_channels, [kernel_size[0], 1, 1],                            # :59-60
                                       stride=[stride[0], 1, 1], padding=[padding[0], 0, 0], bias=bias)

    def forward(self, x):                                                                                   # :62-65
        return self.temporal_conv(self.relu(self.bn(self.spatial_conv(x))))
