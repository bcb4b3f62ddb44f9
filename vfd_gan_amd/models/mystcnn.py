"""(2+1)D auto-encoder baseline on HIP kernels, behind the surface of the reference's models/mystcnn.py
(C2plus1d_Block :6-49, AutoEncoder :52-88).  Same constructor signatures, attribute names and
state_dict keys.  It reuses the hot path's kernels only: (1,3,3) / (3,1,1) / 1x1x1 / 3x3x3 Conv3d, BatchNorm3d + ReLU (with the
AvgPool3d(2) of the down path inside the BatchNorm pass), trilinear x2 up-sampling written together with the channel
concatenation, Dropout.  ``MyGAN`` takes it as its generator under ``--ae`` (models/mygannet.py:224-227 intends that; the
reference itself builds an instance and then calls it without an input, so the flag cannot run there)."""
import torch.nn as tnn

from .. import _lib
from .. import functional as F
from .. import nn as hnn
from ..functional import ClTensor


class C2plus1d_Block(tnn.Module):
    def __init__(self, in_ch, out_ch, k=5):
        super(C2plus1d_Block, self).__init__()
        self.conv = hnn.Conv3d(in_ch, out_ch, 1, stride=1)

        self.spaceconv = hnn.Conv3d(in_ch, in_ch, (1, 3, 3), stride=1, padding=(0, 1, 1), dilation=1, bias=False)
        self.pointwise = hnn.Conv3d(in_ch, out_ch, (3, 1, 1), stride=1, padding=(1, 0, 0), dilation=1, bias=False)

        self.bn1 = hnn.BatchNorm3d(in_ch)
        self.bn2 = hnn.BatchNorm3d(out_ch)

        self.avgpool = hnn.AvgPool3d(2)
        self.dropout = hnn.Dropout(p=0.25)
        self.upsamp = hnn.Upsample(scale_factor=2, mode='trilinear', align_corners=True)

        self.relu = hnn.ReLU(inplace=True)
        self.conv_last = hnn.Conv3d(out_ch + out_ch, out_ch, 3, stride=1, padding=1, dilation=1, bias=False)

    def forward(self, x, down_samp=False):
        plain = not isinstance(x, ClTensor)
        if plain:
            x = F.to_cl(x)
        inp = x
        main = [self.spaceconv, self.bn1, self.relu, self.pointwise, self.bn2, self.relu]
        if down_samp:
            x = hnn.run_fused(main + [self.avgpool], x)          # BatchNorm -> ReLU -> AvgPool3d(2) in one pass where it can be
            inp = self.avgpool(self.conv(inp))
            x = F.cat_channels(x, inp)
        else:
            x = hnn.run_fused(main, x)
            self.upsamp.check()
            inp = self.conv(self.upsamp(self.dropout(inp)))
            x = F.upsample_cat(x, inp)                           # cat([Upsample(x), inp]) without the up-sampled intermediate
        x = self.conv_last(x)
        return x.to_torch() if plain else x


class AutoEncoder(tnn.Module):
    def __init__(self):
        super(AutoEncoder, self).__init__()

        self.down_sep1 = C2plus1d_Block(3, 64)
        self.down_sep2 = C2plus1d_Block(64, 128)
        self.down_sep3 = C2plus1d_Block(128, 256)
        self.down_sep4 = C2plus1d_Block(256, 512)

        self.up_sep1 = C2plus1d_Block(512, 256)
        self.up_sep2 = C2plus1d_Block(256 + 256, 256)
        self.up_sep3 = C2plus1d_Block(256 + 128, 128)
        self.up_sep4 = C2plus1d_Block(128 + 64, 64)

        self.conv_last = hnn.Conv3d(64, 1, 3, stride=1, padding=1, bias=False)
        self.sigmoid = hnn.Sigmoid()

    def forward(self, x):
        plain = not isinstance(x, ClTensor)
        if plain:
            x = F.to_cl(x)
        # Encoder (T, H, W halve per level: 16x128x128 -> 1x8x8)
        down_sep1 = self.down_sep1(x, down_samp=True)
        down_sep2 = self.down_sep2(down_sep1, down_samp=True)
        down_sep3 = self.down_sep3(down_sep2, down_samp=True)
        down_sep4 = self.down_sep4(down_sep3, down_samp=True)
        # Decoder with skip connections
        up_sep1 = self.up_sep1(down_sep4, down_samp=False)
        up_sep2 = self.up_sep2(F.cat_channels(up_sep1, down_sep3), down_samp=False)
        up_sep3 = self.up_sep3(F.cat_channels(up_sep2, down_sep2), down_samp=False)
        up_sep4 = self.up_sep4(F.cat_channels(up_sep3, down_sep1), down_samp=False)
        predict = self.conv_last(up_sep4, act=_lib.ACT_SIGMOID)      # conv_last + sigmoid in one kernel
        return predict.to_torch() if plain else predict
