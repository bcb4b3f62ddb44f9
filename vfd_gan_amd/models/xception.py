"""Per-frame Xception encoder + up-sampling decoder baseline on HIP kernels, behind the surface of the reference's
models/xception.py (SepaConv :6-21, Block :23-71, DeConv :73-88, Xception :92-174).  Same
constructor signatures, attribute names and state_dict keys.  New to the kernel set for this net only: MaxPool3d (vfd_maxpool_*),
the residual add (vfd_add) and the (1,2,2) trilinear up-sampling (vfd_upsample_*)."""
import torch.nn as tnn

from .. import _lib
from .. import functional as F
from .. import nn as hnn
from ..functional import ClTensor

__all__ = ['xception']


class SepaConv(tnn.Module):
    def __init__(self, in_ch, out_ch):
        super(SepaConv, self).__init__()
        self.conv1 = hnn.Conv3d(in_ch, in_ch, (1, 3, 3), stride=1, padding=(0, 1, 1), dilation=1, bias=False)
        self.pointwise = hnn.Conv3d(in_ch, out_ch, (1, 1, 1), stride=1, padding=(0, 0, 0), dilation=1, bias=False)
        self.relu = hnn.ReLU()

    def forward(self, x):
        # conv -> ReLU -> conv -> ReLU: both activations ride in the conv epilogues
        return hnn.run_fused([self.conv1, self.relu, self.pointwise, self.relu], x)


class Block(tnn.Module):
    def __init__(self, in_fi, out_fi, reps, strides=1, start_with_relu=True, grow_first=True):
        super(Block, self).__init__()

        if out_fi != in_fi or strides != 1:
            self.skip = hnn.Conv3d(in_fi, out_fi, 1, stride=(1, strides, strides), bias=False)
            self.skipbn = hnn.BatchNorm3d(out_fi)
        else:
            self.skip = None

        self.relu = hnn.ReLU(inplace=True)
        rep = []
        filters = in_fi

        if grow_first:
            rep.append(self.relu)
            rep.append(SepaConv(in_fi, out_fi))
            rep.append(hnn.BatchNorm3d(out_fi))
            filters = out_fi

        for i in range(reps - 1):
            rep.append(self.relu)
            rep.append(SepaConv(filters, filters))
            rep.append(hnn.BatchNorm3d(filters))

        if not grow_first:
            rep.append(self.relu)
            rep.append(SepaConv(in_fi, out_fi))
            rep.append(hnn.BatchNorm3d(out_fi))

        if not start_with_relu:
            rep = rep[1:]
        else:
            rep[0] = hnn.ReLU(inplace=False)

        if strides != 1:
            rep.append(hnn.MaxPool3d((1, 3, 3), (1, strides, strides), padding=(0, 1, 1)))
        self.rep = hnn.Sequential(*rep)

    def forward(self, inp):
        x = self.rep(inp)
        if self.skip is not None:
            skip = hnn.run_fused([self.skip, self.skipbn], inp)
        else:
            skip = inp
        return F.add(x, skip)            # `x += skip`


class DeConv(tnn.Module):
    def __init__(self, in_fi, out_fi):
        super(DeConv, self).__init__()
        self.conv = hnn.Conv3d(in_fi, out_fi, (1, 3, 3), stride=1, padding=(0, 1, 1), bias=False)
        self.bn = hnn.BatchNorm3d(out_fi)
        self.lrelu = hnn.LeakyReLU(0.2, inplace=True)
        self.dropout = hnn.Dropout(p=0.25)
        self.upsamp = hnn.Upsample(scale_factor=(1, 2, 2), mode='trilinear', align_corners=True)

    def forward(self, x):
        return hnn.run_fused([self.conv, self.bn, self.lrelu, self.dropout, self.upsamp], x)


class Xception(tnn.Module):
    def __init__(self, ich=3):
        super(Xception, self).__init__()

        self.conv1 = hnn.Conv3d(ich, 32, (1, 3, 3), stride=(1, 2, 2), padding=(0, 1, 1), bias=False)
        self.bn1 = hnn.BatchNorm3d(32)
        self.relu = hnn.ReLU(inplace=True)

        self.conv2 = hnn.Conv3d(32, 64, (1, 3, 3), stride=1, padding=(0, 1, 1), bias=False)
        self.bn2 = hnn.BatchNorm3d(64)

        self.block1 = Block(64, 128, reps=2, strides=2, start_with_relu=False, grow_first=True)
        self.block2 = Block(128, 256, reps=2, strides=2, start_with_relu=False, grow_first=True)
        self.block3 = Block(256, 728, reps=2, strides=2, start_with_relu=False, grow_first=True)

        self.block4 = Block(728, 728, reps=3, strides=1, start_with_relu=True, grow_first=True)
        self.block5 = Block(728, 728, reps=3, strides=1, start_with_relu=True, grow_first=True)
        self.block6 = Block(728, 728, reps=3, strides=1, start_with_relu=True, grow_first=True)
        self.block7 = Block(728, 728, reps=3, strides=1, start_with_relu=True, grow_first=True)

        self.block8 = Block(728, 728, reps=3, strides=1, start_with_relu=True, grow_first=True)
        self.block9 = Block(728, 728, reps=3, strides=1, start_with_relu=True, grow_first=True)
        self.block10 = Block(728, 728, reps=3, strides=1, start_with_relu=True, grow_first=True)
        self.block11 = Block(728, 728, reps=3, strides=1, start_with_relu=True, grow_first=True)

        self.block12 = Block(728, 1024, reps=2, strides=1, start_with_relu=True, grow_first=False)

        self.conv3 = SepaConv(1024, 1536)
        self.bn3 = hnn.BatchNorm3d(1536)

        self.conv4 = SepaConv(1536, 2048)
        self.bn4 = hnn.BatchNorm3d(2048)

        # Decoder
        self.uconv1 = DeConv(2048, 1024)
        self.uconv2 = DeConv(1024, 256)
        self.uconv3 = DeConv(256, 128)
        self.uconv4 = DeConv(128, 32)

        self.conv_last = hnn.Conv3d(32, 1, (1, 3, 3), stride=1, padding=(0, 1, 1))
        self.sigmoid = hnn.Sigmoid()

    def forward(self, x):
        plain = not isinstance(x, ClTensor)
        if plain:
            x = F.to_cl(x)
        x = hnn.run_fused([self.conv1, self.bn1, self.relu, self.conv2, self.bn2, self.relu], x)
        for i in range(1, 13):
            x = getattr(self, "block%d" % i)(x)
        x = hnn.run_fused([self.bn3, self.relu], self.conv3(x))
        x = hnn.run_fused([self.bn4, self.relu], self.conv4(x))
        x = self.uconv4(self.uconv3(self.uconv2(self.uconv1(x))))
        predict = self.conv_last(x, act=_lib.ACT_SIGMOID)
        return predict.to_torch() if plain else predict
