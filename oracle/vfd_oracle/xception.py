"""Per-frame Xception encoder + up-sampling decoder baseline, CPU float32 restatement of reference models/xception.py
(SepaConv :6-21, Block :23-71, DeConv :73-88, Xception :92-174).  Attribute names (= state_dict keys) are the reference's."""
import torch.nn as nn


class SepaConv(nn.Module):
    """:6-21  (1,3,3) conv -> ReLU -> 1x1x1 conv -> ReLU, no biases (a FULL (1,3,3) conv, not depthwise: groups=1 as written)."""

    def __init__(self, in_ch, out_ch):
        super().__init__()
        self.conv1 = nn.Conv3d(in_ch, in_ch, (1, 3, 3), stride=1, padding=(0, 1, 1), dilation=1, bias=False)
        self.pointwise = nn.Conv3d(in_ch, out_ch, (1, 1, 1), stride=1, padding=(0, 0, 0), dilation=1, bias=False)
        self.relu = nn.ReLU()

    def forward(self, x):
        return self.relu(self.pointwise(self.relu(self.conv1(x))))


class Block(nn.Module):
    """:23-71  `reps` x [ReLU, SepaConv, BN] (+ MaxPool3d((1,3,3), (1,s,s), (0,1,1)) when strided) plus a 1x1x1-conv + BN
    shortcut where the shape changes; the two are added."""

    def __init__(self, in_fi, out_fi, reps, strides=1, start_with_relu=True, grow_first=True):
        super().__init__()
        if out_fi != in_fi or strides != 1:                                                              # :27-31
            self.skip = nn.Conv3d(in_fi, out_fi, 1, stride=(1, strides, strides), bias=False)
            self.skipbn = nn.BatchNorm3d(out_fi)
        else:
            self.skip = None
        self.relu = nn.ReLU(inplace=True)
        rep, filters = [], in_fi
        if grow_first:                                                                                   # :37-41
            rep += [self.relu, SepaConv(in_fi, out_fi), nn.BatchNorm3d(out_fi)]
            filters = out_fi
        for _ in range(reps - 1):                                                                        # :43-46
            rep += [self.relu, SepaConv(filters, filters), nn.BatchNorm3d(filters)]
        if not grow_first:                                                                               # :48-51
            rep += [self.relu, SepaConv(in_fi, out_fi), nn.BatchNorm3d(out_fi)]
        if not start_with_relu:                                                                          # :53-56
            rep = rep[1:]
        else:
            rep[0] = nn.ReLU(inplace=False)
        if strides != 1:                                                                                 # :58-59
            rep.append(nn.MaxPool3d((1, 3, 3), (1, strides, strides), padding=(0, 1, 1)))
        self.rep = nn.Sequential(*rep)

    def forward(self, inp):                                                                              # :61-71
        x = self.rep(inp)
        skip = self.skipbn(self.skip(inp)) if self.skip is not None else inp
        return x + skip


class DeConv(nn.Module):
    """:73-88  (1,3,3) conv -> BN -> LeakyReLU(0.2) -> Dropout(.25) -> trilinear up-sampling of H and W."""

    def __init__(self, in_fi, out_fi):
        super().__init__()
        self.conv = nn.Conv3d(in_fi, out_fi, (1, 3, 3), stride=1, padding=(0, 1, 1), bias=False)
        self.bn = nn.BatchNorm3d(out_fi)
        self.lrelu = nn.LeakyReLU(0.2, inplace=True)
        self.dropout = nn.Dropout(p=0.25)
        self.upsamp = nn.Upsample(scale_factor=(1, 2, 2), mode='trilinear', align_corners=True)

    def forward(self, x):
        return self.upsamp(self.dropout(self.lrelu(self.bn(self.conv(x)))))


class Xception(nn.Module):
    def __init__(self, ich=3):                                                                           # :93-131
        super().__init__()
        self.conv1 = nn.Conv3d(ich, 32, (1, 3, 3), stride=(1, 2, 2), padding=(0, 1, 1), bias=False)
        self.bn1 = nn.BatchNorm3d(32)
        self.relu = nn.ReLU(inplace=True)
        self.conv2 = nn.Conv3d(32, 64, (1, 3, 3), stride=1, padding=(0, 1, 1), bias=False)
        self.bn2 = nn.BatchNorm3d(64)
        self.block1 = Block(64, 128, reps=2, strides=2, start_with_relu=False, grow_first=True)
        self.block2 = Block(128, 256, reps=2, strides=2, start_with_relu=False, grow_first=True)
        self.block3 = Block(256, 728, reps=2, strides=2, start_with_relu=False, grow_first=True)
        for i in range(4, 12):                                                                           # block4 .. block11
            setattr(self, "block%d" % i, Block(728, 728, reps=3, strides=1, start_with_relu=True, grow_first=True))
        self.block12 = Block(728, 1024, reps=2, strides=1, start_with_relu=True, grow_first=False)
        self.conv3 = SepaConv(1024, 1536)
        self.bn3 = nn.BatchNorm3d(1536)
        self.conv4 = SepaConv(1536, 2048)
        self.bn4 = nn.BatchNorm3d(2048)
        self.uconv1 = DeConv(2048, 1024)
        self.uconv2 = DeConv(1024, 256)
        self.uconv3 = DeConv(256, 128)
        self.uconv4 = DeConv(128, 32)
        self.conv_last = nn.Conv3d(32, 1, (1, 3, 3), stride=1, padding=(0, 1, 1))
        self.sigmoid = nn.Sigmoid()

    def forward(self, x):                                                                                # :134-174
        x = self.relu(self.bn1(self.conv1(x)))
        x = self.relu(self.bn2(self.conv2(x)))
        for i in range(1, 13):
            x = getattr(self, "block%d" % i)(x)
        x = self.relu(self.bn3(self.conv3(x)))
        x = self.relu(self.bn4(self.conv4(x)))
        x = self.uconv4(self.uconv3(self.uconv2(self.uconv1(x))))
        return self.sigmoid(self.conv_last(x))
