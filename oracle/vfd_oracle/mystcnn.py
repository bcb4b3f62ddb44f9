"""(2+1)D auto-encoder baseline, CPU float32 restatement of reference models/mystcnn.py (C2plus1d_Block :6-49,
AutoEncoder :52-88).  Attribute names (= state_dict keys) are the reference's; SURVEY.md section 8f N4."""
import torch
import torch.nn as nn


class C2plus1d_Block(nn.Module):
    """:6-49.  Main branch: (1,3,3) conv -> BN -> ReLU -> (3,1,1) conv -> BN -> ReLU; shortcut: 1x1x1 conv of the block input;
    both resampled (down: AvgPool3d(2); up: trilinear x2, the shortcut through Dropout first), concatenated, 3x3x3 conv."""

    def __init__(self, in_ch, out_ch, k=5):
        super().__init__()
        self.conv = nn.Conv3d(in_ch, out_ch, 1, stride=1)                                                  # :10
        self.spaceconv = nn.Conv3d(in_ch, in_ch, (1, 3, 3), stride=1, padding=(0, 1, 1), bias=False)       # :12
        self.pointwise = nn.Conv3d(in_ch, out_ch, (3, 1, 1), stride=1, padding=(1, 0, 0), bias=False)      # :13
        self.bn1 = nn.BatchNorm3d(in_ch)                                                                   # :15-16
        self.bn2 = nn.BatchNorm3d(out_ch)
        self.avgpool = nn.AvgPool3d(2)                                                                     # :18-20
        self.dropout = nn.Dropout(p=0.25)
        self.upsamp = nn.Upsample(scale_factor=2, mode='trilinear', align_corners=True)
        self.relu = nn.ReLU(inplace=True)                                                                  # :22
        self.conv_last = nn.Conv3d(out_ch + out_ch, out_ch, 3, stride=1, padding=1, bias=False)            # :23

    def forward(self, x, down_samp=False):                                                                 # :25-49
        inp = x
        x = self.relu(self.bn1(self.spaceconv(x)))
        x = self.relu(self.bn2(self.pointwise(x)))
        if down_samp:
            x = self.avgpool(x)
            inp = self.avgpool(self.conv(inp))
        else:
            x = self.upsamp(x)
            inp = self.conv(self.upsamp(self.dropout(inp)))
        return self.conv_last(torch.cat([x, inp], dim=1))


class AutoEncoder(nn.Module):
    def __init__(self):                                                                                    # :53-67
        super().__init__()
        self.down_sep1 = C2plus1d_Block(3, 64)
        self.down_sep2 = C2plus1d_Block(64, 128)
        self.down_sep3 = C2plus1d_Block(128, 256)
        self.down_sep4 = C2plus1d_Block(256, 512)
        self.up_sep1 = C2plus1d_Block(512, 256)
        self.up_sep2 = C2plus1d_Block(256 + 256, 256)
        self.up_sep3 = C2plus1d_Block(256 + 128, 128)
        self.up_sep4 = C2plus1d_Block(128 + 64, 64)
        self.conv_last = nn.Conv3d(64, 1, 3, stride=1, padding=1, bias=False)
        self.sigmoid = nn.Sigmoid()

    def forward(self, x):                                                                                  # :69-88
        d1 = self.down_sep1(x, down_samp=True)
        d2 = self.down_sep2(d1, down_samp=True)
        d3 = self.down_sep3(d2, down_samp=True)
        d4 = self.down_sep4(d3, down_samp=True)
        u = self.up_sep1(d4, down_samp=False)
        u = self.up_sep2(torch.cat([u, d3], dim=1), down_samp=False)
        u = self.up_sep3(torch.cat([u, d2], dim=1), down_samp=False)
        u = self.up_sep4(torch.cat([u, d1], dim=1), down_samp=False)
        return self.sigmoid(self.conv_last(u))


def step(model, opt, inp, gt):
    """One training step of reference lib/train_stcnn.py:103-108: BCELoss(model(input), gt), backward, Adam."""
    opt.zero_grad()
    predict = model(inp)
    err = nn.BCELoss()(predict, gt)
    err.backward()
    opt.step()
    return {"err": err.item()}, predict.detach()


def make_optimizer(model, lr=2e-5, beta1=0.5):
    """lib/train_stcnn.py:91"""
    return torch.optim.Adam(model.parameters(), lr=lr, betas=(beta1, 0.999))
