"""MyGAN nets + step, CPU float32 restatement of reference models/mygannet.py (nets :13-213, step :275-367).

Generalisation: SDisc.linear / TDisc.linear in-features are computed from (nfr, isize) instead of the
hard-wired nfr=16, isize=128 (:134, :176); at 16x128x128 they equal the reference's (4096 and 256)."""
import types

import torch
import torch.nn as nn

from .losses import gray2rgb, l2_loss, weighted_bce
from .spatiotempconv import SpatioTemporalConv


class NetgConv(nn.Module):
    def __init__(self, in_fi, out_fi, kernel_size=3):                                    # :14-20
        super().__init__()
        self.conv = SpatioTemporalConv(in_fi, out_fi, kernel_size, padding=kernel_size // 2)
        self.bn = nn.BatchNorm3d(out_fi)
        self.lrelu = nn.LeakyReLU(0.2, inplace=True)

    def forward(self, x):
        return self.lrelu(self.bn(self.conv(x)))


class NetG(nn.Module):
    def __init__(self, nc=3, ngf=32):                                                    # :32-53
        super().__init__()
        self.dconv1 = NetgConv(nc, ngf)
        self.dconv2 = NetgConv(ngf, ngf * 2)
        self.dconv3 = NetgConv(ngf * 2, ngf * 4)
        self.dconv4 = NetgConv(ngf * 4, ngf * 8)
        self.dconv5 = NetgConv(ngf * 8, ngf * 16)
        self.avgpool = nn.AvgPool3d(2)
        self.uconv5 = NetgConv(ngf * 16, ngf * 8)
        self.uconv4 = NetgConv(ngf * 8 + ngf * 8, ngf * 8)
        self.uconv3 = NetgConv(ngf * 8 + ngf * 4, ngf * 4)
        self.uconv2 = NetgConv(ngf * 4 + ngf * 2, ngf * 2)
        self.uconv1 = NetgConv(ngf * 2 + ngf, ngf)
        self.dropout = nn.Dropout(p=0.25)
        self.upsamp = nn.Upsample(scale_factor=2, mode='trilinear', align_corners=True)
        self.conv_last = nn.Conv3d(ngf, 1, 3, stride=1, padding=1, bias=False)
        self.sigmoid = nn.Sigmoid()

    def forward(self, x):                                                                # :55-101
        dconv1 = self.dconv1(x)
        dconv2 = self.dconv2(self.avgpool(dconv1))
        dconv3 = self.dconv3(self.avgpool(dconv2))
        dconv4 = self.dconv4(self.avgpool(dconv3))
        latent_i = self.dconv5(self.avgpool(dconv4))
        x = self.upsamp(self.dropout(self.uconv5(latent_i)))
        x = self.upsamp(self.dropout(self.uconv4(torch.cat([x, dconv4], dim=1))))
        x = self.upsamp(self.dropout(self.uconv3(torch.cat([x, dconv3], dim=1))))
        x = self.upsamp(self.dropout(self.uconv2(torch.cat([x, dconv2], dim=1))))
        x = self.uconv1(torch.cat([x, dconv1], dim=1))
        return self.sigmoid(self.conv_last(x))


class NetdConv(nn.Module):
    def __init__(self, in_fi, out_fi, kernel_size=None, padding=None):                   # :105-110
        super().__init__()
        self.conv = SpatioTemporalConv(in_fi, out_fi, kernel_size, padding=padding)
        self.bn = nn.BatchNorm3d(out_fi)
        self.lrelu = nn.LeakyReLU()

    def forward(self, x):
        return self.lrelu(self.bn(self.conv(x)))


class SDisc(nn.Module):
    def __init__(self, nc, nfr, ndf=32, kernel=None, padding=None, isize=128):           # :120-135
        super().__init__()
        mk = lambda i, o: NetdConv(i, o, kernel_size=kernel, padding=padding)
        self.dconv1, self.dconv2, self.dconv3 = mk(nc, ndf), mk(ndf, ndf * 2), mk(ndf * 2, ndf * 4)
        self.dconv4, self.dconv5, self.dconv6 = mk(ndf * 4, ndf * 8), mk(ndf * 8, ndf * 16), mk(ndf * 16, ndf * 32)
        self.avgpool = nn.AvgPool3d((1, 2, 2))
        self.gpool = nn.AvgPool3d((nfr, 1, 1), stride=1)
        self.linear = nn.Linear(ndf * 32 * (isize // 64) * (isize // 64), 1)
        self.sigmoid = nn.Sigmoid()

    def forward(self, x):                                                                # :138-162
        for conv in (self.dconv1, self.dconv2, self.dconv3, self.dconv4, self.dconv5, self.dconv6):
            x = self.avgpool(conv(x))
        features = x
        x = self.gpool(features)
        classifier = self.sigmoid(self.linear(x.view(x.shape[0], -1)))
        return classifier.squeeze(1), features


class TDisc(nn.Module):
    def __init__(self, nc, isize, ndf=32, kernel=None, padding=None, nfr=16):            # :165-177
        super().__init__()
        mk = lambda i, o: NetdConv(i, o, kernel_size=kernel, padding=padding)
        self.dconv1, self.dconv2, self.dconv3 = mk(nc, ndf), mk(ndf, ndf * 2), mk(ndf * 2, ndf * 4)
        self.avgpool = nn.AvgPool3d((2, 1, 1))
        self.gpool = nn.AvgPool3d((1, isize, isize), stride=1)
        self.linear = nn.Linear(ndf * 4 * (nfr // 8), 1)
        self.sigmoid = nn.Sigmoid()

    def forward(self, x):                                                                # :180-196
        for conv in (self.dconv1, self.dconv2, self.dconv3):
            x = self.avgpool(conv(x))
        features = x
        x = self.gpool(features)
        classifier = self.sigmoid(self.linear(x.view(x.shape[0], -1)))
        return classifier.squeeze(1), features


class NetD(nn.Module):
    def __init__(self, args):                                                            # :201-206
        super().__init__()
        self.spatdisc = SDisc(3, args.nfr, kernel=(1, 3, 3), padding=(0, 1, 1), isize=args.isize)
        self.tempdisc = TDisc(3, args.isize, kernel=(3, 1, 1), padding=(1, 0, 0), nfr=args.nfr)

    def forward(self, x, y):                                                             # :208-213
        s_cls, s_feat = self.spatdisc(x)
        t_cls, t_feat = self.tempdisc(y)
        return s_cls, s_feat, t_cls, t_feat


def make_optimizers(netg, netd, lr=2e-5, beta1=0.5):
    """:270-273"""
    return (torch.optim.Adam(netg.parameters(), lr=lr, betas=(beta1, 0.999)),
            torch.optim.Adam(netd.parameters(), lr=lr, betas=(beta1, 0.999)))


def step(netg, netd, opt_g, opt_d, inp, gt, gt_flow, pre_flow, w_adv=1, w_con=10, fns=None):
    """One optimize_params() of reference models/mygannet.py:350-367.  `gt_flow` / `pre_flow` stand where
    video_to_flow (CPU Farneback, lib/utils.py:94-129, out of scope) feeds NetD at :281-286."""
    l_bce = nn.BCELoss()
    l2_loss_, weighted_bce_, gray2rgb_ = (fns.l2_loss, fns.weighted_bce, fns.gray2rgb) if fns else (l2_loss, weighted_bce, gray2rgb)
    b = inp.shape[0]
    real_label, gout_label = torch.ones(b), torch.zeros(b)
    netg.train(); netd.train()                                                           # :352-353
    predict = netg(inp)                                                                  # forward_g :275-276
    pre_3ch, gt_3ch = gray2rgb_(predict.detach()), gray2rgb_(gt.detach())                  # forward_d :279-286
    s_pr, s_fr, t_pr, t_fr = netd(gt_3ch, gt_flow.detach())
    s_pf, s_ff, t_pf, t_ff = netd(pre_3ch.detach(), pre_flow.detach())
    opt_g.zero_grad()                                                                    # :359
    err_g_adv_s, err_g_adv_t = l2_loss_(s_fr, s_ff), l2_loss_(t_fr, t_ff)                  # backward_g :305-312
    err_g_adv = err_g_adv_s + err_g_adv_t
    err_g_con = weighted_bce_(predict, gt)                                               # pos_weight=2 always (:265-266)
    err_g = err_g_adv * w_adv + err_g_con * w_con
    err_g.backward(retain_graph=True)
    opt_g.step()                                                                         # :361
    opt_d.zero_grad()                                                                    # :364
    e_rs, e_rt = l_bce(s_pr, real_label), l_bce(t_pr, real_label)                        # backward_d :323-331
    e_fs, e_ft = l_bce(s_pf, gout_label), l_bce(t_pf, gout_label)
    err_d_real, err_d_fake = (e_rs + e_rt) * 0.5, (e_fs + e_ft) * 0.5
    err_d = (err_d_real + err_d_fake) * 0.5
    err_d.backward()
    opt_d.step()                                                                         # :366
    return {"err_g": err_g.item(), "err_g_adv": err_g_adv.item(), "err_g_adv_s": err_g_adv_s.item(),
            "err_g_adv_t": err_g_adv_t.item(), "err_g_con": err_g_con.item(), "err_d_real_s": e_rs.item(),
            "err_d_real_t": e_rt.item(), "err_d_fake_s": e_fs.item(), "err_d_fake_t": e_ft.item(),
            "err_d_real": err_d_real.item(), "err_d_fake": err_d_fake.item(), "err_d": err_d.item()}, predict.detach()


def make_args(nfr=16, isize=128):
    return types.SimpleNamespace(nfr=nfr, isize=isize)
