"""GANomaly nets + step, CPU float32 restatement of reference models/ganomaly.py (nets :24-175, step :459-519).

Generalised pyramid for non-power-of-two frame sizes exactly as vfd_gan_amd/models/ganomaly.py documents; at
powers of two the module graph, child names and state_dict keys are the reference's."""
import types

import torch
import torch.nn as nn

from .losses import l2_loss


def pyramid_floor(isize):
    csize = isize // 2
    while csize >= 8 and csize % 2 == 0:
        csize //= 2
    return csize


class Encoder(nn.Module):
    def __init__(self, isize, nz, nc, ndf, ngpu, n_extra_layers=0, add_final_conv=True):   # :29
        super().__init__()
        assert isize % 16 == 0
        main = nn.Sequential()
        main.add_module('initial-conv-{0}-{1}'.format(nc, ndf), nn.Conv2d(nc, ndf, 4, 2, 1, bias=False))      # :36-37
        main.add_module('initial-relu-{0}'.format(ndf), nn.LeakyReLU(0.2, inplace=True))                      # :38-39
        csize, cndf = isize // 2, ndf
        for t in range(n_extra_layers):                                                                       # :43-49
            main.add_module('extra-layers-{0}-{1}-conv'.format(t, cndf), nn.Conv2d(cndf, cndf, 3, 1, 1, bias=False))
            main.add_module('extra-layers-{0}-{1}-batchnorm'.format(t, cndf), nn.BatchNorm2d(cndf))
            main.add_module('extra-layers-{0}-{1}-relu'.format(t, cndf), nn.LeakyReLU(0.2, inplace=True))
        floor = pyramid_floor(isize)
        while csize > floor:                                                                                  # :51-61
            main.add_module('pyramid-{0}-{1}-conv'.format(cndf, cndf * 2), nn.Conv2d(cndf, cndf * 2, 4, 2, 1, bias=False))
            main.add_module('pyramid-{0}-batchnorm'.format(cndf * 2), nn.BatchNorm2d(cndf * 2))
            main.add_module('pyramid-{0}-relu'.format(cndf * 2), nn.LeakyReLU(0.2, inplace=True))
            cndf, csize = cndf * 2, csize // 2
        if add_final_conv:                                                                                    # :64-66
            main.add_module('final-{0}-{1}-conv'.format(cndf, 1), nn.Conv2d(cndf, nz, csize, 1, 0, bias=False))
        self.main = main

    def forward(self, input):
        return self.main(input)


class Decoder(nn.Module):
    def __init__(self, isize, nz, nc, ngf, ngpu, n_extra_layers=0):                                           # :83
        super().__init__()
        assert isize % 16 == 0
        floor = pyramid_floor(isize)
        cngf, tisize = ngf // 2, floor                                                                        # :88-91
        while tisize != isize:
            cngf, tisize = cngf * 2, tisize * 2
        main = nn.Sequential()
        main.add_module('initial-{0}-{1}-convt'.format(nz, cngf), nn.ConvTranspose2d(nz, cngf, floor, 1, 0, bias=False))
        main.add_module('initial-{0}-batchnorm'.format(cngf), nn.BatchNorm2d(cngf))
        main.add_module('initial-{0}-relu'.format(cngf), nn.ReLU(True))
        csize = floor
        while csize < isize // 2:                                                                             # :102-111
            main.add_module('pyramid-{0}-{1}-convt'.format(cngf, cngf // 2), nn.ConvTranspose2d(cngf, cngf // 2, 4, 2, 1, bias=False))
            main.add_module('pyramid-{0}-batchnorm'.format(cngf // 2), nn.BatchNorm2d(cngf // 2))
            main.add_module('pyramid-{0}-relu'.format(cngf // 2), nn.ReLU(True))
            cngf, csize = cngf // 2, csize * 2
        for t in range(n_extra_layers):                                                                       # :114-120
            main.add_module('extra-layers-{0}-{1}-conv'.format(t, cngf), nn.Conv2d(cngf, cngf, 3, 1, 1, bias=False))
            main.add_module('extra-layers-{0}-{1}-batchnorm'.format(t, cngf), nn.BatchNorm2d(cngf))
            main.add_module('extra-layers-{0}-{1}-relu'.format(t, cngf), nn.ReLU(True))
        main.add_module('final-{0}-{1}-convt'.format(cngf, nc), nn.ConvTranspose2d(cngf, nc, 4, 2, 1, bias=False))  # :122-125
        main.add_module('final-{0}-tanh'.format(nc), nn.Tanh())
        self.main = main

    def forward(self, input):
        return self.main(input)


class NetD(nn.Module):
    def __init__(self, opt):                                                                                  # :142-149
        super().__init__()
        model = Encoder(opt.isize, 1, opt.nc, opt.ngf, opt.ngpu, opt.extralayers)
        layers = list(model.main.children())
        self.features = nn.Sequential(*layers[:-1])
        self.classifier = nn.Sequential(layers[-1])
        self.classifier.add_module('Sigmoid', nn.Sigmoid())

    def forward(self, x):                                                                                     # :151-157
        features = self.features(x)
        classifier = self.classifier(features).view(-1, 1).squeeze(1)
        return classifier, features


class NetG(nn.Module):
    def __init__(self, opt):                                                                                  # :165-169
        super().__init__()
        self.encoder1 = Encoder(opt.isize, opt.nz, opt.nc, opt.ngf, opt.ngpu, opt.extralayers)
        self.decoder = Decoder(opt.isize, opt.nz, opt.nc, opt.ngf, opt.ngpu, opt.extralayers)
        self.encoder2 = Encoder(opt.isize, opt.nz, opt.nc, opt.ngf, opt.ngpu, opt.extralayers)

    def forward(self, x):                                                                                     # :171-175
        latent_i = self.encoder1(x)
        gen_imag = self.decoder(latent_i)
        latent_o = self.encoder2(gen_imag)
        return gen_imag, latent_i, latent_o


DEFAULTS = dict(nz=100, ngf=64, nc=3, ngpu=1, extralayers=0, w_adv=1.0, w_con=50.0, w_enc=1.0, lr=2e-4, beta1=0.5)


def make_opt(**over):
    d = dict(DEFAULTS)
    d.update(over)
    return types.SimpleNamespace(**d)


def fold_frames(clip):
    """(B,C,T,H,W) -> (B*T,C,H,W)."""
    b, c, t, h, w = clip.shape
    return clip.permute(0, 2, 1, 3, 4).reshape(b * t, c, h, w)


def make_optimizers(netg, netd, opt):
    """:455-456"""
    return (torch.optim.Adam(netg.parameters(), lr=opt.lr, betas=(opt.beta1, 0.999)),
            torch.optim.Adam(netd.parameters(), lr=opt.lr, betas=(opt.beta1, 0.999)))


def step(netg, netd, opt_g, opt_d, x, opt, l2=l2_loss):
    """One optimize_params() of reference models/ganomaly.py:502-519 on frames `x` (N,3,S,S).
    Works on any module pair with the NetG / NetD call signatures (the reference's own classes in the fixture
    generator, this file's classes in the tests).  Returns the loss scalars."""
    l_adv, l_con, l_enc, l_bce = l2, nn.L1Loss(), l2, nn.BCELoss()                # :437-440
    real_label = torch.ones(x.shape[0])
    fake_label = torch.zeros(x.shape[0])
    fake, latent_i, latent_o = netg(x)                                           # forward_g :459-462
    pred_real, feat_real = netd(x)                                               # forward_d :465-469
    pred_fake, feat_fake = netd(fake.detach())
    opt_g.zero_grad()                                                            # :509
    err_g_adv = l_adv(netd(x)[1], netd(fake)[1])                                 # backward_g :472-481
    err_g_con = l_con(fake, x)
    err_g_enc = l_enc(latent_o, latent_i)
    err_g = err_g_adv * opt.w_adv + err_g_con * opt.w_con + err_g_enc * opt.w_enc
    err_g.backward(retain_graph=True)
    opt_g.step()                                                                 # :511
    opt_d.zero_grad()                                                            # :514
    err_d_real = l_bce(pred_real, real_label)                                    # backward_d :484-493
    err_d_fake = l_bce(pred_fake, fake_label)
    err_d = (err_d_real + err_d_fake) * 0.5
    err_d.backward()
    opt_d.step()                                                                 # :516
    return {"err_g": err_g.item(), "err_g_adv": err_g_adv.item(), "err_g_con": err_g_con.item(),
            "err_g_enc": err_g_enc.item(), "err_d": err_d.item(), "err_d_real": err_d_real.item(),
            "err_d_fake": err_d_fake.item()}, fake.detach()
