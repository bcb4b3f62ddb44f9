"""AnoGAN nets + step, CPU float32 restatement of reference models/anogan.py (NetG :39-79, NetD :81-119,
step :229-250).  Generalised to (nfr, isize) multiples of 8 (seed volume (512, nfr/8, isize/8, isize/8) and
Linear(256*(nfr/8)*(isize/8)^2, 1)); at the reference's fixed 16x128x128 the modules are identical."""
import torch
import torch.nn as nn


class NetG(nn.Module):
    def __init__(self, nfr=16, isize=128):
        super().__init__()
        self.seed_shape = (512, nfr // 8, isize // 8, isize // 8)
        feat = 512 * (nfr // 8) * (isize // 8) * (isize // 8)
        self.layer1 = nn.Sequential(nn.Linear(100, feat), nn.BatchNorm1d(feat), nn.ReLU())                  # :43-47
        self.layer2 = nn.Sequential(                                                                        # :49-60
            nn.Dropout(p=0.25), nn.ConvTranspose3d(512, 256, 3, 2, 1, 1), nn.Conv3d(256, 256, 3, 1, 1),
            nn.BatchNorm3d(256), nn.LeakyReLU(),
            nn.Dropout(p=0.25), nn.ConvTranspose3d(256, 128, 3, 2, 1, 1), nn.Conv3d(128, 128, 3, 1, 1),
            nn.BatchNorm3d(128), nn.LeakyReLU())
        self.layer3 = nn.Sequential(                                                                        # :62-72
            nn.Dropout(p=0.25), nn.ConvTranspose3d(128, 64, 3, 1, 1), nn.Conv3d(64, 64, 3, 1, 1),
            nn.BatchNorm3d(64), nn.LeakyReLU(),
            nn.Dropout(p=0.25), nn.ConvTranspose3d(64, 3, 3, 2, 1, 1), nn.Conv3d(3, 3, 3, 1, 1), nn.Sigmoid())

    def forward(self, z):                                                                                   # :74-79
        x = self.layer1(z)
        x = x.view(x.size()[0], *self.seed_shape)
        return self.layer3(self.layer2(x))


class NetD(nn.Module):
    def __init__(self, nfr=16, isize=128):
        super().__init__()
        self.layer1 = nn.Sequential(                                                                        # :84-93
            nn.Conv3d(3, 32, 3, stride=1, padding=1), nn.BatchNorm3d(32), nn.LeakyReLU(),
            nn.Conv3d(32, 64, 3, stride=1, padding=1), nn.Conv3d(64, 64, 3, stride=1, padding=1),
            nn.BatchNorm3d(64), nn.LeakyReLU(64), nn.AvgPool3d(2))     # LeakyReLU(64): slope 64, as written at :91
        self.layer2 = nn.Sequential(                                                                        # :95-105
            nn.Conv3d(64, 128, 3, stride=1, padding=1), nn.Conv3d(128, 128, 3, stride=1, padding=1),
            nn.BatchNorm3d(128), nn.LeakyReLU(), nn.AvgPool3d(2),
            nn.Conv3d(128, 256, 3, stride=1, padding=1), nn.BatchNorm3d(256), nn.LeakyReLU(), nn.AvgPool3d(2))
        self.fc = nn.Sequential(nn.Linear(256 * (nfr // 8) * (isize // 8) * (isize // 8), 1), nn.Sigmoid())   # :107-110

    def forward(self, x):                                                                                   # :112-119
        x = self.layer2(self.layer1(x))
        x = x.view(x.size()[0], -1)
        return self.fc(x), x


def make_optimizers(netg, netd, lr):
    """:139-140 — G runs at 5*lr, betas are hard-coded (0.5, 0.999)."""
    return (torch.optim.Adam(netg.parameters(), lr=5 * lr, betas=(0.5, 0.999)),
            torch.optim.Adam(netd.parameters(), lr=lr, betas=(0.5, 0.999)))


def step(netg, netd, g_opt, d_opt, real, z):
    """One optimize_params() of reference models/anogan.py:229-250; `z` replaces torch.randn(B,100) at :237."""
    loss = nn.BCELoss()
    b = real.shape[0]
    ones, zeros = torch.ones(b), torch.zeros(b)
    netd.zero_grad()                                                     # :231
    dis_loss_real = loss(netd(real)[0].view(-1), ones)                   # :233-235
    dis_loss_real.backward()
    gen_fake = netg(z)                                                   # :238
    dis_loss_fake = loss(netd(gen_fake.detach())[0].view(-1), zeros)     # :239-241
    dis_loss_fake.backward()
    dis_loss = dis_loss_real + dis_loss_fake
    d_opt.step()                                                         # :243
    netg.zero_grad()                                                     # :246
    gen_loss = loss(netd(gen_fake)[0].view(-1), ones)                    # :247-249
    gen_loss.backward(retain_graph=True)
    g_opt.step()                                                         # :250
    return {"err_d": dis_loss.item(), "err_d_real": dis_loss_real.item(), "err_d_fake": dis_loss_fake.item(),
            "err_g": gen_loss.item()}, gen_fake.detach()
