"""AnoGAN nets and training step on HIP kernels, behind the surface of the reference's models/anogan.py
(NetG :39-79, NetD :81-119, AnoGAN :121-143, optimize_params :229-250).

Generalisation (SURVEY.md sections 0, 8d): the reference hard-wires 16x128x128; here the generator's seed volume is
(512, nfr/8, isize/8, isize/8) and the discriminator's Linear has 256*(nfr/8)*(isize/8)^2 inputs, which at the
defaults (nfr=16, isize=128) are the reference's modules exactly (same state_dict keys and shapes).
"""
import torch
import torch.nn as tnn

from .. import dist as vdist
from .. import functional as F
from .. import nn as hnn
from .. import optim as hoptim
from ..functional import ClTensor
from ..lib.train_gan import GANBaseModel
from ..lib.utils import weights_init


class NetG(tnn.Module):
    def __init__(self, nfr=16, isize=128):
        super(NetG, self).__init__()
        self.seed_shape = (512, nfr // 8, isize // 8, isize // 8)
        feat = 512 * (nfr // 8) * (isize // 8) * (isize // 8)
        self.layer1 = hnn.Sequential(
            hnn.Linear(100, feat),
            hnn.BatchNorm1d(feat),
            hnn.ReLU(),
        )
        self.layer2 = hnn.Sequential(
            hnn.Dropout(p=0.25),
            hnn.ConvTranspose3d(512, 256, 3, 2, 1, 1),
            hnn.Conv3d(256, 256, 3, 1, 1),
            hnn.BatchNorm3d(256),
            hnn.LeakyReLU(),
            hnn.Dropout(p=0.25),
            hnn.ConvTranspose3d(256, 128, 3, 2, 1, 1),
            hnn.Conv3d(128, 128, 3, 1, 1),
            hnn.BatchNorm3d(128),
            hnn.LeakyReLU()
        )
        self.layer3 = hnn.Sequential(
            hnn.Dropout(p=0.25),
            hnn.ConvTranspose3d(128, 64, 3, 1, 1),
            hnn.Conv3d(64, 64, 3, 1, 1),
            hnn.BatchNorm3d(64),
            hnn.LeakyReLU(),
            hnn.Dropout(p=0.25),
            hnn.ConvTranspose3d(64, 3, 3, 2, 1, 1),
            hnn.Conv3d(3, 3, 3, 1, 1),
            hnn.Sigmoid()
        )

    def forward(self, z):
        plain = not isinstance(z, ClTensor)
        x = self.layer1(F.to_cl(z) if plain else z)
        x = F.unflatten(x, self.seed_shape)          # x.view(N, 512, 2, 16, 16) of the reference (:76)
        x = self.layer2(x)
        x = self.layer3(x)
        return x.to_torch() if plain else x


class NetD(tnn.Module):
    def __init__(self, nfr=16, isize=128):
        super(NetD, self).__init__()
        self.layer1 = hnn.Sequential(
            hnn.Conv3d(3, 32, 3, stride=1, padding=1),
            hnn.BatchNorm3d(32),
            hnn.LeakyReLU(),
            hnn.Conv3d(32, 64, 3, stride=1, padding=1),
            hnn.Conv3d(64, 64, 3, stride=1, padding=1),
            hnn.BatchNorm3d(64),
            hnn.LeakyReLU(64),     # slope 64, exactly as the reference writes it (:91)
            hnn.AvgPool3d(2)
        )
        self.layer2 = hnn.Sequential(
            hnn.Conv3d(64, 128, 3, stride=1, padding=1),
            hnn.Conv3d(128, 128, 3, stride=1, padding=1),
            hnn.BatchNorm3d(128),
            hnn.LeakyReLU(),
            hnn.AvgPool3d(2),
            hnn.Conv3d(128, 256, 3, stride=1, padding=1),
            hnn.BatchNorm3d(256),
            hnn.LeakyReLU(),
            hnn.AvgPool3d(2)
        )
        self.fc = hnn.Sequential(
            hnn.Linear(256 * (nfr // 8) * (isize // 8) * (isize // 8), 1),
            hnn.Sigmoid()
        )

    def forward(self, x):
        plain = not isinstance(x, ClTensor)
        x = self.layer1(F.to_cl(x) if plain else x)
        x = self.layer2(x)
        feature = x                      # the reference returns x.view(N, -1); the Linear below takes the block
        out = self.fc(x)                 # Linear over the flattened (C,D,H,W) features + Sigmoid, one kernel
        if plain:
            return out.to_torch(), F.flatten(feature).to_torch()
        return out, feature


class AnoGAN(GANBaseModel):
    def __init__(self, args, dataloader):
        super(AnoGAN, self).__init__(args, dataloader)
        self.netg = NetG(args.nfr, args.isize).to(self.device)
        self.netd = NetD(args.nfr, args.isize).to(self.device)
        self.netg.apply(weights_init)
        self.netd.apply(weights_init)
        vdist.broadcast_module(self.netg)
        vdist.broadcast_module(self.netd)
        self.netg.train()
        self.netd.train()

        self.loss = F.bce_loss
        # reference :139-140 — the generator runs at 5*lr and betas are hard-coded (args.beta1 is ignored)
        self.g_opt = hoptim.Adam(self.netg.parameters(), lr=5 * args.lr, betas=(0.5, 0.999))
        self.d_opt = hoptim.Adam(self.netd.parameters(), lr=args.lr, betas=(0.5, 0.999))
        self.reducer_g = vdist.GradReducer.for_optimizer(self.g_opt)
        self.reducer_d = vdist.GradReducer.for_optimizer(self.d_opt)
        self.ones_label, self.zeros_label = 1.0, 0.0
        torch.cuda.manual_seed(4321 + 7919 * self.rank)     # per-rank noise streams
        self.z = None                                       # tests may impose the noise
        if self.load_pretrained():           # --resume (reference :133-143); last, so that a checkpointed RNG state survives
            vdist.broadcast_module(self.netg)
            vdist.broadcast_module(self.netd)

    def set_input(self, data):
        super(AnoGAN, self).set_input(data)
        self.real_cl = F.to_cl(self.real)

    # ---- the step, reference :229-250, in the three pieces data-parallel graph capture needs ------------------------
    def _d_phase(self, join=True):
        """NetD on real and on G(z).detach() (reference :231-241): TWO backward passes deposit into netD's gradients,
        so its reducer is armed for two (a bucket is reduced once both passes have written it)."""
        F.dropout_begin_step(self.device)
        self.d_opt.zero_grad()
        self.reducer_d.arm(passes=2)
        dis_real = self.netd(self.real_cl)[0]
        self.dis_loss_real = self.loss(dis_real, self.ones_label)
        self.dis_loss_real.backward()

        # torch's default device generator is graph-safe (its Philox offset advances under hipGraph replay)
        z = self.z if self.z is not None else torch.randn(self.args.batchsize, 100, device=self.device)
        self.gen_fake_t = self.netg(F.to_cl(z))
        dis_fake = self.netd(self.gen_fake_t.detach())[0]
        self.dis_loss_fake = self.loss(dis_fake, self.zeros_label)
        self.dis_loss_fake.backward()
        self.dis_loss = self.dis_loss_real + self.dis_loss_fake
        if join:
            self.reducer_d.finish()

    def _g_phase(self, join=True):
        """NetG (reference :246-250).  netD's own gradients from this backward are never used by the reference
        (the next step starts with netd.zero_grad()), so netD is frozen here and only the data gradient flows."""
        self.g_opt.zero_grad()
        for p in self.netd.parameters():
            p.requires_grad_(False)
        self.reducer_d.enabled = False
        try:
            dis_fake = self.netd(self.gen_fake_t)[0]
            self.gen_loss = self.loss(dis_fake, self.ones_label)
            self.gen_loss.backward()
        finally:
            for p in self.netd.parameters():
                p.requires_grad_(True)
            self.reducer_d.enabled = True
        if join:
            self.reducer_g.finish()

    def _publish(self):
        self.gen_fake = self.gen_fake_t.detach()
        self.errors_dict.update({'d/err_d/train': self.dis_loss, 'g/err_g/train': self.gen_loss,
                                 'd/err_d_real/train': self.dis_loss_real, 'd/err_d_fake/train': self.dis_loss_fake})

    def test(self):
        """In-loop evaluation sweep, reference :145-227: both nets in EVAL mode (running BatchNorm statistics, no Dropout) under
        torch.no_grad(); per test batch the three discriminator losses, predict = grey(normalise(|G(z) - real|))
        (predict_forg :24-37, on the device), threshold + 5x5 opening for the summaries; ROC / PR / F1 of `predict` against
        `gt` over all test pixels, a checkpoint when ROC (else PR) improves.  The reference leaves the nets in eval mode
        afterwards and its optimize_params never switches back (models/anogan.py has no .train() call in the step); here
        training mode is restored, as the per-step BatchNorm statistics of the training step require."""
        import numpy as np
        from ..lib.evaluate import evaluate
        from ..lib.utils import morphology_proc, predict_forg, threshold
        was_training = self.netg.training
        self.netg.eval()
        self.netd.eval()
        gen_loss_, dis_loss_real_, dis_loss_fake_ = [], [], []
        gts, predicts = [], []
        try:
            with torch.no_grad():
                for i, data in enumerate(self.dataloader['test']):
                    input, real, gt, lb = (d.to(self.device, non_blocking=True) for d in data)
                    real_cl = F.to_cl(real)
                    dis_loss_real_.append(self.loss(self.netd(real_cl)[0], self.ones_label))
                    z = self.z if self.z is not None else torch.randn(self.args.batchsize, 100, device=self.device)
                    gen_fake_cl = self.netg(F.to_cl(z))
                    dis_fake_ = self.netd(gen_fake_cl)[0]
                    dis_loss_fake_.append(self.loss(dis_fake_, self.zeros_label))
                    gen_loss_.append(self.loss(dis_fake_, self.ones_label))        # reference runs netd(gen_fake_) a second time: same values
                    gen_fake_ = gen_fake_cl.to_torch()
                    predict_ = predict_forg(gen_fake_, real)
                    t_pre_ = threshold(predict_)
                    m_pre_ = morphology_proc(t_pre_)
                    gts.append(gt.permute(0, 2, 3, 4, 1))
                    predicts.append(predict_.permute(0, 2, 3, 4, 1))
                    self.color_video_dict.update({'test/input-real-gen': torch.cat([input, real, gen_fake_], dim=3)})
                    self.gray_video_dict.update({'test/gt-pre-th-morph': torch.cat([gt, predict_, t_pre_, m_pre_], dim=3)})
                    self.hist_dict.update({"test/inp": input, "test/gt": gt, "test/gen": gen_fake_, "test/predict": predict_,
                                           "test/t_pre": t_pre_, "test/m_pre": m_pre_})
                tonp = lambda xs: torch.stack([v.detach().float().reshape(()) for v in xs]).cpu().numpy().astype(np.float64)  # noqa: E731
                gen_l, dr, df = tonp(gen_loss_), tonp(dis_loss_real_), tonp(dis_loss_fake_)
                gts_np = np.asarray(torch.stack(gts).cpu().numpy(), dtype=np.int32).flatten()
                pre_np = np.asarray(torch.stack(predicts).cpu().numpy()).flatten()
        finally:
            if was_training:
                self.netg.train()
                self.netd.train()
        saveto = self.save_root_dir if self.rank == 0 else None
        roc = evaluate(gts_np, pre_np, self.best_roc, self.epoch, saveto, metric='roc')
        pr = evaluate(gts_np, pre_np, self.best_pr, self.epoch, saveto, metric='pr')
        f1 = evaluate(gts_np, pre_np, metric='f1_score')
        if roc > self.best_roc:
            self.best_roc = roc
            self.save_weights('roc')
        elif pr > self.best_pr:
            self.best_pr = pr
            self.save_weights('pr')
        self.score_dict.update({"score/roc": roc, "score/pr": pr, "score/f1": f1})
        # the reference files the generator loss under 'd/err_d/test' and the discriminator loss under 'd/err_g/test' (:224-225)
        self.errors_dict.update({'d/err_d/test': float(np.mean(gen_l)), 'd/err_g/test': float(np.mean(dr + df))})
        self.test_losses = {"gen_loss": float(np.mean(gen_l)), "dis_loss_real": float(np.mean(dr)), "dis_loss_fake": float(np.mean(df))}
        return {"roc": roc, "pr": pr, "f1": f1}

    def optimize_params(self):
        self._d_phase()
        self.d_opt.step()
        self._g_phase()
        self.g_opt.step()
        self._publish()

    def step_program(self):
        """Graph capture under data parallelism (vfd_gan_amd.graph.GraphedStep).  The generator phase reads the
        UPDATED discriminator, so netD's reduction cannot be deferred; netG's is joined right before its Adam update."""
        def a():
            self._d_phase(join=False)

        def b():
            self.d_opt.step()
            self._g_phase(join=False)

        def c():
            self.g_opt.step()
            self._publish()
        return [("graph", a), ("reduce", self.reducer_d), ("join", self.reducer_d), ("graph", b),
                ("reduce", self.reducer_g), ("join", self.reducer_g), ("graph", c)]
