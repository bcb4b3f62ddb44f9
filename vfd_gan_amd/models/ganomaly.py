"""GANomaly nets and training step on HIP kernels, behind the surface of the reference's models/ganomaly.py.

Reference: Encoder :24-76, Decoder :79-133, NetD :137-157, NetG :160-175, step :459-519.  That file is the upstream
2-D image GANomaly (dead code in the reference: it imports lib.networks / lib.visualizer / lib.loss, which are not
in its tree, and reads options lib/args.py never defines).  Here it is wired to the clip contract of the rest of
the repo: a (B,3,T,H,W) clip is folded to B*T frames (free reshape in channels-last) and the 2-D nets run on them.

Generalisation (SURVEY.md sections 0, 8d): the reference pyramid only works for power-of-two ``isize``
(``Decoder.__init__`` does not terminate at 112 / 224).  Here the pyramid halves while ``csize >= 8`` and the final
encoder conv / initial decoder conv-transpose use kernel = the remaining extent (4 for powers of two — where this
reduces EXACTLY to the reference, same modules, same state_dict keys — and 7 for 112 / 224).
"""
import types

import torch
import torch.nn as tnn

from .. import functional as F
from .. import nn as hnn
from .. import optim as hoptim
from .. import dist as vdist
from ..functional import ClTensor
from ..lib.train_gan import GANBaseModel
from ..lib.utils import weights_init_dcgan


def _pyramid_floor(isize):
    """Spatial extent at which the encoder pyramid stops (4 for powers of two, as in the reference)."""
    csize = isize // 2
    while csize >= 8 and csize % 2 == 0:
        csize //= 2
    return csize


class Encoder(tnn.Module):
    """DCGAN encoder (reference models/ganomaly.py:24-76; same signature and child names)."""

    def __init__(self, isize, nz, nc, ndf, ngpu, n_extra_layers=0, add_final_conv=True):
        super(Encoder, self).__init__()
        self.ngpu = ngpu
        assert isize % 16 == 0, "isize has to be a multiple of 16"

        main = hnn.Sequential()
        main.add_module('initial-conv-{0}-{1}'.format(nc, ndf), hnn.Conv2d(nc, ndf, 4, 2, 1, bias=False))
        main.add_module('initial-relu-{0}'.format(ndf), hnn.LeakyReLU(0.2, inplace=True))
        csize, cndf = isize // 2, ndf

        for t in range(n_extra_layers):
            main.add_module('extra-layers-{0}-{1}-conv'.format(t, cndf), hnn.Conv2d(cndf, cndf, 3, 1, 1, bias=False))
            main.add_module('extra-layers-{0}-{1}-batchnorm'.format(t, cndf), hnn.BatchNorm2d(cndf))
            main.add_module('extra-layers-{0}-{1}-relu'.format(t, cndf), hnn.LeakyReLU(0.2, inplace=True))

        floor = _pyramid_floor(isize)
        while csize > floor:
            in_feat, out_feat = cndf, cndf * 2
            main.add_module('pyramid-{0}-{1}-conv'.format(in_feat, out_feat),
                            hnn.Conv2d(in_feat, out_feat, 4, 2, 1, bias=False))
            main.add_module('pyramid-{0}-batchnorm'.format(out_feat), hnn.BatchNorm2d(out_feat))
            main.add_module('pyramid-{0}-relu'.format(out_feat), hnn.LeakyReLU(0.2, inplace=True))
            cndf = cndf * 2
            csize = csize // 2

        if add_final_conv:
            main.add_module('final-{0}-{1}-conv'.format(cndf, 1), hnn.Conv2d(cndf, nz, csize, 1, 0, bias=False))

        self.main = main

    def forward(self, input):
        return _io(self.main, input)


class Decoder(tnn.Module):
    """DCGAN decoder (reference models/ganomaly.py:79-133)."""

    def __init__(self, isize, nz, nc, ngf, ngpu, n_extra_layers=0):
        super(Decoder, self).__init__()
        self.ngpu = ngpu
        assert isize % 16 == 0, "isize has to be a multiple of 16"

        floor = _pyramid_floor(isize)
        cngf, tisize = ngf // 2, floor
        while tisize != isize:
            cngf = cngf * 2
            tisize = tisize * 2

        main = hnn.Sequential()
        main.add_module('initial-{0}-{1}-convt'.format(nz, cngf), hnn.ConvTranspose2d(nz, cngf, floor, 1, 0, bias=False))
        main.add_module('initial-{0}-batchnorm'.format(cngf), hnn.BatchNorm2d(cngf))
        main.add_module('initial-{0}-relu'.format(cngf), hnn.ReLU(True))

        csize = floor
        while csize < isize // 2:
            main.add_module('pyramid-{0}-{1}-convt'.format(cngf, cngf // 2),
                            hnn.ConvTranspose2d(cngf, cngf // 2, 4, 2, 1, bias=False))
            main.add_module('pyramid-{0}-batchnorm'.format(cngf // 2), hnn.BatchNorm2d(cngf // 2))
            main.add_module('pyramid-{0}-relu'.format(cngf // 2), hnn.ReLU(True))
            cngf = cngf // 2
            csize = csize * 2

        for t in range(n_extra_layers):
            main.add_module('extra-layers-{0}-{1}-conv'.format(t, cngf), hnn.Conv2d(cngf, cngf, 3, 1, 1, bias=False))
            main.add_module('extra-layers-{0}-{1}-batchnorm'.format(t, cngf), hnn.BatchNorm2d(cngf))
            main.add_module('extra-layers-{0}-{1}-relu'.format(t, cngf), hnn.ReLU(True))

        main.add_module('final-{0}-{1}-convt'.format(cngf, nc), hnn.ConvTranspose2d(cngf, nc, 4, 2, 1, bias=False))
        main.add_module('final-{0}-tanh'.format(nc), hnn.Tanh())
        self.main = main

    def forward(self, input):
        return _io(self.main, input)


def _io(fn, x):
    """Run `fn` on a ClTensor; accept / return plain (N,C,H,W) float tensors at the user boundary."""
    if isinstance(x, ClTensor):
        return fn(x)
    y = fn(F.to_cl(x))
    return y.to_torch()


class NetD(tnn.Module):
    """Discriminator (reference models/ganomaly.py:137-157): Encoder(nz=1) split into features + classifier."""

    def __init__(self, opt):
        super(NetD, self).__init__()
        model = Encoder(opt.isize, 1, opt.nc, opt.ngf, opt.ngpu, opt.extralayers)
        layers = list(model.main.children())

        self.features = hnn.Sequential(*layers[:-1])
        self.classifier = hnn.Sequential(layers[-1])
        self.classifier.add_module('Sigmoid', hnn.Sigmoid())

    def forward(self, x):
        plain = not isinstance(x, ClTensor)
        xc = F.to_cl(x) if plain else x
        features = self.features(xc)
        classifier = self.classifier(features)          # (N,1,1,1) block == view(-1,1).squeeze(1)
        if plain:
            return classifier.to_torch().view(-1, 1).squeeze(1), features.to_torch()
        return classifier, features


class NetG(tnn.Module):
    """Generator (reference models/ganomaly.py:160-175): encoder -> decoder -> encoder."""

    def __init__(self, opt):
        super(NetG, self).__init__()
        self.encoder1 = Encoder(opt.isize, opt.nz, opt.nc, opt.ngf, opt.ngpu, opt.extralayers)
        self.decoder = Decoder(opt.isize, opt.nz, opt.nc, opt.ngf, opt.ngpu, opt.extralayers)
        self.encoder2 = Encoder(opt.isize, opt.nz, opt.nc, opt.ngf, opt.ngpu, opt.extralayers)

    def forward(self, x):
        plain = not isinstance(x, ClTensor)
        xc = F.to_cl(x) if plain else x
        latent_i = self.encoder1(xc)
        gen_imag = self.decoder(latent_i)
        # gen_imag has THREE consumers in the training step (encoder2 here, the L1 term and netD in Ganomaly): three handles whose
        # gradients are summed by one launch in float32 (functional.fanout) instead of autograd's two bf16 add kernels
        gen_enc, gen_imag, self._gen_for_d = F.fanout(gen_imag, 3)
        latent_o = self.encoder2(gen_enc)
        if plain:
            return gen_imag.to_torch(), latent_i.to_torch(), latent_o.to_torch()
        return gen_imag, latent_i, latent_o


# Upstream GANomaly option defaults.  They are NOT citeable in the reference tree (models/ganomaly.py reads an
# options object that lib/args.py never defines), so they are declared here as this build's defaults.
GANOMALY_DEFAULTS = dict(nz=100, ngf=64, nc=3, ngpu=1, extralayers=0, w_adv=1.0, w_con=50.0, w_enc=1.0,
                         lr=2e-4, beta1=0.5)


def make_opt(args=None, **over):
    d = dict(GANOMALY_DEFAULTS)
    if args is not None:
        d.update(isize=args.isize, nc=getattr(args, "ich", 3), batchsize=args.batchsize)
    d.update(over)
    return types.SimpleNamespace(**d)


def fold_frames(x):
    """(B,C,T,H,W) clip block -> B*T frames (B*T,C,H,W): a free view in channels-last."""
    t = x.t
    n, d, h, w, cp = t.shape
    return ClTensor(t.reshape(n * d, 1, h, w, cp), x.C, 2)


class Ganomaly(GANBaseModel):
    """Training step of reference models/ganomaly.py:459-519 (forward_g, forward_d, backward_g, backward_d,
    optimize_params, reinit_d) with the 4-tuple clip contract of lib/train_gan.py:69."""

    @property
    def name(self):
        return 'Ganomaly'

    def __init__(self, args, dataloader, opt=None):
        super(Ganomaly, self).__init__(args, dataloader)
        self.opt = opt if opt is not None else make_opt(args)
        self.netg = NetG(self.opt).to(self.device)
        self.netd = NetD(self.opt).to(self.device)
        self.netg.apply(weights_init_dcgan)
        self.netd.apply(weights_init_dcgan)
        vdist.broadcast_module(self.netg)
        vdist.broadcast_module(self.netd)

        self.l_adv = F.l2_loss
        self.l_con = F.l1_loss
        self.l_enc = F.l2_loss
        self.l_bce = F.bce_loss
        self.real_label, self.fake_label = 1.0, 0.0

        self.netg.train()
        self.netd.train()
        self.optimizer_d = hoptim.Adam(self.netd.parameters(), lr=self.opt.lr, betas=(self.opt.beta1, 0.999))
        self.optimizer_g = hoptim.Adam(self.netg.parameters(), lr=self.opt.lr, betas=(self.opt.beta1, 0.999))
        self.reducer_g = vdist.GradReducer.for_optimizer(self.optimizer_g)
        self.reducer_d = vdist.GradReducer.for_optimizer(self.optimizer_d)
        if self.load_pretrained():           # --resume (reference: right after weights_init)
            vdist.broadcast_module(self.netg)
            vdist.broadcast_module(self.netd)

    # ---- reference-named phases ------------------------------------------------------------------------------
    def set_input(self, data):
        super(Ganomaly, self).set_input(data)
        self.x = fold_frames(F.to_cl(self.input))   # frames, channels-last, compute dtype

    def forward_g(self):
        self.fake, self.latent_i, self.latent_o = self.netg(self.x)
        self.fake_d = self.netg._gen_for_d          # netD's handle of the generated frames (see NetG.forward)

    def forward_d(self):
        """netd(input) and netd(fake), ONCE each.  The reference evaluates both a second time inside backward_g
        (models/ganomaly.py:485) with unchanged weights and inputs: identical values.  Here the first results serve both
        backward passes — netd(fake) is kept attached to netG's graph and back-propagated twice (into netG with netD frozen,
        then into netD) — and the only other effect of the repeated forwards, a second BatchNorm running-statistics update
        with the same batch statistics, is applied directly (backward_g), in the reference's order."""
        bns = [m for m in self.netd.modules() if isinstance(m, hnn._BNS)]
        for m in bns:
            m._keep_batch_stats = True
        self.pred_real, self.feat_real = self.netd(self.x)
        stats_real = [(m, m._batch_stats) for m in bns if m.training]
        with F.collect_pools() as self._dfake_pools:
            self.pred_fake, self.feat_fake = self.netd(self.fake_d)
        stats_fake = [(m, m._batch_stats) for m in bns if m.training]
        for m in bns:
            m._keep_batch_stats = False
        self._d_repeat_stats = stats_real + stats_fake

    def backward_g(self, join=True):
        # The reference lets this backward also deposit gradients into netD's parameters and then discards them
        # (optimizer_d.zero_grad() at :515 runs before they are ever used).  Freezing netD here skips exactly that
        # discarded filter-gradient work; everything that survives the step is unchanged.
        netd_params = list(self.netd.parameters())
        for p in netd_params:
            p._vfd_frozen = True
        self.reducer_d.enabled = False
        try:
            for m, st in self._d_repeat_stats:      # the BatchNorm side effect of the reference's repeated netd(input), netd(fake)
                m._batch_stats = st
                m.repeat_running_update()
            self.err_g_adv = self.l_adv(self.feat_real.detach(), self.feat_fake)
            self.err_g_con = self.l_con(self.fake, self.x)
            self.err_g_enc = self.l_enc(self.latent_o, self.latent_i)
            self.err_g = F.weighted_sum((self.err_g_adv, self.opt.w_adv), (self.err_g_con, self.opt.w_con),
                                        (self.err_g_enc, self.opt.w_enc))
            # netd(fake)'s graph is walked again by backward_d: keep it
            torch.autograd.backward(self.err_g, inputs=[p for p in self.netg.parameters() if p.requires_grad], retain_graph=True)
        finally:
            for p in netd_params:
                p._vfd_frozen = False
            self.reducer_d.enabled = True
        if join:
            self.reducer_g.finish()

    def backward_d(self, join=True):
        self.err_d_real = self.l_bce(self.pred_real, self.real_label)
        self.err_d_fake = self.l_bce(self.pred_fake, self.fake_label)
        self.err_d = F.weighted_sum((self.err_d_real, 0.5), (self.err_d_fake, 0.5))
        for pool in self._dfake_pools:          # BatchNorm / bias sum buffers of netd(fake): second walk of that graph
            F.zero_(pool)
        skip = self.fake_d.t.data_ptr()         # ... which stops at netD's first layer (the reference detaches fake here)
        F._SKIP_INPUT_GRAD.add(skip)
        try:
            torch.autograd.backward(self.err_d, inputs=[p for p in self.netd.parameters() if p.requires_grad])
        finally:
            F._SKIP_INPUT_GRAD.discard(skip)
        if join:
            self.reducer_d.finish()

    def test(self):
        """Evaluation sweep of reference models/ganomaly.py:332-406 over ``self.dataloader['test']``: per frame the anomaly
        score mean((latent_i - latent_o)^2) over the nz latent channels, min-max scaled over the whole test set, ROC AUC
        against the labels (every frame of a clip carries the clip's label ``lb``).  As in the reference the nets are NOT
        switched to eval mode (there is no ``.eval()`` in that file): under torch.no_grad() BatchNorm normalises with each
        test batch's statistics and keeps updating the running ones.  Returns the reference's performance dict."""
        import time
        from collections import OrderedDict
        from ..lib.evaluate import evaluate
        scores, labels, times = [], [], []
        with torch.no_grad():
            for data in self.dataloader['test']:
                t0 = time.time()
                input, real, gt, lb = (d.to(self.device, non_blocking=True) for d in data)
                x = fold_frames(F.to_cl(input))
                self.fake, latent_i, latent_o = self.netg(x)
                li, lo = latent_i.to_torch().float(), latent_o.to_torch().float()
                err = torch.mean(torch.pow(li - lo, 2), dim=1)
                scores.append(err.reshape(err.size(0)))
                labels.append(lb.reshape(-1).repeat_interleave(err.size(0) // lb.numel()))
                times.append(time.time() - t0)
            an = torch.cat(scores)
            self.gt_labels = torch.cat(labels).long()
            self.an_scores = (an - torch.min(an)) / (torch.max(an) - torch.min(an))
        auc = evaluate(self.gt_labels.cpu().numpy(), self.an_scores.cpu().numpy(), metric='roc')
        import numpy as np
        return OrderedDict([('Avg Run Time (ms/batch)', float(np.mean(np.array(times)[:100]) * 1000)), ('AUC', auc)])

    def reinit_d(self):
        """Reference :496-500: re-apply the initialiser to netD.  Replicas must stay identical, so rank 0's new
        weights are broadcast."""
        self.netd.apply(weights_init_dcgan)
        vdist.broadcast_module(self.netd)
        F.invalidate_weight_cache()
        # A hipGraph replay runs no Python: the packed (K-major, compute-dtype) copies the captured kernels read are refreshed
        # only by the captured repack launch after each Adam step.  Re-pack netD's copies NOW, in place, or the next replay's
        # netD forward / backward would read the pre-reinit filters next to the re-initialised BatchNorm weights.
        self.optimizer_d._epoch[0] += 1
        F.repack_owned(self.optimizer_d._epoch)
        if self.rank == 0:
            print('   Reloading net d')

    def d_collapsed(self):
        """Reference :519 `err_d.item() < 1e-5` — the step's one host sync.  Under data parallelism the decision is
        taken on the rank-averaged loss so that every replica takes the same branch."""
        e = self.err_d.detach().clone()
        if self.world_size > 1:
            torch.distributed.all_reduce(e)
            e = e / self.world_size
        return e.item() < 1e-5

    def step_program(self):
        """The step for hipGraph capture under data parallelism (vfd_gan_amd.graph.GraphedStep): collective-free graphs
        with the gradient reductions issued between them.  optimizer_g.step() is moved behind backward_d — backward_d
        reads netD's weights and the activations forward_d saved BEFORE the generator update in the reference order too
        (models/ganomaly.py:502-519), so the result is identical — which lets netG's all-reduce run beside netD's
        backward pass, and netD's beside netG's Adam update."""
        def a():
            self.forward_g()
            self.forward_d()
            self.optimizer_g.zero_grad()
            self.backward_g(join=False)

        def b():
            self.optimizer_d.zero_grad()
            self.backward_d(join=False)

        def c():
            self.optimizer_g.step()

        def d():
            self.optimizer_d.step()
            self._publish_errors()
        return [("graph", a), ("reduce", self.reducer_g), ("graph", b), ("reduce", self.reducer_d),
                ("join", self.reducer_g), ("graph", c), ("join", self.reducer_d), ("graph", d)]

    def _publish_errors(self):
        self.errors_dict.update({'g/err_g/train': self.err_g, 'g/err_g_adv/train': self.err_g_adv,
                                 'g/err_g_con/train': self.err_g_con, 'g/err_g_enc/train': self.err_g_enc,
                                 'd/err_d/train': self.err_d, 'd/err_d_real/train': self.err_d_real,
                                 'd/err_d_fake/train': self.err_d_fake})

    def optimize_params(self, check_collapse=True):
        self.forward_g()
        self.forward_d()

        self.optimizer_g.zero_grad()
        self.backward_g()
        self.optimizer_g.step()

        self.optimizer_d.zero_grad()
        self.backward_d()
        self.optimizer_d.step()

        self._publish_errors()
        if check_collapse and self.d_collapsed():   # reference :519 (one host sync per step, as there)
            self.reinit_d()
