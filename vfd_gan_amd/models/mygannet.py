"""MyGAN ((2+1)D U-Net generator + spatial / temporal dual discriminator) on HIP kernels, behind the surface of the
reference's models/mygannet.py (NetgConv :13-28, NetG :31-101, NetdConv :104-116, SDisc :119-162, TDisc :164-196,
NetD :200-213, MyGAN :216-367).

Generalisation: SDisc.linear / TDisc.linear in-features follow (nfr, isize) instead of the reference's hard-wired
nfr=16, isize=128 (:134, :176); at those defaults they are 4096 and 256 as in the reference.
Optical flow (lib/utils.py:94-129, CPU Farneback through cv2) is outside the hot path: the temporal discriminator's
input streams are inputs of the step (`self.gt_flow`, `self.pre_flow`; synthetic stand-ins by default).
"""
import types

import torch
import torch.nn as tnn

from .. import _lib
from .. import dist as vdist
from .. import functional as F
from .. import nn as hnn
from .. import optim as hoptim
from ..functional import ClTensor
from ..lib.data import synthetic_flow
from ..lib.train_gan import GANBaseModel
from ..lib.utils import weights_init
from .spatiotempconv import SpatioTemporalConv


def _conv_bn_act(block, x, slope, pool=None, keep_full=False):
    """SpatioTemporalConv -> BatchNorm3d -> LeakyReLU(slope) with the BatchNorm statistics taken from the temporal
    conv's epilogue (bf16) and normalise+activate in one pass.  `pool`: the AvgPool3d module the caller applies to the result
    (the discriminators: to nothing else; NetG's encoder, keep_full: the full-resolution result is a skip connection too and
    (pooled, full) is returned): absorbed into the BatchNorm pass where it can be (functional._BnActPool)."""
    if block.bn.training and hnn.use_epilogue_stats(x):
        k = F.stats_buffer_numel(block.bn.num_features)
        buf = F.zeros(4 * k, torch.float32, x.t.device)      # forward statistics (float64) | backward sums | bias sums: one fill
        fstats, bsums, brep = buf[:2 * k].view(torch.float64), buf[2 * k:3 * k], buf[3 * k:]
        tbias = block.conv.temporal_conv.bias
        tok = {"taken": False, "rep": brep} if tbias is not None else None
        x = block.conv(x, stats=fstats, bias_token=tok)
        if pool is not None and not hnn._NO_HANDOVER and block.bn.momentum is not None:
            pks = F.pool_fusable(pool.kernel_size, pool.stride, pool.padding, tuple(x.t.shape[1:4]))
            if pks is not None:
                return block.bn.forward_pooled(x, _lib.ACT_LRELU, slope, fstats, bsums, tbias, tok, pks, keep_full)
        x = block.bn(x, act=_lib.ACT_LRELU, slope=slope, sums=fstats, bwd_sums=bsums, conv_bias=tbias, bias_token=tok)
    else:
        x = block.bn(block.conv(x), act=_lib.ACT_LRELU, slope=slope)
    if pool is None:
        return x
    return (pool(x), x) if keep_full else pool(x)


class NetgConv(tnn.Module):
    def __init__(self, in_fi, out_fi, kernel_size=3):
        super(NetgConv, self).__init__()
        padding = kernel_size // 2
        self.conv = SpatioTemporalConv(in_fi, out_fi, kernel_size, padding=padding)
        self.bn = hnn.BatchNorm3d(out_fi)
        self.lrelu = hnn.LeakyReLU(0.2, inplace=True)

    def forward(self, x, pool=None):
        """pool: the encoder's AvgPool3d -> returns (pooled, full-resolution) from one BatchNorm pass."""
        return _conv_bn_act(self, x, self.lrelu.negative_slope, pool, keep_full=pool is not None)


class NetG(tnn.Module):
    def __init__(self, nc=3, ngf=32):
        super(NetG, self).__init__()
        self.dconv1 = NetgConv(nc, ngf)
        self.dconv2 = NetgConv(ngf, ngf * 2)
        self.dconv3 = NetgConv(ngf * 2, ngf * 4)
        self.dconv4 = NetgConv(ngf * 4, ngf * 8)
        self.dconv5 = NetgConv(ngf * 8, ngf * 16)

        self.avgpool = hnn.AvgPool3d(2)

        self.uconv5 = NetgConv(ngf * 16, ngf * 8)
        self.uconv4 = NetgConv(ngf * 8 + ngf * 8, ngf * 8)
        self.uconv3 = NetgConv(ngf * 8 + ngf * 4, ngf * 4)
        self.uconv2 = NetgConv(ngf * 4 + ngf * 2, ngf * 2)
        self.uconv1 = NetgConv(ngf * 2 + ngf, ngf)

        self.dropout = hnn.Dropout(p=0.25)
        self.upsamp = hnn.Upsample(scale_factor=2, mode='trilinear', align_corners=True)

        self.conv_last = hnn.Conv3d(ngf, 1, 3, stride=1, padding=1, bias=False)
        self.sigmoid = hnn.Sigmoid()

    def forward(self, x):
        plain = not isinstance(x, ClTensor)
        if plain:
            x = F.to_cl(x)
        # each encoder level feeds its pooled form to the next level and its full-resolution form to the decoder (skip):
        # both come out of the level's BatchNorm pass, and that pass's backward sums the two gradients that come back
        p1, dconv1 = self.dconv1(x, pool=self.avgpool)
        p2, dconv2 = self.dconv2(p1, pool=self.avgpool)
        p3, dconv3 = self.dconv3(p2, pool=self.avgpool)
        p4, dconv4 = self.dconv4(p3, pool=self.avgpool)
        latent_i = self.dconv5(p4)

        # decoder joints: cat([Upsample(x), skip]) written in one pass (F.upsample_cat; self.upsamp's configuration is the
        # one that pass implements: scale 2, trilinear, align_corners=True, checked by hnn.Upsample at construction use)
        self.upsamp.check()
        x = self.dropout(self.uconv5(latent_i))
        x = self.dropout(self.uconv4(F.upsample_cat(x, dconv4)))
        x = self.dropout(self.uconv3(F.upsample_cat(x, dconv3)))
        x = self.dropout(self.uconv2(F.upsample_cat(x, dconv2)))
        x = self.uconv1(F.upsample_cat(x, dconv1))
        predict = self.conv_last(x, act=_lib.ACT_SIGMOID)      # conv_last + sigmoid in one kernel
        return predict.to_torch() if plain else predict


class NetdConv(tnn.Module):
    def __init__(self, in_fi, out_fi, kernel_size=None, padding=None):
        super(NetdConv, self).__init__()
        self.conv = SpatioTemporalConv(in_fi, out_fi, kernel_size, padding=padding)
        self.bn = hnn.BatchNorm3d(out_fi)
        self.lrelu = hnn.LeakyReLU()

    def forward(self, x, pool=None):
        return _conv_bn_act(self, x, self.lrelu.negative_slope, pool)


class SDisc(tnn.Module):
    def __init__(self, nc, nfr, ndf=32, kernel=None, padding=None, isize=128):
        super(SDisc, self).__init__()
        netdconv = lambda in_fi, out_fi: NetdConv(in_fi, out_fi, kernel_size=kernel, padding=padding)  # noqa: E731
        self.dconv1 = netdconv(nc, ndf)
        self.dconv2 = netdconv(ndf, ndf * 2)
        self.dconv3 = netdconv(ndf * 2, ndf * 4)
        self.dconv4 = netdconv(ndf * 4, ndf * 8)
        self.dconv5 = netdconv(ndf * 8, ndf * 16)
        self.dconv6 = netdconv(ndf * 16, ndf * 32)

        self.avgpool = hnn.AvgPool3d((1, 2, 2))
        self.gpool = hnn.AvgPool3d((nfr, 1, 1), stride=1)
        self.linear = hnn.Linear(ndf * 32 * (isize // 64) * (isize // 64), 1)
        self.sigmoid = hnn.Sigmoid()

    def forward(self, x):
        plain = not isinstance(x, ClTensor)
        if plain:
            x = F.to_cl(x)
        for conv in (self.dconv1, self.dconv2, self.dconv3, self.dconv4, self.dconv5, self.dconv6):
            x = conv(x, pool=self.avgpool)      # conv -> BatchNorm -> LeakyReLU -> AvgPool, the pool inside the BatchNorm pass
        features = x
        x = self.gpool(features)
        classifier = self.linear(x, act=_lib.ACT_SIGMOID)     # Linear over the flattened block + Sigmoid
        if plain:
            return classifier.to_torch().squeeze(1), features.to_torch()
        return classifier, features


class TDisc(tnn.Module):
    def __init__(self, nc, isize, ndf=32, kernel=None, padding=None, nfr=16):
        super(TDisc, self).__init__()
        netdconv = lambda in_fi, out_fi: NetdConv(in_fi, out_fi, kernel_size=kernel, padding=padding)  # noqa: E731
        self.dconv1 = netdconv(nc, ndf)
        self.dconv2 = netdconv(ndf, ndf * 2)
        self.dconv3 = netdconv(ndf * 2, ndf * 4)

        self.avgpool = hnn.AvgPool3d((2, 1, 1))
        self.gpool = hnn.AvgPool3d((1, isize, isize), stride=1)
        self.linear = hnn.Linear(ndf * 4 * (nfr // 8), 1)
        self.sigmoid = hnn.Sigmoid()

    def forward(self, x):
        plain = not isinstance(x, ClTensor)
        if plain:
            x = F.to_cl(x)
        for conv in (self.dconv1, self.dconv2, self.dconv3):
            x = conv(x, pool=self.avgpool)
        features = x
        x = self.gpool(features)
        classifier = self.linear(x, act=_lib.ACT_SIGMOID)
        if plain:
            return classifier.to_torch().squeeze(1), features.to_torch()
        return classifier, features


class NetD(tnn.Module):
    def __init__(self, args):
        super(NetD, self).__init__()
        self.spatdisc = SDisc(3, args.nfr, kernel=(1, 3, 3), padding=(0, 1, 1), isize=args.isize)
        self.tempdisc = TDisc(3, args.isize, kernel=(3, 1, 1), padding=(1, 0, 0), nfr=args.nfr)

    def forward(self, x, y):
        s_cls, s_feat = self.spatdisc(x)
        t_cls, t_feat = self.tempdisc(y)
        return s_cls, s_feat, t_cls, t_feat


class MyGAN(GANBaseModel):
    def __init__(self, args, dataloader):
        super(MyGAN, self).__init__(args, dataloader)
        if getattr(args, "ae", False):
            # reference :224-227 means this (`netg = AutoEncoder()`), but then CALLS the instance without an input where it
            # should hand it over (:233 / :239), so --ae cannot run there; here the (2+1)D auto-encoder baseline is the generator
            from .mystcnn import AutoEncoder
            self.netg = AutoEncoder().to(self.device)
        else:
            self.netg = NetG(getattr(args, "ich", 3)).to(self.device)
        self.netd = NetD(args).to(self.device)
        self.netg.apply(weights_init)
        self.netd.apply(weights_init)
        vdist.broadcast_module(self.netg)
        vdist.broadcast_module(self.netd)

        self.real_label, self.gout_label = 1.0, 0.0
        self.l_adv = F.l2_loss
        self.l_con = F.weighted_bce          # pos_weight=2 always: the reference overwrites its lambda (:265-266)
        self.l_bce = F.bce_loss

        self.optimizer_d = hoptim.Adam(self.netd.parameters(), lr=self.args.lr, betas=(self.args.beta1, 0.999))
        self.optimizer_g = hoptim.Adam(self.netg.parameters(), lr=self.args.lr, betas=(self.args.beta1, 0.999))
        self.reducer_g = vdist.GradReducer.for_optimizer(self.optimizer_g)
        self.reducer_d = vdist.GradReducer.for_optimizer(self.optimizer_d)
        if self.load_pretrained():           # --resume (reference: right after weights_init)
            vdist.broadcast_module(self.netg)
            vdist.broadcast_module(self.netd)
        self.gt_flow = self.pre_flow = None

    def set_input(self, data, gt_flow=None, pre_flow=None):
        super(MyGAN, self).set_input(data)
        self.input_cl, self.gt_cl = F.to_cl(self.input), F.to_cl(self.gt)
        B = self.input.shape[0]
        if gt_flow is None:   # stand-ins for video_to_flow (lib/utils.py:94-129), SURVEY.md 8(d)
            gt_flow = synthetic_flow(B, self.args.nfr, self.args.isize, seed=4321 + self.global_step)
            pre_flow = synthetic_flow(B, self.args.nfr, self.args.isize, seed=8642 + self.global_step)
        self.gt_flow = F.to_cl(gt_flow.to(self.device, non_blocking=True))
        self.pre_flow = F.to_cl(pre_flow.to(self.device, non_blocking=True))

    def forward_g(self):
        self.predict = self.netg(self.input_cl)

    def forward_d(self):
        pre_3ch = F.gray2rgb(self.predict.detach())
        gt_3ch = F.gray2rgb(self.gt_cl)
        self.s_pred_real, self.s_feat_real, self.t_pred_real, self.t_feat_real = self.netd(gt_3ch, self.gt_flow)
        self.s_pred_fake, self.s_feat_fake, self.t_pred_fake, self.t_feat_fake = self.netd(pre_3ch, self.pre_flow)

    def backward_g(self, join=True):
        # Everything netD sees is detached from netG (reference :279-286), so the adversarial term has no gradient
        # path to netG; its gradients w.r.t. netD are discarded by optimizer_d.zero_grad() (:364).  It is therefore
        # evaluated for its VALUE only, and only the reconstruction term is back-propagated.
        err_g_adv_s = self.l_adv(self.s_feat_real.detach(), self.s_feat_fake.detach())
        err_g_adv_t = self.l_adv(self.t_feat_real.detach(), self.t_feat_fake.detach())
        err_g_adv = F.weighted_sum((err_g_adv_s, 1.0), (err_g_adv_t, 1.0))
        err_g_con = self.l_con(self.predict, self.gt_cl)
        err_g = F.weighted_sum((err_g_adv, self.args.w_adv), (err_g_con, self.args.w_con))
        err_g.backward()
        if join:
            self.reducer_g.finish()
        self.errors_dict.update({'g/err_g/train': err_g, 'g/err_g_adv/train': err_g_adv, 'g/err_g_adv_s/train': err_g_adv_s,
                                 'g/err_g_adv_t/train': err_g_adv_t, 'g/err_g_con/train': err_g_con})

    def backward_d(self, join=True):
        err_d_real_s = self.l_bce(self.s_pred_real, self.real_label)
        err_d_real_t = self.l_bce(self.t_pred_real, self.real_label)
        err_d_fake_s = self.l_bce(self.s_pred_fake, self.gout_label)
        err_d_fake_t = self.l_bce(self.t_pred_fake, self.gout_label)
        err_d_real = F.weighted_sum((err_d_real_s, 0.5), (err_d_real_t, 0.5))
        err_d_fake = F.weighted_sum((err_d_fake_s, 0.5), (err_d_fake_t, 0.5))
        err_d = F.weighted_sum((err_d_real, 0.5), (err_d_fake, 0.5))
        self.errors_dict.update({'d/err_d_real_s/train': err_d_real_s, 'd/err_d_real_t/train': err_d_real_t,
                                 'd/err_d_fake_s/train': err_d_fake_s, 'd/err_d_fake_t/train': err_d_fake_t,
                                 'd/err_d_real/train': err_d_real, 'd/err_d_fake/train': err_d_fake, 'd/err_d/train': err_d})
        err_d.backward()
        if join:
            self.reducer_d.finish()

    def reinit_d(self):
        """Reference :253-256.  The initialiser writes through `.data` (no version bump): the packed-filter cache is
        invalidated explicitly, and under data parallelism rank 0's new weights are broadcast so replicas stay equal."""
        self.netd.apply(weights_init)
        vdist.broadcast_module(self.netd)
        F.invalidate_weight_cache()
        # A hipGraph replay runs no Python: the packed (K-major, compute-dtype) copies the captured kernels read are refreshed
        # only by the captured repack launch after each Adam step.  Re-pack netD's copies NOW, in place, or the next replay's
        # netD forward / backward would read the pre-reinit filters next to the re-initialised BatchNorm weights.
        self.optimizer_d._epoch[0] += 1
        F.repack_owned(self.optimizer_d._epoch)
        if self.rank == 0:
            print('Reloading Net d')

    def step_program(self):
        """Graph capture under data parallelism (vfd_gan_amd.graph.GraphedStep): as Ganomaly.step_program — netG's Adam
        update is moved behind netD's backward pass (which reads nothing of netG: everything netD saw was detached,
        reference :279-286), so netG's all-reduce overlaps netD's backward and netD's overlaps netG's update."""
        def a():
            F.dropout_begin_step(self.device)
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
        return [("graph", a), ("reduce", self.reducer_g), ("graph", b), ("reduce", self.reducer_d),
                ("join", self.reducer_g), ("graph", c), ("join", self.reducer_d), ("graph", d)]

    def test(self, flows=None):
        """In-loop evaluation sweep, reference :369-475: the nets run under torch.no_grad() in whatever mode they are in — the
        reference never calls .eval() here, so BatchNorm keeps using (and updating) batch statistics and Dropout stays
        active; predict -> threshold -> 5x5 opening (on the device, lib/utils.morphology_proc), both discriminator passes,
        the 12 loss means, ROC / PR / F1 over all test pixels (lib/evaluate.py) and a checkpoint when ROC (else PR) improves.
        `flows(i, B)` -> (gt_flow, pre_flow) stands in for video_to_flow (CPU Farneback, SURVEY.md 8f N1: not built);
        default: the synthetic streams training uses."""
        import numpy as np
        from ..lib.evaluate import evaluate
        from ..lib.utils import morphology_proc, threshold
        keys = ("err_g_adv_s", "err_g_adv_t", "err_g_con", "err_d_real_s", "err_d_real_t", "err_d_fake_s", "err_d_fake_t")
        acc = {k: [] for k in keys}
        predicts, gts = [], []
        with torch.no_grad():
            for i, data in enumerate(self.dataloader['test']):
                input, real, gt, lb = (d.to(self.device, non_blocking=True) for d in data)
                F.dropout_begin_step(self.device)
                B = input.shape[0]
                predict_ = self.netg(F.to_cl(input))                     # ClTensor (N,1,T,H,W)
                p_t = predict_.to_torch()
                t_pre_ = threshold(p_t)
                m_pre_ = morphology_proc(t_pre_)
                gts.append(gt.permute(0, 2, 3, 4, 1))
                predicts.append(m_pre_.permute(0, 2, 3, 4, 1))
                if flows is not None:
                    gt_flow_, pre_flow_ = flows(i, B)
                else:
                    gt_flow_ = synthetic_flow(B, self.args.nfr, self.args.isize, seed=14321 + i)
                    pre_flow_ = synthetic_flow(B, self.args.nfr, self.args.isize, seed=18642 + i)
                gt_cl = F.to_cl(gt)
                gt_3ch_, pre_3ch_ = F.gray2rgb(gt_cl), F.gray2rgb(predict_)
                gf, pf = F.to_cl(gt_flow_.to(self.device)), F.to_cl(pre_flow_.to(self.device))
                s_pred_real_, s_feat_real_, t_pred_real_, t_feat_real_ = self.netd(gt_3ch_, gf)
                s_pred_fake_, s_feat_fake_, t_pred_fake_, t_feat_fake_ = self.netd(pre_3ch_, pf)
                acc["err_g_adv_s"].append(self.l_adv(s_feat_real_, s_feat_fake_))
                acc["err_g_adv_t"].append(self.l_adv(t_feat_real_, t_feat_fake_))
                acc["err_g_con"].append(self.l_con(predict_, gt_cl))
                acc["err_d_real_s"].append(self.l_bce(s_pred_real_, self.real_label))
                acc["err_d_real_t"].append(self.l_bce(t_pred_real_, self.real_label))
                acc["err_d_fake_s"].append(self.l_bce(s_pred_fake_, self.gout_label))
                acc["err_d_fake_t"].append(self.l_bce(t_pred_fake_, self.gout_label))
                self.color_video_dict.update({'test/input-real': torch.cat([input, real], dim=3)})
                self.gray_video_dict.update({'test/gt-pre-th-morph': torch.cat([gt, p_t, t_pre_, m_pre_], dim=3)})
                self.hist_dict.update({"test/inp": input, "test/gt": gt, "test/predict": p_t, "test/t_pre": t_pre_, "test/m_pre": m_pre_})
            # ONE device -> host transfer for the sweep's scalars (the reference pays an .item() sync per scalar and batch)
            e = {k: torch.stack([v.detach().float().reshape(()) for v in acc[k]]).cpu().numpy().astype(np.float64) for k in keys}
            gts_np = np.asarray(torch.stack(gts).cpu().numpy(), dtype=np.int32).flatten()
            pre_np = np.asarray(torch.stack(predicts).cpu().numpy()).flatten()
        err_g_adv = e["err_g_adv_s"] + e["err_g_adv_t"]
        err_g = e["err_g_adv_t"] * self.args.w_adv + e["err_g_con"] * self.args.w_con      # reference :416: temporal term only
        err_d_real = (e["err_d_real_s"] + e["err_d_real_t"]) * 0.5
        err_d_fake = (e["err_d_fake_s"] + e["err_d_fake_t"]) * 0.5
        err_d = (err_d_real + err_d_fake) * 0.5
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
        self.errors_dict.update({
            'd/err_d_real_s/test': float(np.mean(e["err_d_real_s"])), 'd/err_d_real_t/test': float(np.mean(e["err_d_real_t"])),
            'd/err_d_fake_s/test': float(np.mean(e["err_d_fake_s"])), 'd/err_d_fake_t/test': float(np.mean(e["err_d_fake_t"])),
            'd/err_d_real/test': float(np.mean(err_d_real)), 'd/err_d_fake/test': float(np.mean(err_d_fake)),
            'd/err_d/test': float(np.mean(err_d)),
            'g/err_g_adv_s/test': float(np.mean(e["err_g_adv_s"])), 'g/err_g_adv_t/test': float(np.mean(e["err_g_adv_t"])),
            'g/err_g_adv/test': float(np.mean(err_g_adv)), 'g/err_g_con/test': float(np.mean(e["err_g_con"])),
            'g/err_g/test': float(np.mean(err_g))})
        return {"roc": roc, "pr": pr, "f1": f1}

    def optimize_params(self):
        F.dropout_begin_step(self.device)
        self.netg.train()
        self.netd.train()

        self.forward_g()
        self.forward_d()

        self.optimizer_g.zero_grad()
        self.backward_g()
        self.optimizer_g.step()

        self.optimizer_d.zero_grad()
        self.backward_d()
        self.optimizer_d.step()


def make_args(nfr=16, isize=128, **kw):
    d = dict(nfr=nfr, isize=isize, ich=3, batchsize=2, lr=2e-5, beta1=0.5, w_adv=1, w_con=10, pos_weight=2, freq=10 ** 9,
             ep=1, model="mygan", result_root="./results", gpu=[0], ae=False)
    d.update(kw)
    return types.SimpleNamespace(**d)
