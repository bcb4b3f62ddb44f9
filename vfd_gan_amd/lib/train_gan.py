"""Trainer base with the attributes and loop of the reference's lib/train_gan.py:17-85.

Same public surface (``GANBaseModel(args, dataloader)``, ``.train()``, ``.save_weights(name_head)``, the
``*_dict`` summaries, ``save_root_dir`` / ``weight_dir`` naming, ``args.txt``); subclasses supply
``optimize_params()`` and optionally ``test()``.  TensorBoard (absent in this image) is replaced by a JSON-lines
scalar log; the periodic full test sweep is an eval feature outside the hot path and runs
only when a subclass defines ``test``.
"""
import json
import os
from collections import OrderedDict
from datetime import datetime

import torch

from .. import dist as vdist


class ScalarLog:
    """Minimal stand-in for torch.utils.tensorboard.SummaryWriter (scalars only, JSON lines)."""

    def __init__(self, log_dir):
        self.path = os.path.join(log_dir, "scalars.jsonl")

    def add_scalars(self, main_tag, tag_scalar_dict, global_step=None):
        with open(self.path, "a") as f:
            f.write(json.dumps({"step": global_step, "tag": main_tag, **{k: float(v) for k, v in tag_scalar_dict.items()}}) + "\n")

    def add_scalar(self, tag, value, global_step=None):
        self.add_scalars(tag, {"value": value}, global_step)


class GANBaseModel():
    def __init__(self, args, dataloader):
        self.args = args
        self.dataloader = dataloader
        self.device = torch.device("cuda", torch.cuda.current_device()) if torch.cuda.is_available() else None
        if self.device is None:
            raise RuntimeError("vfd_gan_amd trainers need a HIP device (no CPU fallback on the product path)")

        self.global_step = 0
        self.epoch = 0
        self.start_epoch = 0
        self.best_roc = 0
        self.best_pr = 0
        self.color_video_dict = OrderedDict()
        self.gray_video_dict = OrderedDict()
        self.errors_dict = {}
        self.hist_dict = OrderedDict()
        self.score_dict = OrderedDict()

        self.rank, self.world_size = vdist.rank(), vdist.world_size()
        current_time = datetime.now().strftime("%b%d_%H-%M-%S")
        comment = "b{}xd{}xwh{}_lr-{}_w-a{}c{}".format(args.batchsize, args.nfr, args.isize,
                                                       args.lr, args.w_adv, args.w_con)
        self.save_root_dir = os.path.join(args.result_root, args.model, comment, current_time)
        self.weight_dir = os.path.join(self.save_root_dir, 'weights')
        logdir = os.path.join(self.save_root_dir, "runs")
        self.writer = None
        if self.rank == 0:
            os.makedirs(self.weight_dir, exist_ok=True)
            os.makedirs(logdir, exist_ok=True)
            self.writer = ScalarLog(logdir)
            with open(self.save_root_dir + "/args.txt", mode="w") as f:
                json.dump(args.__dict__, f, indent=4)
            print("\n SAVE PATH == {} \n".format(self.save_root_dir))

    def save_weights(self, name_head):
        """Reference lib/train_gan.py:52-57: ``{name}_ep%04d_netG.pth`` / ``_netD.pth`` holding {'epoch','state_dict'}
        (float32, torch's (Cout,Cin,k..) layout: loads into the reference's plain torch.nn classes with strict=True).
        Build extension next to them: ``{name}_ep%04d_optim.pth`` = both Adam states, global_step, epoch — the
        reference saves no optimiser state, so its own resume restarts Adam from zero moments; with this file present
        a resumed run continues bit-identically."""
        if self.rank != 0:
            return
        head = '%s/%s_ep%04d' % (self.weight_dir, name_head, self.epoch)
        torch.save({'epoch': self.epoch + 1, 'state_dict': self.netg.state_dict()}, head + '_netG.pth')
        torch.save({'epoch': self.epoch + 1, 'state_dict': self.netd.state_dict()}, head + '_netD.pth')
        og, od = self._optimizers()
        from .. import functional as F
        torch.save({'epoch': self.epoch + 1, 'global_step': self.global_step, 'g': og.state_dict(), 'd': od.state_dict(),
                    'dropout': F.dropout_state(), 'device_rng': torch.cuda.get_rng_state(self.device)}, head + '_optim.pth')
        return head + '_netG.pth'

    def _optimizers(self):
        return (getattr(self, "optimizer_g", None) or self.g_opt), (getattr(self, "optimizer_d", None) or self.d_opt)

    def load_pretrained(self):
        """``--resume <..._netG.pth>`` (reference models/mygannet.py:245-256, models/anogan.py same block): load netG from
        the given file and netD from the sibling file, both in the reference's {'epoch','state_dict'} format, keys with or
        without DataParallel's ``module.`` prefix (lib/utils.py:15-22).  Called by the model constructors AFTER the
        optimisers exist: ``load_state_dict`` copies in place, so parameters stay views into the Adam arenas.

        The reference derives the netD path as ``resume.rsplit('_', 1)[0] + 'netD.pth'`` — without the underscore its
        own save_weights writes — so it cannot find the files it saved; that literal name is tried first (a user may
        have renamed files to satisfy it), then the ``_netD.pth`` that save_weights produces.  ganomaly's reference
        takes a DIRECTORY holding netG.pth / netD.pth (models/ganomaly.py:430-434): accepted too."""
        from .utils import fix_model_state_dict
        from .. import functional as F
        resume = getattr(self.args, "resume", "") or ""
        if resume == "":
            return False
        if os.path.isdir(resume):
            g_path, d_cands = os.path.join(resume, "netG.pth"), [os.path.join(resume, "netD.pth")]
        else:
            g_path = resume
            stem = resume.rsplit("_", 1)[0]
            d_cands = [stem + "netD.pth", stem + "_netD.pth"]
        d_path = next((c for c in d_cands if os.path.exists(c)), None)
        if not os.path.exists(g_path) or d_path is None:
            raise IOError("Model weights not found: %s / %s" % (g_path, " | ".join(d_cands)))
        if self.rank == 0:
            print("\n Loading pretrained network weight = {}".format(resume))
        g_ck = torch.load(g_path, map_location=self.device, weights_only=True)
        d_ck = torch.load(d_path, map_location=self.device, weights_only=True)
        self.netg.load_state_dict(fix_model_state_dict(g_ck['state_dict']))
        self.netd.load_state_dict(fix_model_state_dict(d_ck['state_dict']))
        F.invalidate_weight_cache()          # packed filter copies are keyed on the parameter version
        self.start_epoch = int(g_ck.get('epoch', 0))
        o_path = g_path[:-len("_netG.pth")] + "_optim.pth" if g_path.endswith("_netG.pth") else None
        if o_path and os.path.exists(o_path):
            o_ck = torch.load(o_path, map_location=self.device, weights_only=True)
            og, od = self._optimizers()
            og.load_state_dict(o_ck['g'])
            od.load_state_dict(o_ck['d'])
            self.global_step = int(o_ck.get('global_step', 0))
            if 'dropout' in o_ck:
                F.set_dropout_state(o_ck['dropout'], self.device)
            if 'device_rng' in o_ck and self.world_size == 1:
                # (the file holds RANK 0's generator state: under data parallelism every rank keeps the rank-offset seed its
                # constructor set, so that replicas go on drawing DIFFERENT noise — SURVEY.md 8e; the bit-identical continuation
                # of tests/test_trainer_resume.py is a one-rank property)
                torch.cuda.set_rng_state(o_ck['device_rng'].cpu(), self.device)
        if self.rank == 0:
            print("\n Done.\n")
        return True

    def set_input(self, data):
        self.input, self.real, self.gt, self.lb = (d.to(self.device, non_blocking=True) for d in data)

    def train(self):
        if self.rank == 0:
            print(" >> Training model %s." % self.args.model)
        for self.epoch in range(getattr(self, "start_epoch", 0), self.args.ep):
            for i, data in enumerate(self.dataloader['train']):
                self.global_step += 1
                self.set_input(data)
                self.optimize_params()
                if self.global_step % self.args.freq == 0:
                    if hasattr(self, "test"):
                        self.test()
                    self.update_summary()
            # The reference checkpoints from test() when a score improves (models/mygannet.py:449-454, anogan :213-216)
            # and, for ganomaly, per epoch (models/ganomaly.py:323-327); the in-loop test sweep is not on the hot path
            # (SURVEY.md 8f N2), so the state of every finished epoch is kept under the name head "last".
            self.last_checkpoint = self.save_weights("last")
            if self.rank == 0:
                print("[TRAIN Epoch %d/%d] step %d %s" % (self.epoch + 1, self.args.ep, self.global_step,
                                                          {k: round(float(v), 5) for k, v in self.errors().items()}))
        if self.rank == 0:
            print(" >> Training model %s.[Done]" % self.args.model)

    def errors(self):
        """Loss scalars of the last step as Python floats (ONE device sync for all of them; the reference pays one
        ``.item()`` sync per scalar, models/mygannet.py:314-342)."""
        if not self.errors_dict:
            return {}
        keys = list(self.errors_dict.keys())
        vals = torch.stack([self.errors_dict[k].detach().float().reshape(()) if torch.is_tensor(self.errors_dict[k])
                            else torch.tensor(float(self.errors_dict[k]), device=self.device) for k in keys]).tolist()
        return dict(zip(keys, vals))

    def update_summary(self):
        if self.writer is None:
            return
        for t, e in self.errors().items():
            spk = t.rsplit('/', 1)
            self.writer.add_scalars(spk[0], {spk[1]: e}, self.global_step)
        for t, s in self.score_dict.items():
            self.writer.add_scalar(t, s, self.global_step)
