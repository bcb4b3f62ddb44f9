"""Supervised baseline trainer with the attributes and loop of the reference's lib/train_stcnn.py:17-197 (``VFD_STCNN``): one
net (``--model c2plus1d`` -> models.mystcnn.AutoEncoder, ``--model xception`` -> models.xception.Xception), BCELoss against the
tamper mask, Adam(lr, (beta1, 0.999)); every ``freq`` steps the test sweep (threshold -> 5x5 opening -> ROC / PR / F1) with a
checkpoint when a score improves.  SURVEY.md section 8f N4.  ``clstm`` (models/convlstm.py) is not built.

Differences from the reference, all forced by the environment or by data parallelism: TensorBoard is replaced by the JSON-lines
scalar log of lib/train_gan.py; one process per GPU with RCCL gradient reduction instead of DataParallel; the step's only host
sync is the loss scalar the summaries ask for."""
import json
import os
from collections import OrderedDict
from datetime import datetime

import numpy as np
import torch

from .. import dist as vdist
from .. import functional as F
from .. import optim as hoptim
from .train_gan import ScalarLog
from .utils import fix_model_state_dict, morphology_proc, threshold, weights_init


class VFD_STCNN():
    def __init__(self, args, dataloader):
        self.args = args
        self.dataloader = dataloader
        if not torch.cuda.is_available():
            raise RuntimeError("vfd_gan_amd trainers need a HIP device (no CPU fallback on the product path)")
        self.device = torch.device("cuda", torch.cuda.current_device())
        self.global_step = 0
        self.epoch = 0
        self.best_roc = 0
        self.best_pr = 0
        self.best_f1 = 0
        self.color_video_dict = OrderedDict()
        self.gray_video_dict = OrderedDict()
        self.errors_dict = OrderedDict()
        self.score_dict = OrderedDict()
        self.rank, self.world_size = vdist.rank(), vdist.world_size()

        current_time = datetime.now().strftime("%b%d_%H-%M-%S")
        comment = "b{}xd{}xwh{}_lr{}".format(args.batchsize, args.nfr, args.isize, args.lr)        # reference :38
        self.save_root_dir = os.path.join(args.result_root, args.model, comment, current_time)
        self.weight_dir = os.path.join(self.save_root_dir, 'weights')
        self.writer = None
        if self.rank == 0:
            os.makedirs(self.weight_dir, exist_ok=True)
            os.makedirs(os.path.join(self.save_root_dir, "runs"), exist_ok=True)
            self.writer = ScalarLog(os.path.join(self.save_root_dir, "runs"))
            with open(self.save_root_dir + "/args.txt", mode="w") as f:
                json.dump(args.__dict__, f, indent=4)
            print("\n SAVE PATH == {} \n".format(self.save_root_dir))

        if args.model == "c2plus1d":                                                                 # reference :55-69
            from ..models.mystcnn import AutoEncoder
            model = AutoEncoder()
        elif args.model == "xception":
            from ..models.xception import Xception
            model = Xception()
        else:
            raise NotImplementedError("model %r: only c2plus1d and xception of the supervised baselines are built" % (args.model,))
        self.model = model.to(self.device)
        self.model.apply(weights_init)
        vdist.broadcast_module(self.model)

        self.loss = F.bce_loss
        self.opt = hoptim.Adam(self.model.parameters(), lr=args.lr, betas=(args.beta1, 0.999))
        self.reducer = vdist.GradReducer.for_optimizer(self.opt)
        resume = getattr(args, "resume", "") or ""
        if resume != "":                                                                            # reference :80-88
            if not os.path.exists(resume):
                raise IOError("Model weights not found")
            state_dict = torch.load(resume, map_location=self.device, weights_only=True)['state_dict']
            self.model.load_state_dict(fix_model_state_dict(state_dict))
            F.invalidate_weight_cache()
            vdist.broadcast_module(self.model)

    def set_input(self, data):
        self.input, self.real, self.gt, self.lb = (d.to(self.device, non_blocking=True) for d in data)
        self.input_cl, self.gt_cl = F.to_cl(self.input), F.to_cl(self.gt)

    def optimize_params(self):
        """reference :103-108"""
        F.dropout_begin_step(self.device)
        self.opt.zero_grad()
        self.predict = self.model(self.input_cl)
        self.err = self.loss(self.predict, self.gt_cl)
        self.err.backward()
        self.reducer.finish()
        self.opt.step()
        self.errors_dict.update({'loss/err/train': self.err})

    def errors(self):
        return {k: float(v.detach()) if torch.is_tensor(v) else float(v) for k, v in self.errors_dict.items()}

    def train(self):
        for self.epoch in range(self.args.ep):
            self.model.train()
            for i, data in enumerate(self.dataloader['train']):
                self.global_step += 1
                self.set_input(data)
                self.optimize_params()
                if self.global_step % self.args.freq == 0:
                    self.test()
                    self.update_summary()
            if self.rank == 0:
                print("[TRAIN Epoch %d/%d] step %d %s" % (self.epoch + 1, self.args.ep, self.global_step, self.errors()))
        if self.rank == 0:
            print("Training model Done.")

    def update_summary(self):
        if self.writer is None:
            return
        for t, e in self.errors().items():
            spk = t.rsplit('/', 1)
            self.writer.add_scalars(spk[0], {spk[1]: e}, self.global_step)
        for t, s in self.score_dict.items():
            self.writer.add_scalar(t, s, self.global_step)

    def save_weights(self, head, score):
        """reference :136-140 (it formats the score with %04d, i.e. truncated to an integer: kept)"""
        if self.rank != 0:
            return None
        path = '%s/%s-%04d_step%04d.pth' % (self.weight_dir, head, score, self.global_step)
        torch.save({'epoch': self.global_step + 1, 'state_dict': self.model.state_dict()}, path)
        return path

    def test(self):
        """reference :143-197: eval mode (running BatchNorm statistics, no Dropout), loss and post-processed prediction per test
        batch, ROC / PR / F1 over all test pixels, checkpoint when ROC (else PR) improves.  The reference leaves the net in
        eval mode until the next epoch starts; training mode is restored here (the step needs batch statistics)."""
        from .evaluate import evaluate
        was_training = self.model.training
        self.model.eval()
        errs, predicts, gts = [], [], []
        try:
            with torch.no_grad():
                for i, data in enumerate(self.dataloader['test']):
                    input_, real_, gt_, lb_ = (d.to(self.device, non_blocking=True) for d in data)
                    predict_cl = self.model(F.to_cl(input_))
                    predict_ = predict_cl.to_torch()
                    t_pre_ = threshold(predict_)
                    m_pre_ = morphology_proc(t_pre_)
                    gts.append(gt_.permute(0, 2, 3, 4, 1))
                    predicts.append(m_pre_.permute(0, 2, 3, 4, 1))
                    errs.append(self.loss(predict_cl, F.to_cl(gt_)))
                    self.color_video_dict.update({'test/input-real': torch.cat([input_, real_], dim=3)})
                    self.gray_video_dict.update({'test/mask-pre-th-mor': torch.cat([gt_, predict_, t_pre_, m_pre_], dim=3)})
                errs_np = torch.stack([e.detach().float().reshape(()) for e in errs]).cpu().numpy().astype(np.float64)
                gts_np = np.asarray(torch.stack(gts).cpu().numpy(), dtype=np.int32).flatten()
                pre_np = np.asarray(torch.stack(predicts).cpu().numpy()).flatten()
        finally:
            if was_training:
                self.model.train()
        saveto = self.save_root_dir if self.rank == 0 else None
        roc = evaluate(gts_np, pre_np, self.best_roc, self.epoch, saveto, metric='roc')
        pr = evaluate(gts_np, pre_np, self.best_pr, self.epoch, saveto, metric='pr')
        f1 = evaluate(gts_np, pre_np, metric='f1_score')
        if roc > self.best_roc:
            self.best_roc = roc
            self.save_weights('ROC', self.best_roc)
        elif pr > self.best_pr:
            self.best_pr = pr
            self.save_weights('PR', self.best_pr)
        self.errors_dict.update({'loss/err/test': float(np.mean(errs_np))})
        self.score_dict.update({"score/roc": roc, "score/pr": pr, "score/f1": f1})
        return {"roc": roc, "pr": pr, "f1": f1}
