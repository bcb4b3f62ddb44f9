"""Flag system with the flags and defaults of the reference's lib/args.py:8-39, plus the build's own switches.

Differences, all additive: ``--model ganomaly`` exists; ``--dtype`` picks the MFMA storage type; ``--data
synthetic`` (default, there is no dataset here) ; ``parse()`` does not hard-fail on a host without a GPU
(the reference calls torch.cuda.set_device unconditionally, lib/args.py:52)."""
import argparse

import torch


class Args():
    def __init__(self):
        self.parser = argparse.ArgumentParser()
        self.parser.add_argument('--gpu', default='0', type=str, help='GPU number. Default=0')
        self.parser.add_argument('--ep', default=10, type=int, help='epochs for training. Default=10')

        # Path (the reference's defaults are absolute paths on its author's machine, lib/args.py:12-14)
        self.parser.add_argument('--tr_plist', default="", type=str, help='train data path list. ')
        self.parser.add_argument('--ts_plist', default="", type=str, help='test data path list. ')
        self.parser.add_argument('--result_root', default="./results", type=str, help='save any result path.')

        # Dataloader
        self.parser.add_argument('--isize', default=128, type=int, help='input frame size. Default=128')
        self.parser.add_argument('--ich', default=3, type=int, help='input channel size, RGB=3. Default=3')
        self.parser.add_argument('--nfr', default=16, type=int, help='input num frame. Default=16')
        self.parser.add_argument('--batchsize', default=4, type=int, help='input batch size. Default=4')
        self.parser.add_argument('--workers', default=4, type=int, help='num_workers. Default=4')

        # Network
        self.parser.add_argument('--model', default="mygan", type=str,
                                 help='train model: mygan | anogan | ganomaly | c2plus1d | xception. Default=mygan')

        # Train
        self.parser.add_argument('--lr', default=2e-5, type=float, help='initial learning rate for adam. Default=2e-5')
        self.parser.add_argument('--beta1', default=0.5, type=float, help='momentum term of adam. Default=0.5')
        self.parser.add_argument('--w_adv', default=1, type=int, help='adversarial loss weight. Default=1')
        self.parser.add_argument('--w_con', default=10, type=int, help='reconstruction loss weight. Default=10')
        self.parser.add_argument('--pos_weight', default=2, type=int, help='weighted BCE parameter. Default=2')
        self.parser.add_argument('--freq', default=50, type=int,
                                 help='frequency of update tensorboard and test. Default=50')
        self.parser.add_argument('--resume', default="", type=str, help='Pretrained Model weight path for training')
        self.parser.add_argument('--ae', default=False, action="store_true",
                                 help='Use AutoEncoder on c2plus1d net as Generator')

        # build-only switches
        self.parser.add_argument('--dtype', default="bf16", choices=["bf16", "f32", "fp8"],
                                 help='activation/filter storage type of the HIP kernels (accumulation is always f32)')
        self.parser.add_argument('--data', default="synthetic", choices=["synthetic"],
                                 help='clip source (video decode is out of scope, SURVEY.md section 2 row 9)')
        self.parser.add_argument('--steps_per_epoch', default=8, type=int, help='synthetic batches per epoch')

    def parse(self, argv=None):
        ns = self.parser.parse_args(argv)
        # "--gpu 0,1" -> [0, 1]; negative entries mean "no device" and are dropped (reference lib/args.py:44-49)
        ns.gpu = [g for g in (int(tok) for tok in ns.gpu.split(',')) if g >= 0]
        if ns.gpu and torch.cuda.is_available():
            first = ns.gpu[0]
            torch.cuda.set_device(first if first < torch.cuda.device_count() else 0)
        self.args = ns
        return ns
