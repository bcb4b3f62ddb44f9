"""Dispatcher with the behaviour of the reference's trainer.py:7-56 for the GAN models.

``python -m vfd_gan_amd.trainer --model {mygan,anogan,ganomaly} ...`` (or under torchrun for one process per GPU:
the reference's ``--gpu 0,1`` DataParallel is replaced by RCCL data parallelism, vfd_gan_amd/dist.py).  Of the
supervised baselines (lib/train_stcnn.py) ``c2plus1d`` and ``xception`` are built on the hot path's kernels;
``clstm`` (models/convlstm.py) is not and is rejected with the reference's own "is None" message.
"""
from __future__ import print_function

from . import dist as vdist
from . import functional as F
from .lib.args import Args


def build_model(args, dataloader):
    """MODEL LOAD of reference trainer.py:16-40."""
    if args.model == 'mygan':
        from .models.mygannet import MyGAN
        return MyGAN(args, dataloader)
    if args.model == 'anogan':
        from .models.anogan import AnoGAN
        return AnoGAN(args, dataloader)
    if args.model == 'ganomaly':
        from .models.ganomaly import Ganomaly
        return Ganomaly(args, dataloader)
    if args.model in ('c2plus1d', 'xception'):          # the supervised (2+1)D / Xception baselines (reference trainer.py:29-34)
        from .lib.train_stcnn import VFD_STCNN
        return VFD_STCNN(args, dataloader)
    print("\n %s is None." % (args.model))
    exit()


def main(args):
    # -- DATA LOAD --
    from .lib.data import DataLoader
    dataloader = DataLoader(args, rank=vdist.rank()).load_data()

    # -- MODEL LOAD --
    if vdist.rank() == 0:
        print("--Load model--")
    model = build_model(args, dataloader)
    model.train()
    return model


if __name__ == '__main__':
    args = Args().parse()
    vdist.init_from_env()
    F.set_compute_dtype(args.dtype)
    main(args)
