"""Worker for the data-parallel GPU test: N ranks share cuda:0 and talk over gloo (RCCL needs one GPU per rank, so
the collectives themselves are exercised by the driver's multi-GPU bench; this checks the step logic around them:
bucket hooks, phase graphs, reductions between graphs, 1/world gradient scaling folded into Adam)."""
import json
import os
import sys
import tempfile
import types

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from vfd_gan_amd import dist as vdist, functional as F  # noqa: E402
from vfd_gan_amd.graph import GraphedStep  # noqa: E402
from vfd_gan_amd.lib.data import synthetic_batch  # noqa: E402

mode, out = sys.argv[1], sys.argv[2]
which = sys.argv[3] if len(sys.argv) > 3 else "ganomaly"
if os.environ.get("VFD_DIST_SINGLE") == "1":
    rank, world = vdist.init_from_env(backend="nccl")       # one rank, RCCL: every collective issued for real (identity)
    assert vdist.collectives_on() and torch.distributed.get_backend() == "nccl"
else:
    rank, world = vdist.init_from_env(backend="gloo") if int(os.environ.get("WORLD_SIZE", "1")) > 1 else (0, 1)
torch.cuda.set_device(0)
F.set_compute_dtype(torch.float32)
torch.manual_seed(3)
if which == "ganomaly":
    from vfd_gan_amd.models import ganomaly as HG
    B, T, S = 2, 2, 32
    args = types.SimpleNamespace(batchsize=B, nfr=T, isize=S, ich=3, lr=2e-4, beta1=0.5, w_adv=1, w_con=50, freq=10 ** 9, ep=1,
                                 model="ganomaly", result_root=tempfile.mkdtemp(), gpu=[0])
    model = HG.Ganomaly(args, None, opt=HG.make_opt(isize=S, ngf=8))
elif which == "anogan":
    from vfd_gan_amd.models import anogan as HA
    B, T, S = 2, 8, 16
    args = types.SimpleNamespace(batchsize=B, nfr=T, isize=S, ich=3, lr=2e-5, beta1=0.5, w_adv=1, w_con=10, pos_weight=2,
                                 freq=10 ** 9, ep=1, model="anogan", result_root=tempfile.mkdtemp(), gpu=[0], ae=False)
    model = HA.AnoGAN(args, None)
    for mm in model.netg.modules():
        if isinstance(mm, torch.nn.Dropout):
            mm.p = 0.0
    model.z = torch.randn(B, 100, generator=torch.Generator().manual_seed(5)).cuda()    # same noise on every rank
else:
    from vfd_gan_amd.models import mygannet as HM
    B, T, S = 2, 16, 64
    args = types.SimpleNamespace(batchsize=B, nfr=T, isize=S, ich=3, lr=2e-5, beta1=0.5, w_adv=1, w_con=10, pos_weight=2,
                                 freq=10 ** 9, ep=1, model="mygan", result_root=tempfile.mkdtemp(), gpu=[0], ae=False)
    model = HM.MyGAN(args, None)
    for mm in model.netg.modules():
        if isinstance(mm, torch.nn.Dropout):
            mm.p = 0.0
batch = synthetic_batch(B, T, S, 3, seed=77)      # every rank steps the SAME clips: averaged gradients == 1-rank gradients
model.set_input(batch)


def eager():
    if which == "ganomaly":
        model.optimize_params(check_collapse=False)
    else:
        model.optimize_params()


if mode == "graph":
    step = GraphedStep(model, warmup=2).capture()
    for _ in range(3):
        step.replay()
else:
    for _ in range(5):
        eager()
torch.cuda.synchronize()
if rank == 0:
    sd = {k: v.detach().cpu().double().sum().item() for k, v in model.netg.state_dict().items() if v.dtype.is_floating_point}
    sd.update({"D." + k: v.detach().cpu().double().sum().item() for k, v in model.netd.state_dict().items() if v.dtype.is_floating_point})
    json.dump({"errors": model.errors(), "sums": sd, "world": world}, open(out, "w"))
if vdist.collectives_on():
    vdist.barrier()
    torch.distributed.destroy_process_group()
