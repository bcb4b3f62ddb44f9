"""Worker for the data-parallel GPU test: N ranks share cuda:0 and talk over gloo (RCCL needs one GPU per rank, so
the collectives themselves are exercised by the driver's multi-GPU bench; this checks the step logic around them:
bucket hooks, phase graphs, reductions between graphs, 1/world gradient scaling folded into Adam).

Every rank steps its OWN clips (seed 77 + rank) and, for anogan, its own noise: a reduction that is missing, issued too
early, mis-sliced (bucket offsets) or that mixes ranks changes rank 0's result.  The expected result comes from mode
"emulate2": ONE process, no process group, two replicas of the model stepping shard 0 and shard 1 through the model's own
step_program() with the gradient arenas summed by hand at each ("reduce", reducer) entry (per-replica BatchNorm statistics,
1/2 folded into Adam) — what two ranks compute, without any collective.  Right after every reduction the ranks also check
that their gradient arenas are identical (all_gather of a checksum), which a reduction issued after join() would fail."""
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


def build(shard):
    """The model of rank `shard` with that rank's clips (and noise) loaded."""
    torch.manual_seed(3)        # identical initial weights on every rank / replica
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
        model.z = torch.randn(B, 100, generator=torch.Generator().manual_seed(5 + shard)).cuda()    # per-rank noise
    else:
        from vfd_gan_amd.models import mygannet as HM
        B, T, S = 2, 16, 64
        args = types.SimpleNamespace(batchsize=B, nfr=T, isize=S, ich=3, lr=2e-5, beta1=0.5, w_adv=1, w_con=10, pos_weight=2,
                                     freq=10 ** 9, ep=1, model="mygan", result_root=tempfile.mkdtemp(), gpu=[0], ae=False)
        model = HM.MyGAN(args, None)
    for mm in model.netg.modules():
        if isinstance(mm, torch.nn.Dropout):
            mm.p = 0.0
    model.set_input(synthetic_batch(B, T, S, 3, seed=77 + shard))      # every rank steps its OWN clips
    return model


def eager(model):
    if which == "ganomaly":
        model.optimize_params(check_collapse=False)
    else:
        model.optimize_params()


def arena_check(reducer):
    """After a reduction every rank must hold the SAME gradient arena (and a finite one)."""
    if world < 2:
        return
    torch.cuda.synchronize()
    a = reducer.arena.detach().double()
    sig = torch.stack([a.sum(), a.abs().sum(), (a * torch.arange(a.numel(), device=a.device, dtype=torch.float64)).sum()]).cpu()
    sigs = [torch.zeros_like(sig) for _ in range(world)]
    torch.distributed.all_gather(sigs, sig)
    assert torch.isfinite(sig).all() and float(sig[1]) > 0, sig
    for s in sigs[1:]:
        assert torch.equal(s, sigs[0]), (sigs, "gradient arenas differ between ranks after the reduction")


STEPS = 5      # eager: 5 steps; graph: 2 warm-up + 3 replays; emulate2: 5 program walks
if mode == "emulate2":
    assert world == 1
    reps = [build(0), build(1)]
    progs = [m.step_program() for m in reps]
    for m in reps:
        for o in (getattr(m, "optimizer_g", None) or m.g_opt, getattr(m, "optimizer_d", None) or m.d_opt):
            o.grad_scale = 0.5
    for _ in range(STEPS):
        for entries in zip(*progs):
            kind = entries[0][0]
            if kind == "graph":
                for _, fn in entries:
                    fn()
                    F.join_side_stream()
            elif kind == "reduce":
                F.join_side_stream()
                total = entries[0][1].arena + entries[1][1].arena
                for _, r in entries:
                    r.arena.copy_(total)
    model = reps[0]
else:
    model = build(rank)
    if world > 1:
        # check the arenas right where the optimiser consumes them: wrap both optimisers' step()
        for o, r in ((getattr(model, "optimizer_g", None) or model.g_opt, model.reducer_g),
                     (getattr(model, "optimizer_d", None) or model.d_opt, model.reducer_d)):
            def wrapped(closure=None, _o=o, _r=r, _step=o.step):
                if not torch.cuda.is_current_stream_capturing():
                    arena_check(_r)
                return _step()
            o.step = wrapped
    if mode == "graph":
        step = GraphedStep(model, warmup=2).capture()
        for _ in range(STEPS - 2):
            step.replay()
            if world > 1:
                arena_check(model.reducer_d)      # replays run no Python inside the step: check after it (netD's update is the last phase)
    else:
        for _ in range(STEPS):
            eager(model)
torch.cuda.synchronize()
if rank == 0:
    sd = {k: v.detach().cpu().double().sum().item() for k, v in model.netg.state_dict().items() if v.dtype.is_floating_point}
    sd.update({"D." + k: v.detach().cpu().double().sum().item() for k, v in model.netd.state_dict().items() if v.dtype.is_floating_point})
    json.dump({"errors": model.errors(), "sums": sd, "world": world}, open(out, "w"))
if vdist.collectives_on():
    vdist.barrier()
    torch.distributed.destroy_process_group()
