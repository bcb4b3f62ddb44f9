#!/usr/bin/env python3
"""Step time of the three models on synthetic clips (eager; one GPU) with a kernel-time table from HIP events on the
MFMA kernels.  The headline benchmark is bench.py (ganomaly); this is the sanity check of BASELINE configs 3-4."""
import argparse
import os
import sys
import tempfile
import time
import types

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from vfd_gan_amd import functional as F  # noqa: E402
from vfd_gan_amd.lib.data import synthetic_batch  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--model", default="anogan")
ap.add_argument("--batch", type=int, default=8)
ap.add_argument("--isize", type=int, default=128)
ap.add_argument("--nfr", type=int, default=16)
ap.add_argument("--steps", type=int, default=3)
ap.add_argument("--dtype", default="bf16")
ap.add_argument("--layers", action="store_true")
ap.add_argument("--graph", action="store_true", help="replay a captured hipGraph of the step")
a = ap.parse_args()
F.set_compute_dtype(a.dtype)
args = types.SimpleNamespace(batchsize=a.batch, nfr=a.nfr, isize=a.isize, ich=3, lr=2e-5, beta1=0.5, w_adv=1, w_con=10, pos_weight=2,
                             freq=10 ** 9, ep=1, model=a.model, result_root=tempfile.mkdtemp(), gpu=[0], ae=False)
if a.model == "anogan":
    from vfd_gan_amd.models.anogan import AnoGAN as M
elif a.model == "mygan":
    from vfd_gan_amd.models.mygannet import MyGAN as M
else:
    from vfd_gan_amd.models.ganomaly import Ganomaly as M
m = M(args, None)
m.set_input(synthetic_batch(a.batch, a.nfr, a.isize, 3, seed=1))
if a.graph:
    from vfd_gan_amd.graph import GraphedStep
    step = GraphedStep(m, warmup=2).capture()
    for _ in range(2):
        step.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        step.replay()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / a.steps
    print("%s %dx%dx%d batch %d %s [hipGraph]: %.1f ms/step = %.1f clips/s ; peak mem %.1f GB ; losses %s" % (
        a.model, a.nfr, a.isize, a.isize, a.batch, a.dtype, dt * 1e3, a.batch / dt, torch.cuda.max_memory_allocated() / 2 ** 30,
        {k: round(v, 4) for k, v in m.errors().items()}))
    sys.exit(0)
for _ in range(2):
    m.optimize_params()
torch.cuda.synchronize()
timer = F.KernelTimer()
F.set_kernel_timer(timer)
t0 = time.perf_counter()
for _ in range(a.steps):
    m.optimize_params()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / a.steps
F.set_kernel_timer(None)
print("%s %dx%dx%d batch %d %s: %.1f ms/step = %.1f clips/s ; peak mem %.1f GB" % (a.model, a.nfr, a.isize, a.isize, a.batch, a.dtype,
      dt * 1e3, a.batch / dt, torch.cuda.max_memory_allocated() / 2 ** 30))
tot = 0.0
for k, v in sorted(timer.summary().items()):
    print("  %-34s x%-4d %8.2f ms/step %8.1f TF/s" % (k, v["launches"] // a.steps, v["ms"] / a.steps, v["flops"] / max(v["ms"], 1e-9) / 1e9))
    tot += v["ms"] / a.steps
print("  MFMA kernels total %.2f ms/step" % tot)
if a.layers:
    for (name, geom), d in sorted(timer.by_geometry().items(), key=lambda kv: -kv[1]["ms"])[:25]:
        print("  %-30s %-66s x%-3d %8.1f us %7.1f TF/s" % (name, geom, d["launches"] // a.steps, d["ms"] * 1e3 / d["launches"],
                                                        d["flops"] / max(d["ms"], 1e-9) / 1e9))
