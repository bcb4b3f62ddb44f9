#!/usr/bin/env python3
"""Headline benchmark: video clips/sec per GAN step on MI355X (BASELINE.json).

    python bench.py --gpus N --steps K --warmup W
    (N > 1: launched by the driver as  python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...)

Workload (config.workload): BASELINE.json configs[1] — ganomaly, 16x112x112 clips, bf16 MFMA storage with f32
accumulation / f32 master weights, batch 32 clips per GPU (= 512 frames of 3x112x112 through the 2-D nets).
One "step" = one full optimize_params(): G forward, 4 D forwards, backward_g, Adam(G), backward_d, Adam(D)
(reference models/ganomaly.py:502-519).  Synthetic clips are generated on the host and are resident
in HBM before the timed region.  Data parallel (weak scaling): every rank steps its own 32 clips, gradients are
summed over RCCL inside the step.

Rank 0 prints ONE JSON line.  `roofline` is measured live with HIP events around every launch of the dominant
MFMA kernel during the timed steps; `cpu_baseline` times the oracle (the CPU restatement of the reference step,
stock torch.nn float32) on this box's host cores on a bounded sample.
"""
import argparse
import json
import os
import sys
import tempfile
import time
import types

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as tdist  # noqa: E402

MFMA_PEAK_BF16_TFLOPS = 2500.0   # dense bf16 MFMA peak, MI355X_MICROARCH.md "Chip-level parameters"
HBM_PEAK_GBS = 8000.0


def ganomaly_step_flops(isize, frames, ngf=64, nz=100, nc=3):
    """Algorithmic FLOPs (2*MAC of every conv / conv-transpose) of ONE ganomaly step on `frames` frames:
    3*G + 12*D forward-equivalents (1 G fwd + 2 for its backward; 4 D fwd + 4 D backward passes of 2 each),
    the reference's own accounting."""
    def enc(nz_out):
        f, c, s = 0.0, ngf, isize // 2
        f += 2.0 * s * s * ngf * nc * 16
        floor = isize // 2
        while floor >= 8 and floor % 2 == 0:
            floor //= 2
        while s > floor:
            s //= 2
            f += 2.0 * s * s * (2 * c) * c * 16
            c *= 2
        f += 2.0 * nz_out * c * s * s
        return f
    g = 2 * enc(nz) + enc(nz)   # the decoder mirrors the encoder
    d = enc(1)
    return frames * (3 * g + 12 * d)


def build_model(args_ns, dtype):
    from vfd_gan_amd import functional as F
    from vfd_gan_amd.models.ganomaly import Ganomaly
    F.set_compute_dtype(dtype)
    torch.manual_seed(1234)     # identical initial weights on every rank (and broadcast again inside)
    import contextlib
    with contextlib.redirect_stdout(sys.stderr):     # the trainer base class announces its save path (reference behaviour):
        return Ganomaly(args_ns, None)               # stdout carries the ONE JSON line only


def host_cores():
    """CPU cores this process may actually use: affinity mask, capped by the cgroup CPU quota (the GPU box hands a
    1-GPU job a 16-core share of a 256-thread host; oversubscribing torch's pool 16x makes the baseline ~10x slower)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            quota, period = f.read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return min(n, int(os.environ.get("VFD_CPU_BASELINE_CORES", "16")))


def cpu_baseline(isize, nfr, steps=3, clips=16):
    """The oracle step on the host cores, bounded sample (~10-20 s): `clips` clips, 1 warm-up + `steps` timed steps."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    from vfd_oracle import ganomaly as OG
    from vfd_gan_amd.lib.data import synthetic_batch
    cores = host_cores()
    torch.set_num_threads(cores)
    torch.manual_seed(1234)
    opt = OG.make_opt(isize=isize)
    og, od = OG.NetG(opt), OG.NetD(opt)
    opt_g, opt_d = OG.make_optimizers(og, od, opt)
    x = OG.fold_frames(synthetic_batch(clips, nfr, isize, 3, seed=1234)[0])
    OG.step(og, od, opt_g, opt_d, x, opt)
    t0 = time.perf_counter()
    for _ in range(steps):
        OG.step(og, od, opt_g, opt_d, x, opt)
    dt = (time.perf_counter() - t0) / steps
    return {"value": round(clips / dt, 4), "unit": "clips/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": "oracle ganomaly step (stock torch.nn f32 on the host cores), %d clips = %d frames %dx%d, 1 warm-up + %d "
                      "timed steps (%.2f s/step)" % (clips, clips * nfr, isize, isize, steps, dt)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=32, help="clips per GPU")
    ap.add_argument("--isize", type=int, default=112)
    ap.add_argument("--nfr", type=int, default=16)
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "f32"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-timer", action="store_true")
    ap.add_argument("--no-graph", action="store_true", help="launch every kernel from Python each step instead of replaying a hipGraph")
    ap.add_argument("--layers", action="store_true", help="print a per-geometry table of the MFMA kernels to stderr")
    a = ap.parse_args()

    from vfd_gan_amd import dist as vdist
    from vfd_gan_amd import functional as F
    from vfd_gan_amd.lib.data import synthetic_batch

    world = int(os.environ.get("WORLD_SIZE", "1"))
    if a.gpus > 1 and world == 1:
        raise SystemExit("bench.py --gpus %d must be launched with torch.distributed.run (one process per GPU)" % a.gpus)
    rank, world = vdist.init_from_env()
    local = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local % max(torch.cuda.device_count(), 1))
    dev = torch.device("cuda", torch.cuda.current_device())
    dtype = torch.bfloat16 if a.dtype == "bf16" else torch.float32

    ns = types.SimpleNamespace(batchsize=a.batch, nfr=a.nfr, isize=a.isize, ich=3, lr=2e-4, beta1=0.5, w_adv=1, w_con=50,
                               freq=10 ** 9, ep=1, model="ganomaly", result_root=tempfile.mkdtemp(prefix="vfd_bench_"), gpu=[local])
    model = build_model(ns, dtype)
    batch = synthetic_batch(a.batch, a.nfr, a.isize, 3, seed=1234 + rank)
    model.set_input(batch)          # clips resident in HBM (layout conversion to channels-last bf16 included)
    torch.cuda.synchronize()

    def barrier():
        vdist.barrier()
        torch.cuda.synchronize()

    use_graph = not a.no_graph
    timer = None
    if use_graph:
        from vfd_gan_amd.graph import GraphedStep
        ok = 1
        try:
            step = GraphedStep(model, warmup=max(a.warmup, 2)).capture()      # eager warm-up steps + one captured step
        except Exception as e:  # noqa: BLE001  (capture problems must not take the benchmark down)
            ok = 0
            print("rank %d: hipGraph capture failed (%s: %s) - falling back to eager launches" % (rank, type(e).__name__, e),
                  file=sys.stderr, flush=True)
            torch.cuda.synchronize()
        if world > 1:      # every rank must take the same path, or their collectives no longer match
            flag = torch.tensor([ok], dtype=torch.int32, device=dev)
            tdist.all_reduce(flag, op=tdist.ReduceOp.MIN)
            ok = int(flag.item())
        use_graph = bool(ok)
    if use_graph:
        run = step.replay
        for _ in range(a.warmup):
            run()
    else:
        run = model.optimize_params
        for _ in range(a.warmup):
            run()
        if not a.no_kernel_timer:
            timer = F.KernelTimer()
            F.set_kernel_timer(timer)
    barrier()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        run()
    barrier()
    elapsed = time.perf_counter() - t0
    F.set_kernel_timer(None)
    timer_steps = a.steps
    if use_graph and not a.no_kernel_timer:
        # HIP events cannot be recorded inside a replayed graph: the per-kernel roofline is taken from an eager pass
        # of the SAME step (same kernels, same launches) right after the timed region
        timer_steps = min(a.steps, 5)
        timer = F.KernelTimer()
        F.set_kernel_timer(timer)
        for _ in range(timer_steps):
            model.optimize_params()
        torch.cuda.synchronize()
        F.set_kernel_timer(None)
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        tdist.all_reduce(t, op=tdist.ReduceOp.MAX)
        elapsed = float(t.item())

    if rank == 0:
        ms = elapsed / a.steps * 1e3
        clips_s = world * a.batch * a.steps / elapsed
        step_flops = ganomaly_step_flops(a.isize, a.batch * a.nfr)
        out = {
            "metric": "video clips/sec (Nx3x16x112x112) per GAN step", "value": round(clips_s, 3), "unit": "clips/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(ms, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": a.dtype, "data": "synthetic",
            "config": {"workload": "ganomaly 16x112x112 %s batch=%d clips/GPU (BASELINE.json configs[1]): frames folded "
                                   "to (%d,3,%d,%d), generalised pyramid 112-56-28-14-7, nz=100 ngf=64, full "
                                   "optimize_params (G fwd, 4 D fwd, backward_g, Adam, backward_d, Adam)"
                                   % (a.dtype, a.batch, a.batch * a.nfr, a.isize, a.isize),
                       "global_batch": world * a.batch, "frames_per_clip": a.nfr, "parallelism": "dp%d" % world,
                       "launch": "hipGraph replay of the captured step" if use_graph else "eager (one Python launch per kernel)"},
            "step_algorithmic_tflop": round(step_flops / 1e12, 3),
            "step_mfma_frac": round(step_flops / (ms * 1e-3) / 1e12 / MFMA_PEAK_BF16_TFLOPS, 4),
            "losses": {k: round(v, 6) for k, v in model.errors().items()},
        }
        if timer is not None and a.layers:
            for (name, geom), d in sorted(timer.by_geometry().items(), key=lambda kv: -kv[1]["ms"]):
                print("%-34s %-62s x%-3d %8.1f us/launch %7.1f TF/s" % (name, geom, d["launches"] // timer_steps,
                      d["ms"] * 1e3 / d["launches"], d["flops"] / max(d["ms"], 1e-9) / 1e9), file=sys.stderr)
        if timer is not None:
            summ = timer.summary()
            dom = max(summ.items(), key=lambda kv: kv[1]["ms"])
            name, d = dom
            tf = d["flops"] / (d["ms"] * 1e-3) / 1e12
            peak = MFMA_PEAK_BF16_TFLOPS if a.dtype == "bf16" else 157.3
            traffic = None      # HBM bytes per launch from the PMC passes (separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE
            tpath = os.path.join(ROOT, "profiles", "r01_traffic.json")   # runs folded by tools/pmc_traffic.py)
            if a.dtype == "bf16" and os.path.exists(tpath):
                traffic = json.load(open(tpath)).get(name, {}).get("hbm_bytes_per_launch")
            out["roofline"] = {"bound": "mfma", "kernel": name, "achieved": round(tf, 2), "peak": peak, "unit": "TFLOP/s",
                               "frac": round(tf / peak, 4), "traffic": traffic,
                               "launches_per_step": d["launches"] // timer_steps,
                               "timed_in": ("eager pass after the timed region" if use_graph else "timed region"),
                               "avg_launch_us": round(d["ms"] * 1e3 / d["launches"], 2),
                               "avg_launch_gflop": round(d["flops"] / d["launches"] / 1e9, 3)}
            out["kernels"] = {k: {"launches_per_step": v["launches"] // timer_steps, "ms_per_step": round(v["ms"] / timer_steps, 3),
                                  "tflops": round(v["flops"] / max(v["ms"], 1e-9) / 1e9, 2)} for k, v in sorted(summ.items())}
        if world == 1 and not a.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(a.isize, a.nfr)
        print(json.dumps(out), flush=True)
    if world > 1:
        vdist.barrier()
        tdist.destroy_process_group()


if __name__ == "__main__":
    main()
