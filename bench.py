#!/usr/bin/env python3
"""Headline benchmark: video clips/sec per GAN step on MI355X (BASELINE.json).

    python bench.py --gpus N --steps K --warmup W            [--model ganomaly|anogan|mygan]
    (N > 1: launched by the driver as  python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...; started
     plainly with --gpus N > 1 it launches exactly that itself, as a child process, before this process touches the GPU)

Default workload (config.workload): BASELINE.json configs[1] — ganomaly, 16x112x112 clips, bf16 MFMA storage with f32
accumulation / f32 master weights, batch 32 clips per GPU (= 512 frames of 3x112x112 through the 2-D nets).
``--model anogan`` = configs[2] (3-D conv G/D step, 16x112x112, batch 32), ``--model mygan`` = configs[3]'s per-GPU
workload ((2+1)D U-Net + spatial/temporal discriminators, 16x224x224, batch 8 clips per GPU).
One "step" = one full optimize_params() of the model (reference models/ganomaly.py:502-519, models/anogan.py:229-250,
models/mygannet.py:350-367).  Synthetic clips are generated on the host and are resident in HBM before
the timed region.  Data parallel (weak scaling): every rank steps its own clips, gradients are summed over RCCL
inside the step.

Rank 0 prints ONE JSON line.  `roofline` is the dominant MFMA kernel: algorithmic FLOPs / its average launch duration
from HIP events on the launch stream, taken in an eager pass of the same step right after the timed graph replays
(events cannot be recorded inside a replayed graph) with the stream kept GPU-bound (a device-side delay in front of
every step lets the host run ahead, so an event pair brackets the kernel and not the host's launch gap);
`cpu_baseline` times the oracle (the CPU restatement of the reference step, stock torch.nn float32) on this box's host
cores on a bounded sample; `recon_parity` is BASELINE.json's "recon-MSE vs ref": the loss scalars of ONE step of the HIP
path in float32 (north_star: <= 1e-4 relative) and in bf16 (reported) against that same oracle step on identical weights
and clips; `secondary` (default workload, 1 GPU) carries the anogan (configs[2]) and mygan (configs[3], per-GPU share)
clips/s and step MFMA fractions from short runs of the same harness.
"""
import argparse
import contextlib
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
MFMA_PEAK_F32_TFLOPS = 157.3
HBM_PEAK_GBS = 8000.0

# reference accounting of one step in forward-equivalents of the two nets: 1 fwd + 2 for a backward
STEP_MULT = {"ganomaly": (3, 12), "anogan": (3, 9), "mygan": (3, 10)}
DEFAULTS = {"ganomaly": dict(batch=32, isize=112, steps=100), "anogan": dict(batch=32, isize=112, steps=20),
            "mygan": dict(batch=8, isize=224, steps=20)}


def make_args(model, batch, nfr, isize, local):
    d = dict(batchsize=batch, nfr=nfr, isize=isize, ich=3, freq=10 ** 9, ep=1, model=model,
             result_root=tempfile.mkdtemp(prefix="vfd_bench_"), gpu=[local], ae=False, pos_weight=2)
    d.update(dict(lr=2e-4, beta1=0.5, w_adv=1, w_con=50) if model == "ganomaly" else dict(lr=2e-5, beta1=0.5, w_adv=1, w_con=10))
    return types.SimpleNamespace(**d)


def build_model(model, args_ns, dtype):
    from vfd_gan_amd import functional as F
    F.set_compute_dtype(dtype)
    torch.manual_seed(1234)     # identical initial weights on every rank (and broadcast again inside)
    with contextlib.redirect_stdout(sys.stderr):     # the trainer base class announces its save path (reference behaviour):
        if model == "ganomaly":                      # stdout carries the ONE JSON line only
            from vfd_gan_amd.models.ganomaly import Ganomaly as M
        elif model == "anogan":
            from vfd_gan_amd.models.anogan import AnoGAN as M
        else:
            from vfd_gan_amd.models.mygannet import MyGAN as M
        return M(args_ns, None)


def forward_flops(model_name, m):
    """Algorithmic FLOPs (2*MAC of every conv / conv-transpose / linear) of ONE forward of netG and of netD on the
    resident batch, counted by the launch timer's per-launch FLOP figures (no restated layer table)."""
    from vfd_gan_amd import functional as F
    out = []
    with torch.no_grad():
        for which in ("g", "d"):
            t = F.KernelTimer()
            F.set_kernel_timer(t)
            try:
                if model_name == "ganomaly":
                    m.netg(m.x) if which == "g" else m.netd(m.x)
                elif model_name == "anogan":
                    if which == "g":
                        m.netg(F.to_cl(torch.randn(m.args.batchsize, 100, device=m.device)))
                    else:
                        m.netd(m.real_cl)
                else:
                    if which == "g":
                        m.netg(m.input_cl)
                    else:
                        m.netd(F.gray2rgb(m.gt_cl), m.gt_flow)
            finally:
                F.set_kernel_timer(None)
            torch.cuda.synchronize()
            out.append(sum(r[1] for r in t.records))
    return out


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


def _p0(net):
    for m in net.modules():
        if isinstance(m, torch.nn.Dropout):
            m.p = 0.0


def cpu_baseline(model, isize, nfr):
    """The oracle step on the host cores, bounded sample (~10-30 s): 1 warm-up + a few timed steps on a few clips.
    Returns (baseline dict, parity context): the warm-up step doubles as the reference of recon_parity() — its loss
    scalars, the weights it started from and its clips (Dropout off on both sides: the CPU and device RNG streams differ)."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    from vfd_gan_amd.lib.data import synthetic_batch, synthetic_flow
    from vfd_oracle.weights import fill_module
    cores = host_cores()
    torch.set_num_threads(cores)
    torch.manual_seed(1234)
    ctx = {"model": model, "isize": isize, "nfr": nfr}
    snap = lambda net: {k: v.clone() for k, v in net.state_dict().items()}    # noqa: E731
    if model == "ganomaly":
        from vfd_oracle import ganomaly as OG
        clips, steps = 16, 3
        opt = OG.make_opt(isize=isize)
        og, od = fill_module(OG.NetG(opt), 7), fill_module(OG.NetD(opt), 8)
        ctx.update(sd_g=snap(og), sd_d=snap(od), clips=clips, batch=synthetic_batch(clips, nfr, isize, 3, seed=1234))
        opt_g, opt_d = OG.make_optimizers(og, od, opt)
        x = OG.fold_frames(ctx["batch"][0])
        run = lambda: OG.step(og, od, opt_g, opt_d, x, opt)[0]    # noqa: E731
    elif model == "anogan":
        from vfd_oracle import anogan as OA
        clips, steps = 2, 2
        og, od = fill_module(OA.NetG(nfr, isize), 7).train(), fill_module(OA.NetD(nfr, isize), 8).train()
        _p0(og)
        g_opt, d_opt = OA.make_optimizers(og, od, 2e-5)
        batch, z = synthetic_batch(clips, nfr, isize, 3, seed=1234), torch.randn(clips, 100)
        ctx.update(sd_g=snap(og), sd_d=snap(od), clips=clips, batch=batch, z=z)
        run = lambda: OA.step(og, od, g_opt, d_opt, batch[1], z)[0]   # noqa: E731
    else:
        from vfd_oracle import mygannet as OM
        clips, steps = 2, 2
        og, od = fill_module(OM.NetG(), 7).train(), fill_module(OM.NetD(OM.make_args(nfr, isize)), 8).train()
        _p0(og)
        opt_g, opt_d = OM.make_optimizers(og, od)
        batch = synthetic_batch(clips, nfr, isize, 3, seed=1234)
        gf, pf = synthetic_flow(clips, nfr, isize, 1), synthetic_flow(clips, nfr, isize, 2)
        ctx.update(sd_g=snap(og), sd_d=snap(od), clips=clips, batch=batch, gf=gf, pf=pf)
        run = lambda: OM.step(og, od, opt_g, opt_d, batch[0], batch[2], gf, pf)[0]    # noqa: E731
    ctx["ref"] = run()
    t0 = time.perf_counter()
    for _ in range(steps):
        run()
    dt = (time.perf_counter() - t0) / steps
    return {"value": round(clips / dt, 4), "unit": "clips/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": "oracle %s step (stock torch.nn f32 on the host cores), %d clips of %dx%dx%d, 1 warm-up + %d "
                      "timed steps (%.2f s/step)" % (model, clips, nfr, isize, isize, steps, dt)}, ctx


RECON_KEY = {"ganomaly": "err_g_con", "anogan": "err_d", "mygan": "err_g_con"}      # L1 / BCE / weighted_bce


def recon_parity(ctx, local):
    """BASELINE.json metric, second half ("recon-MSE vs ref"): ONE optimize_params of the HIP path on the weights and clips
    of the oracle's first step (cpu_baseline's warm-up step), float32 (north_star tolerance 1e-4 relative) and bf16 (the
    benchmarked storage type: deviation reported, not gated).  The oracle is the checker here, nothing more."""
    from vfd_gan_amd import functional as F
    model, ref, out = ctx["model"], ctx["ref"], {}
    prev = F.get_compute_dtype()
    try:
        for tag, dt in (("f32", torch.float32), ("bf16", torch.bfloat16)):
            m = build_model(model, make_args(model, ctx["clips"], ctx["nfr"], ctx["isize"], local), dt)
            m.netg.load_state_dict(ctx["sd_g"])
            m.netd.load_state_dict(ctx["sd_d"])
            _p0(m.netg)
            F.invalidate_weight_cache()
            if model == "mygan":
                m.set_input(ctx["batch"], gt_flow=ctx["gf"], pre_flow=ctx["pf"])
            else:
                m.set_input(ctx["batch"])
            if model == "anogan":
                m.z = ctx["z"].to(m.device)
            m.optimize_params(check_collapse=False) if model == "ganomaly" else m.optimize_params()
            got = {k.split("/")[1]: v for k, v in m.errors().items()}
            out[tag] = {k: abs(got[k] - v) / max(abs(v), 1e-12) for k, v in ref.items()}
            out[tag + "_losses"] = {k: got[k] for k in ref}
            del m
            torch.cuda.empty_cache()
    finally:
        F.set_compute_dtype(prev)
    key = RECON_KEY[model]
    return {"recon_loss": key, "cpu_oracle": round(ref[key], 8), "hip_f32": round(out["f32_losses"][key], 8),
            "recon_rel_err": float("%.3e" % out["f32"][key]), "recon_rel_err_bf16": float("%.3e" % out["bf16"][key]),
            "worst_loss_rel_err_f32": float("%.3e" % max(out["f32"].values())), "worst_loss_rel_err_bf16": float("%.3e" % max(out["bf16"].values())),
            "tolerance_f32": 1e-4,
            "sample": "one optimize_params on %d clips of %dx%dx%d, weights and clips of the oracle's first step, Dropout off"
                      % (ctx["clips"], ctx["nfr"], ctx["isize"], ctx["isize"])}


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--model", default="ganomaly", choices=["ganomaly", "anogan", "mygan"])
    ap.add_argument("--steps", type=int, default=None)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=None, help="clips per GPU")
    ap.add_argument("--isize", type=int, default=None)
    ap.add_argument("--nfr", type=int, default=16)
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "f32", "fp8"],
                    help="fp8: bf16 storage, e4m3 operands for the forward / data-gradient GEMMs of the wide layers (functional.set_fp8)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-timer", action="store_true")
    ap.add_argument("--no-graph", action="store_true", help="launch every kernel from Python each step instead of replaying a hipGraph")
    ap.add_argument("--layers", action="store_true", help="print a per-geometry table of the MFMA kernels to stderr")
    ap.add_argument("--no-secondary", action="store_true", help="skip the short anogan / mygan runs of the default line's `secondary` key")
    a = ap.parse_args(argv)
    a.explicit = {k for k in ("batch", "isize") if getattr(a, k) is not None}      # (--steps / --warmup are the driver's to choose)
    for k, v in DEFAULTS[a.model].items():
        if getattr(a, k) is None:
            setattr(a, k, v)
    return a


def run_workload(a, rank, world, local, dev):
    """Build the model, capture the step, time a.steps replays (barrier + synchronize on both sides, max over ranks) and
    take the per-kernel pass.  Returns the JSON line's dict on rank 0, None elsewhere."""
    from vfd_gan_amd import dist as vdist
    from vfd_gan_amd import functional as F
    from vfd_gan_amd.lib.data import synthetic_batch
    dtype = {"bf16": torch.bfloat16, "f32": torch.float32, "fp8": "fp8"}[a.dtype]

    model = build_model(a.model, make_args(a.model, a.batch, a.nfr, a.isize, local), dtype)
    batch = synthetic_batch(a.batch, a.nfr, a.isize, 3, seed=1234 + rank)
    model.set_input(batch)          # clips resident in HBM (layout conversion to channels-last bf16 included)
    torch.cuda.synchronize()
    g_fwd, d_fwd = forward_flops(a.model, model)
    mg, md = STEP_MULT[a.model]
    step_flops_ref = mg * g_fwd + md * d_fwd

    def eager_step():
        if a.model == "ganomaly":
            model.optimize_params(check_collapse=False)
        else:
            model.optimize_params()

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
    run = step.replay if use_graph else eager_step
    for _ in range(a.warmup):
        run()
    barrier()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        run()
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        tdist.all_reduce(t, op=tdist.ReduceOp.MAX)
        elapsed = float(t.item())
    ms = elapsed / a.steps * 1e3

    timer_steps = 0
    event_overhead_us = 0.0
    if not a.no_kernel_timer:      # every rank takes part (the eager step issues the gradient collectives)
        # per-kernel pass, after the timed region: eager launches of the SAME step with HIP events around every MFMA
        # kernel.  Eager mode is host-bound (Python issues a launch every ~5 us), so a device-side delay of about one
        # eager step's host time goes in front of each step: the host runs ahead, the launches queue up behind the
        # delay and then execute back to back, and an event pair measures the kernel, not the launch gap.
        timer_steps = 3
        eager_step()
        torch.cuda.synchronize()
        h0 = time.perf_counter()
        eager_step()
        host_s = time.perf_counter() - h0       # host time of one eager step (launch-bound)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        torch.cuda._sleep(20_000_000)           # calibrate the delay kernel's tick on this device
        e1.record()
        torch.cuda.synchronize()
        ticks_per_s = 20_000_000 / max(e0.elapsed_time(e1) * 1e-3, 1e-6)
        cyc = int(max(host_s, ms * 1e-3) * 1.3 * ticks_per_s)
        timer = F.KernelTimer()
        empty = []          # event pairs with NOTHING between them, in the same queued state: the marker-to-marker latency
        for _ in range(timer_steps):
            torch.cuda._sleep(cyc)
            for _ in range(16):
                ea, eb = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                ea.record()
                eb.record()
                empty.append((ea, eb))
            F.set_kernel_timer(timer)
            eager_step()
            F.set_kernel_timer(None)
            torch.cuda.synchronize()
        ev = sorted(a_.elapsed_time(b_) * 1e3 for a_, b_ in empty)
        event_overhead_us = ev[len(ev) // 2]
    if world > 1:
        vdist.barrier()

    out = None
    if rank == 0:
        clips_s = world * a.batch * a.steps / elapsed
        peak = MFMA_PEAK_F32_TFLOPS if a.dtype == "f32" else MFMA_PEAK_BF16_TFLOPS      # (fp8 mode: priced against bf16, most FLOPs stay bf16)
        pyr, c_ = [], a.isize
        while c_ >= 8 and c_ % 2 == 0 and (not pyr or pyr[-1] > 7):
            pyr.append(c_)
            c_ //= 2
        if c_ not in pyr:
            pyr.append(c_)
        pyr = "-".join(str(v) for v in pyr)
        gname = ("ganomaly %dx%dx%d %s batch=%d clips/GPU (BASELINE.json configs[%s]): frames folded to (%d,3,%d,%d), generalised pyramid %s, "
                 "nz=100 ngf=64, full optimize_params (G fwd, D fwd on input and on fake, backward_g, Adam, backward_d, Adam)"
                 % (a.nfr, a.isize, a.isize, a.dtype, a.batch, "1" if (a.isize, a.nfr) == (112, 16) else "4 geometry" if a.isize == 224 else "-",
                    a.batch * a.nfr, a.isize, a.isize, pyr))
        names = {"ganomaly": gname,
                 "anogan": "anogan 16x%dx%d %s batch=%d clips/GPU (BASELINE.json configs[2]): 3-D conv NetG (seed volume 512x2x%dx%d) / "
                           "NetD, full optimize_params (D on real, G fwd, D on fake, Adam(D), D on G(z), Adam(G))"
                           % (a.isize, a.isize, a.dtype, a.batch, a.isize // 8, a.isize // 8),
                 "mygan": "mygan 16x%dx%d %s batch=%d clips/GPU (BASELINE.json configs[3], per-GPU workload): (2+1)D U-Net NetG + "
                          "spatial/temporal NetD, full optimize_params; flow streams are synthetic inputs of the step"
                          % (a.isize, a.isize, a.dtype, a.batch)}
        out = {
            "metric": "video clips/sec (Nx3x16x%dx%d) per GAN step" % (a.isize, a.isize), "value": round(clips_s, 3), "unit": "clips/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(ms, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": a.dtype, "data": "synthetic",
            "config": {"workload": names[a.model], "global_batch": world * a.batch, "frames_per_clip": a.nfr,
                       "parallelism": "dp%d" % world,
                       "launch": "hipGraph replay of the captured step" if use_graph else "eager (one Python launch per kernel)"},
            # the reference's accounting: every forward and backward pass of both nets, including the netD gradient work of
            # the generator update that the reference computes and then discards (this build skips it: DESIGN.md 2.3)
            "step_reference_tflop": round(step_flops_ref / 1e12, 3),
            "step_reference_mfma_frac": round(step_flops_ref / (ms * 1e-3) / 1e12 / peak, 4),
            "losses": {k: round(v, 6) for k, v in model.errors().items()},
        }
        if timer is not None:
            summ = timer.summary()
            executed = sum(v["flops"] for v in summ.values()) / timer_steps
            mfma_ms = sum(v["ms"] for v in summ.values()) / timer_steps
            # what the MFMA kernels actually execute per step (sum of the per-launch algorithmic FLOPs of every launch)
            out["step_executed_tflop"] = round(executed / 1e12, 3)
            out["step_mfma_frac"] = round(executed / (ms * 1e-3) / 1e12 / peak, 4)
            out["mfma_kernels_ms_per_step"] = round(mfma_ms, 3)
            if a.layers:
                for (name, geom), d in sorted(timer.by_geometry().items(), key=lambda kv: -kv[1]["ms"]):
                    print("%-34s %-66s x%-3d %8.1f us/launch %7.1f TF/s" % (name, geom, d["launches"] // timer_steps,
                          d["ms"] * 1e3 / d["launches"], d["flops"] / max(d["ms"], 1e-9) / 1e9), file=sys.stderr)
            name, d = max(summ.items(), key=lambda kv: kv[1]["ms"])
            # an event pair around a kernel also spans the second marker's own processing: the median EMPTY pair of the same
            # pass is subtracted (rocprofv3's kernel-trace average of the graph replay is the cross-check, profiles/)
            raw_us = d["ms"] * 1e3 / d["launches"]
            net_us = max(raw_us - event_overhead_us, 0.5 * raw_us)
            tf = d["flops"] / d["launches"] / (net_us * 1e-6) / 1e12
            traffic = None      # HBM bytes per launch of this kernel from the PMC passes (separate rocprofv3 --pmc FETCH_SIZE /
            tpaths = [os.path.join(ROOT, "profiles", "%s_%s_pmc.json" % (r_, a.model)) for r_ in ("r03", "r02")]      # WRITE_SIZE runs: tools/profile_all.sh)
            tpath = next((t_ for t_ in tpaths if os.path.exists(t_)), None)
            if a.dtype == "bf16" and tpath is not None:
                traffic = json.load(open(tpath)).get(name, {}).get("hbm_bytes_per_launch")
            out["roofline"] = {"bound": "mfma", "kernel": name, "achieved": round(tf, 2), "peak": peak, "unit": "TFLOP/s",
                               "frac": round(tf / peak, 4), "traffic": traffic,
                               "launches_per_step": d["launches"] // timer_steps,
                               "timed_in": "eager single-stream pass after the timed region, stream kept GPU-bound by a device-side delay; rocprofv3 cross-check: profiles/r03_<model>_kernel_stats_single_stream.csv",
                               "traffic_from": os.path.relpath(tpath, ROOT) if (traffic is not None and tpath) else None,
                               "avg_launch_us": round(net_us, 2), "avg_launch_us_raw_events": round(raw_us, 2),
                               "event_pair_overhead_us": round(event_overhead_us, 2),
                               "avg_launch_gflop": round(d["flops"] / d["launches"] / 1e9, 3)}
            out["kernels"] = {k: {"launches_per_step": v["launches"] // timer_steps, "ms_per_step": round(v["ms"] / timer_steps, 3),
                                  "tflops": round(v["flops"] / max(v["ms"], 1e-9) / 1e9, 2)} for k, v in sorted(summ.items())}
    return out if rank == 0 else None


def main():
    a = parse_args()
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # not under a launcher: start one rank per GPU ourselves (fresh child processes through torch.distributed.run, exactly
        # the driver's command line; THIS process has not touched the GPU and never will) and hand on rank 0's JSON line
        import socket
        import subprocess
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(a.gpus), "--master-addr", "127.0.0.1",
               "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        raise SystemExit(subprocess.call(cmd, env=env))

    from vfd_gan_amd import dist as vdist
    rank, world = vdist.init_from_env()
    local = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local % max(torch.cuda.device_count(), 1))
    dev = torch.device("cuda", torch.cuda.current_device())
    out = run_workload(a, rank, world, local, dev)
    torch.cuda.synchronize()
    torch.cuda.empty_cache()
    default_line = a.model == "ganomaly" and a.dtype == "bf16" and not a.explicit and a.nfr == 16
    if rank == 0 and world == 1 and default_line and not a.no_secondary:
        # the 3-D workloads of BASELINE.json on the same harness, bounded (10 timed replays each): configs[2] and configs[3]
        out["secondary"] = {}
        for name in ("anogan", "mygan"):
            b = parse_args(["--model", name, "--steps", "10", "--warmup", "2", "--no-cpu-baseline", "--no-secondary"])
            r = run_workload(b, rank, world, local, dev)
            torch.cuda.synchronize()
            torch.cuda.empty_cache()
            out["secondary"][name] = {"workload": r["config"]["workload"], "value": r["value"], "unit": r["unit"], "ms_per_step": r["ms_per_step"],
                                      "steps": r["steps"], "step_mfma_frac": r.get("step_mfma_frac"),
                                      "step_reference_mfma_frac": r["step_reference_mfma_frac"],
                                      "dominant_kernel": {k: r["roofline"][k] for k in ("kernel", "achieved", "frac", "avg_launch_us")} if "roofline" in r else None}
    if rank == 0:
        if world == 1 and not a.no_cpu_baseline:
            out["cpu_baseline"], ctx = cpu_baseline(a.model, a.isize, a.nfr)
            out["recon_parity"] = recon_parity(ctx, local)
            out["recon_rel_err"] = out["recon_parity"]["recon_rel_err"]
        print(json.dumps(out), flush=True)
    if vdist.is_initialized():      # world > 1, or the single-rank RCCL rehearsal (VFD_DIST_SINGLE=1)
        vdist.barrier()
        tdist.destroy_process_group()


if __name__ == "__main__":
    main()
