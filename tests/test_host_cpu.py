"""CPU tests of the host side: the C-ABI library loads and exports every symbol include/vfdgan_hip.h declares, the
ctypes binding covers all of them, the product path refuses to run without the device (no fallback), the flag system
matches the reference's defaults, state_dict keys match the reference's, and the RCCL reducer logic is exercised with
two gloo processes."""
import ctypes
import os
import re
import socket
import subprocess
import sys
import types

import pytest
import torch

from golden_util import load_golden

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
JS, _ = load_golden()


def _header_symbols():
    txt = open(os.path.join(ROOT, "include", "vfdgan_hip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(vfd_[a-z0-9_]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol():
    from vfd_gan_amd import _lib
    syms = _header_symbols()
    assert len(syms) >= 30
    lib = ctypes.CDLL(_lib.LIB_PATH)          # built by __graft_entry__.build() / make -C vfd_gan_amd/csrc
    for s in syms:
        assert hasattr(lib, s), "libvfdgan_hip.so does not export %s" % s
    assert sorted(_lib.SIGNATURES) == syms, (set(syms) ^ set(_lib.SIGNATURES))
    loaded = _lib.load()
    assert loaded.vfd_abi_version() == 1
    # argument checking happens before any launch: usable without a GPU
    assert loaded.vfd_adam_step(None, None, None, None, 0, 0.1, 0.5, 0.999, 1e-8, 1, 1.0, None) == -1
    assert b"adam" in loaded.vfd_last_error()
    d = _lib.ConvDesc()
    nbytes = ctypes.c_size_t()
    assert loaded.vfd_conv_workspace(ctypes.byref(d), 0, ctypes.byref(nbytes)) == -1      # all-zero descriptor rejected


def test_conv_workspace_and_wgrad_split_are_host_computable():
    from vfd_gan_amd import _lib, functional as F
    lib = _lib.load()
    # ganomaly NetD classifier at 112: 512 frames x (512ch 7x7) -> 1: long-K, few pixels -> split-K workspace
    d = F._make_desc(512, (1, 7, 7), 512, (1, 1, 1), 1, (1, 7, 7), (1, 1, 1), (0, 0, 0), False, torch.bfloat16)
    need = ctypes.c_size_t()
    assert lib.vfd_conv_workspace(ctypes.byref(d), 0, ctypes.byref(need)) == 0 and need.value > 0
    # a big pyramid layer needs none
    d2 = F._make_desc(512, (1, 56, 56), 64, (1, 28, 28), 128, (1, 4, 4), (1, 2, 2), (0, 1, 1), False, torch.bfloat16)
    assert lib.vfd_conv_workspace(ctypes.byref(d2), 0, ctypes.byref(need)) == 0 and need.value == 0
    ns, nb = ctypes.c_int32(), ctypes.c_size_t()
    assert lib.vfd_wgrad_workspace(ctypes.byref(d2), ctypes.byref(ns), ctypes.byref(nb)) == 0
    assert ns.value >= 1 and nb.value == ns.value * 128 * 16 * 64 * 4
    # inconsistent geometry is rejected with a message
    d3 = F._make_desc(2, (1, 8, 8), 4, (1, 5, 5), 4, (1, 3, 3), (1, 1, 1), (0, 1, 1), False, torch.float32)
    assert lib.vfd_conv_workspace(ctypes.byref(d3), 0, ctypes.byref(need)) == -1 and b"inconsistent" in lib.vfd_last_error()


def test_product_path_has_no_cpu_fallback():
    from vfd_gan_amd import _lib, functional as F
    with pytest.raises(_lib.HipLibraryError):
        F.to_cl(torch.zeros(1, 3, 4, 4))
    from vfd_gan_amd.models import ganomaly as HG
    net = HG.NetD(HG.make_opt(isize=32, ngf=8))
    with pytest.raises(_lib.HipLibraryError):
        net(torch.zeros(2, 3, 32, 32))
    # nothing under the product package imports the oracle
    for dp, _, fs in os.walk(os.path.join(ROOT, "vfd_gan_amd")):
        for f in fs:
            if f.endswith(".py"):
                assert "vfd_oracle" not in open(os.path.join(dp, f)).read(), f


def test_args_match_reference_defaults():
    from vfd_gan_amd.lib.args import Args
    a = Args().parse([])
    for k, v in JS["args_defaults"].items():
        if k == "gpu":
            assert a.gpu == [0]          # parse() turns "0" into [0] (reference lib/args.py:45-50)
        else:
            assert getattr(a, k) == v, k
    a = Args().parse(["--gpu", "0,2", "--model", "ganomaly", "--dtype", "f32"])
    assert a.gpu == [0, 2] and a.model == "ganomaly" and a.dtype == "f32"


def test_trainer_rejects_unknown_model(capsys):
    from vfd_gan_amd import trainer
    args = types.SimpleNamespace(model="clstm", batchsize=1, nfr=4, isize=32, ich=3, steps_per_epoch=1)      # (the ConvLSTM baseline is not built)
    with pytest.raises(SystemExit):
        trainer.main(args)
    assert "clstm is None" in capsys.readouterr().out


def test_state_dict_keys_match_reference():
    from vfd_gan_amd.models import anogan as HA, ganomaly as HG, mygannet as HM
    from vfd_gan_amd.models.spatiotempconv import SpatioTemporalConv
    R = JS["ganomaly"]
    opt = HG.make_opt(isize=R["cfg"]["isize"], ngf=R["cfg"]["ngf"])
    assert list(HG.NetG(opt).state_dict().keys()) == R["keys_g"] and list(HG.NetD(opt).state_dict().keys()) == R["keys_d"]
    assert list(HA.NetG().state_dict().keys()) == JS["anogan"]["keys_g"]
    assert list(HA.NetD().state_dict().keys()) == JS["anogan"]["keys_d"]
    assert list(HM.NetG().state_dict().keys()) == JS["mygan"]["keys_g"]
    assert list(HM.NetD(HM.make_args()).state_dict().keys()) == JS["mygan"]["keys_d"]
    assert sum(p.numel() for p in HM.NetG().parameters()) == JS["mygan"]["n_params_g"]
    assert list(SpatioTemporalConv(3, 8, 3, padding=1).state_dict().keys()) == JS["spatiotemp"]["keys"]
    # the supervised baselines: reference models/mystcnn.py, models/xception.py
    from vfd_gan_amd.models.mystcnn import AutoEncoder
    from vfd_gan_amd.models.xception import Xception
    assert list(AutoEncoder().state_dict().keys()) == JS["baselines"]["autoencoder"]["keys"]
    ae, xc = AutoEncoder(), Xception()
    assert sum(p.numel() for p in ae.parameters()) == JS["baselines"]["autoencoder"]["n_params"]
    assert list(xc.state_dict().keys()) == JS["baselines"]["xception"]["keys"]
    assert sum(p.numel() for p in xc.parameters()) == JS["baselines"]["xception"]["n_params"]
    for key, m in JS["spatiotemp_intermed"].items():
        i, o, k = key.split(",", 2)
        assert SpatioTemporalConv(int(i), int(o), eval(k)).spatial_conv.out_channels == m
    # checkpoints saved from DataParallel carry a 'module.' prefix (reference lib/utils.py:15-22)
    from vfd_gan_amd.lib.utils import fix_model_state_dict
    sd = {"module." + k: v for k, v in HA.NetD().state_dict().items()}
    HA.NetD().load_state_dict(fix_model_state_dict(sd))


def test_weights_init_semantics():
    from vfd_gan_amd import nn as hnn
    from vfd_gan_amd.lib.utils import weights_init
    mods = {"Conv3d": hnn.Conv3d(2, 2, 3), "ConvTranspose3d": hnn.ConvTranspose3d(2, 2, 3), "Linear": hnn.Linear(4, 4),
            "BatchNorm3d": hnn.BatchNorm3d(4), "BatchNorm1d": hnn.BatchNorm1d(4), "Conv2d": hnn.Conv2d(2, 2, 3),
            "ConvTranspose2d": hnn.ConvTranspose2d(2, 2, 3), "BatchNorm2d": hnn.BatchNorm2d(4)}
    for name, m in mods.items():
        if "BatchNorm" in name:
            m.bias.data.fill_(0.5)
        before = [p.clone() for p in m.parameters()]
        weights_init(m)
        assert [bool((a != b).any()) for a, b in zip(before, m.parameters())] == JS["weights_init_touched"][name], name


def test_bucket_partition():
    from vfd_gan_amd.dist import make_buckets
    slices = [(0, 100), (128, 50), (192, 1000), (1216, 10), (1280, 300)]
    buckets, owner = make_buckets(slices, 400)
    covered = sorted(i for _, _, idx in buckets for i in idx)
    assert covered == list(range(5)) and all(o is not None for o in owner)
    for lo, hi, idx in buckets:
        for i in idx:
            assert lo <= slices[i][0] and slices[i][0] + slices[i][1] <= hi
    # reverse registration order: the LAST parameter sits in the FIRST bucket
    assert 4 in buckets[0][2]


_WORKER = r'''
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, sys.argv[1])
from vfd_gan_amd import dist as vdist
rank, world = vdist.init_from_env(backend="gloo")
torch.manual_seed(0)
lin1, lin2 = torch.nn.Linear(6, 5), torch.nn.Linear(5, 3)
params = list(lin1.parameters()) + list(lin2.parameters())
vdist.broadcast_module(lin1); vdist.broadcast_module(lin2)
total, slices = 0, []
for p in params:
    slices.append((total, p.numel())); total += (p.numel() + 63) // 64 * 64
arena = torch.zeros(total)
for p, (o, n) in zip(params, slices):
    p.grad = arena[o:o + n].view(p.shape)
red = vdist.GradReducer(params, arena, slices, bucket_mb=1e-4)
assert len(red.buckets) >= 2
xs = torch.arange(24, dtype=torch.float32).view(4, 6) / 10.0
for it in range(2):
    arena.zero_()
    x = xs[rank * 2:(rank + 1) * 2] + it
    lin2(torch.tanh(lin1(x))).pow(2).mean().backward()
    red.finish()
    # single-process reference on the concatenated batch: mean loss over 4 rows = average of the two rank means
    ref1, ref2 = torch.nn.Linear(6, 5), torch.nn.Linear(5, 3)
    ref1.load_state_dict(lin1.state_dict()); ref2.load_state_dict(lin2.state_dict())
    ref2(torch.tanh(ref1(xs + it))).pow(2).mean().backward()
    for p, r in zip(params, list(ref1.parameters()) + list(ref2.parameters())):
        assert torch.allclose(p.grad / world, r.grad, rtol=1e-5, atol=1e-7), (rank, it)
# TWO backward passes into the same gradients before the optimiser step (AnoGAN's discriminator, reference
# models/anogan.py:233-241): armed for two passes, no bucket may be reduced after the first, and the reduced
# result is the rank-sum of BOTH passes
launches = []
orig_launch = red._launch
red._launch = lambda b: (launches.append(b), orig_launch(b))[1]
arena.zero_(); red.arm(passes=2)
xa, xb = xs[rank * 2:(rank + 1) * 2], xs[rank * 2:(rank + 1) * 2] * 0.5 - 1.0
lin2(torch.tanh(lin1(xa))).pow(2).mean().backward()
assert launches == [], "a bucket was reduced after the first of two backward passes"
lin2(torch.tanh(lin1(xb))).pow(2).mean().backward()
assert sorted(launches) == list(range(len(red.buckets))), launches
red.finish()
ref1, ref2 = torch.nn.Linear(6, 5), torch.nn.Linear(5, 3)
ref1.load_state_dict(lin1.state_dict()); ref2.load_state_dict(lin2.state_dict())
(ref2(torch.tanh(ref1(xs))).pow(2).mean() + ref2(torch.tanh(ref1(xs * 0.5 - 1.0))).pow(2).mean()).backward()
for p, r in zip(params, list(ref1.parameters()) + list(ref2.parameters())):
    assert torch.allclose(p.grad / world, r.grad, rtol=1e-5, atol=1e-7), rank
red._launch = orig_launch
# graph-mode form: no hooks fire (suspended), every bucket is issued by reduce_async() and joined later
arena.zero_(); red.reset(); red.suspended = True
lin2(torch.tanh(lin1(xa))).pow(2).mean().backward()
red.suspended = False
red.reduce_async(); red.join()
ref1.zero_grad(); ref2.zero_grad()
ref2(torch.tanh(ref1(xs))).pow(2).mean().backward()
for p, r in zip(params, list(ref1.parameters()) + list(ref2.parameters())):
    assert torch.allclose(p.grad / world, r.grad, rtol=1e-5, atol=1e-7), rank
# a frozen parameter must not stall the bucket logic
params[0].requires_grad_(False); red.reset(); arena.zero_()
lin2(torch.tanh(lin1(xs[rank * 2:(rank + 1) * 2]))).sum().backward(); red.finish()
dist.barrier(); dist.destroy_process_group()
print("worker", rank, "ok")
'''


def test_grad_reducer_two_gloo_processes(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(_WORKER)
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE="2", LOCAL_RANK=str(r), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, str(script), ROOT], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    for r, p in enumerate(procs):
        out, _ = p.communicate(timeout=180)
        assert p.returncode == 0 and b"ok" in out, out.decode()[-2000:]


def test_synthetic_clips_contract():
    from vfd_gan_amd.lib.data import DataLoader, synthetic_batch
    a = types.SimpleNamespace(batchsize=2, nfr=4, isize=32, ich=3, steps_per_epoch=3)
    dl = DataLoader(a).load_data()
    assert set(dl) == {"train", "test"} and len(dl["train"]) == 3
    inp, real, gt, lb = next(iter(dl["train"]))
    assert inp.shape == real.shape == (2, 3, 4, 32, 32) and gt.shape == (2, 1, 4, 32, 32) and lb.shape == (2, 4)
    assert float(real.min()) >= -1 and float(real.max()) <= 1 and set(gt.unique().tolist()) <= {0.0, 1.0}
    assert 0.0 < float(gt.mean()) < 0.3 and not torch.equal(inp, real)
    again = synthetic_batch(2, 4, 32, 3, seed=1234)
    assert all(torch.equal(x, y) for x, y in zip(again, (inp, real, gt, lb)))


def test_ring_kernels_isa_audit():
    """The LDS-DMA ring of conv_igemm / conv_wgrad is correct only if the EMITTED instruction stream has the properties
    the hand-counted waits assume (tools/isa_audit.py): per loop iteration exactly the ring's DMAs and no other VMEM op
    (RAW: the counted vmcnt), an lgkmcnt(0) between the last LDS read and the K-step barrier (WAR), no scratch traffic,
    no compiler-inserted vmcnt(0) that drains the ring, no spills in the MFMA kernels.  Checked on the compiler's
    assembly at build level (hipcc cross-compiles without a GPU)."""
    import shutil
    import tempfile
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not available")
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import isa_audit
    rings = 0
    bad = {}
    with tempfile.TemporaryDirectory() as td:
        for src in ("conv_igemm.hip", "conv_halo.hip", "conv_wgrad.hip", "conv_wgrad_halo.hip", "conv_small.hip"):
            for r in isa_audit.audit_file(os.path.join(isa_audit.CSRC, src), td):
                rings += sum(1 for lp in r["loops"] if lp["dma"])
                v = list(r["violations"])
                if "conv_igemm_kernelI5fp8_tLi4ELi4ELi4ELi4E" in r["name"]:
                    # the 16-wave fp8 tile sits exactly at its 128-register cap (64 accumulators + 4 resident 8-register A
                    # fragments): ONE pixel-index word is parked in scratch and re-read on the tap-change path only (not in
                    # the K loop's steady state: no "inside the ring loop" finding is tolerated)
                    v = [x for x in v if x not in ("spill: 1 VGPRs, 8 B scratch", "spill: 1 VGPRs, 4 B scratch")]
                if v:
                    bad[r["name"]] = v
    assert rings >= 20, "the audit found only %d ring loops: parser out of step with the compiler's output" % rings
    assert not bad, bad


def test_bench_gpus_n_launches_one_rank_per_gpu_itself(monkeypatch):
    """`python bench.py --gpus 8 ...` with WORLD_SIZE unset must start the ranks itself: a child
    `python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 ... bench.py <same flags>`, issued
    before the parent has initialised the GPU, whose exit code becomes the parent's."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    import bench
    seen = {}

    def fake_call(cmd, env=None):
        seen["cmd"], seen["env"] = cmd, env
        return 7
    monkeypatch.setattr(subprocess, "call", fake_call)
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "8", "--steps", "5", "--warmup", "2"])
    with pytest.raises(SystemExit) as ex:
        bench.main()
    assert ex.value.code == 7
    cmd = seen["cmd"]
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"] and "--nnodes=1" in cmd
    assert cmd[cmd.index("--nproc-per-node") + 1] == "8" and cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    i = cmd.index(os.path.join(root, "bench.py"))
    assert cmd[i + 1:] == ["--gpus", "8", "--steps", "5", "--warmup", "2"]
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0" or "HSA_ENABLE_IPC_MODE_LEGACY" in os.environ
    assert not torch.cuda.is_initialized()
    # under a launcher (WORLD_SIZE set) nothing is spawned: parse_args keeps the driver's flags
    a = bench.parse_args(["--gpus", "2", "--steps", "3"])
    assert a.gpus == 2 and a.steps == 3 and a.batch == 32 and a.isize == 112 and not a.explicit
