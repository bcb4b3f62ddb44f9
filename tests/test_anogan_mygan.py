"""GPU parity of the anogan and mygan nets / training steps: against the CPU oracle on small generalised
geometries (f32 tight, bf16 stated tolerance) and against the reference's golden vectors at the reference's own
16x128x128 geometry (SURVEY.md sections 0, 8c)."""
import types

import pytest
import torch

from golden_util import check_errs, check_summary, load_golden
from util import relerr, relrms

pytestmark = pytest.mark.gpu
JS, NPZ = load_golden()

# bf16 gates AGAINST THE bf16-FAITHFUL ORACLE (oracle/vfd_oracle/bf16.py: the float32 oracle with a rounding at every tensor the
# HIP path stores; tests/test_bf16_faithful.py holds the per-kernel evidence).  Measured with tools/probe/bf16_parity.py
# (profiles/r03_bf16_parity.txt): losses 1e-3 .. 9e-3, outputs 3e-3 .. 8e-3 relative RMS; against the plain float32 oracle the
# same quantities sit at 1e-2 .. 5e-2.  Whole-net gradients are NOT tightened by the faithful oracle (single-ulp differences are
# amplified x10-20 per stage in both directions: test_bf16_faithful.py) and keep bounds of that size.
BF16_LOSS_TOL = 1e-2
BF16_OUT_TOL = 1e-2
BF16_GRAD_TOL = 0.35
BF16_GRAD_MEDIAN_TOL = 0.2


def _args(tmp, model, B, T, S, **kw):
    d = dict(batchsize=B, nfr=T, isize=S, ich=3, lr=2e-5, beta1=0.5, w_adv=1, w_con=10, pos_weight=2, freq=10 ** 9, ep=1,
             model=model, result_root=str(tmp), gpu=[0], ae=False)
    d.update(kw)
    return types.SimpleNamespace(**d)


def _p0(m):
    for mm in m.modules():
        if isinstance(mm, torch.nn.Dropout):
            mm.p = 0.0


def _mask_provider():
    from vfd_oracle.weights import seeded_tensor
    return lambda shape, p, idx: seeded_tensor(tuple(shape), 1000 + idx, 0.0, 1.0) >= p


def _compare_state(model_sd, ref_sd, f32, lr, steps, skip=()):
    for (k, v), (_, r) in zip(model_sd.items(), ref_sd.items()):
        if any(s in k for s in skip):
            continue
        if "num_batches_tracked" in k:
            assert int(v) == int(r), k
        elif "running_" in k:
            assert relerr(v, r) < (1e-3 if f32 else 6e-2), (k, relerr(v, r))
        else:
            d = (v.detach().cpu().double() - r.detach().double()).abs()
            assert float(d.mean()) <= (0.1 if f32 else 1.0) * lr * steps, (k, float(d.mean()))


# ---------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16], ids=["f32", "bf16"])
def test_anogan_step_small(dt, dev, tmp_path):
    from vfd_gan_amd import functional as F
    from vfd_gan_amd.models import anogan as HA
    from vfd_oracle import anogan as OA
    from vfd_oracle.weights import fill_module, seeded_normal, seeded_tensor
    F.set_compute_dtype(dt)
    B, T, S = 3, 8, 32
    og, od = fill_module(OA.NetG(T, S), 1).train(), fill_module(OA.NetD(T, S), 2).train()
    _p0(og)
    model = HA.AnoGAN(_args(tmp_path, "anogan", B, T, S), None)
    model.netg.load_state_dict(og.state_dict())
    model.netd.load_state_dict(od.state_dict())
    _p0(model.netg)
    F.invalidate_weight_cache()
    from vfd_oracle import bf16 as OB
    g_opt, d_opt = OA.make_optimizers(og, od, 2e-5)
    f32 = dt == torch.float32
    ng, nd = (og, od) if f32 else (OB.Faithful(og), OB.Faithful(od))      # bf16: the bf16-faithful oracle
    for it in range(2):
        z, real = seeded_normal((B, 100), 10 + it), seeded_tensor((B, 3, T, S, S), 20 + it)
        ref, fake_ref = OA.step(ng, nd, g_opt, d_opt, real if f32 else OB.rbf(real), z if f32 else OB.rbf(z))
        model.set_input((real, real, real[:, :1], torch.ones(B, T)))
        model.z = z.to(dev)
        model.optimize_params()
        got = model.errors()
        for k, v in ref.items():
            g = got["%s/%s/train" % (k[4], k)]
            assert abs(g - v) <= ((1e-4 if it == 0 else 2e-4) if f32 else (2e-2 if it == 0 else 6e-2)) * max(abs(v), 1e-3), (it, k, g, v)
        # step 0 compares the same weights; later steps also carry Adam's sign-amplified rounding noise of the
        # previous update (every weight moves ~lr whatever its gradient's size), hence the RMS metric there
        if f32 and it == 0:
            assert relerr(model.gen_fake.to_torch(), fake_ref) < 5e-4, it
        else:
            assert relrms(model.gen_fake.to_torch(), fake_ref) < (5e-3 if f32 else (BF16_OUT_TOL if it == 0 else 4e-2)), (it, relrms(model.gen_fake.to_torch(), fake_ref))
    # A bias that feeds straight into a training-mode BatchNorm has an exactly-zero true gradient (the batch mean
    # absorbs it); what each implementation computes there is rounding noise, which Adam turns into +-lr moves.
    # The reference's own values for those entries are noise, so they are not compared.
    _compare_state(model.netd.state_dict(), od.state_dict(), f32, 2e-5, 2,
                   skip=("layer1.0.bias", "layer1.4.bias", "layer2.1.bias", "layer2.5.bias"))
    _compare_state(model.netg.state_dict(), og.state_dict(), f32, 1e-4, 2,
                   skip=("layer1.0.bias", "layer2.2.bias", "layer2.7.bias", "layer3.2.bias"))


def test_anogan_reference_geometry_golden(dev, tmp_path):
    """16x128x128, B=2, float32: HIP step against the vectors the reference's own NetG / NetD produced."""
    from vfd_gan_amd import functional as F
    from vfd_gan_amd.models import anogan as HA
    from vfd_oracle.weights import fill_module, seeded_normal, seeded_tensor
    F.set_compute_dtype(torch.float32)
    R = JS["anogan"]
    model = HA.AnoGAN(_args(tmp_path, "anogan", 2, 16, 128), None)
    assert list(model.netg.state_dict().keys()) == R["keys_g"] and list(model.netd.state_dict().keys()) == R["keys_d"]
    fill_module(model.netg, R["seeds"]["g"])
    fill_module(model.netd, R["seeds"]["d"])
    F.invalidate_weight_cache()
    z, real = seeded_normal((2, 100), R["seeds"]["z"]), seeded_tensor((2, 3, 16, 128, 128), R["seeds"]["real"])
    # forward with the imposed dropout masks of the fixture
    F.set_dropout_mask_provider(_mask_provider())
    try:
        out = model.netg(F.to_cl(z.to(dev)))
    finally:
        F.set_dropout_mask_provider(None)
    check_summary(out.to_torch(), R["fwd_masked"]["fake"], 5e-4, "masked fake")
    fill_module(model.netg, R["seeds"]["g"])     # running stats were advanced by the forward above
    F.invalidate_weight_cache()
    _p0(model.netg)
    model.set_input((real, real, real[:, :1], torch.ones(2, 16)))
    model.z = z.to(dev)
    model.optimize_params()
    got = model.errors()
    check_errs({k: got["%s/%s/train" % (k[4], k)] for k in R["step_p0"]["errs"]}, R["step_p0"]["errs"], 1e-4)
    check_summary(model.gen_fake.to_torch(), R["step_p0"]["fake"], 1e-3, "fake")
    for k, ref in R["after1"]["d"].items():
        if "running" in k:
            check_summary(model.netd.state_dict()[k], ref, 2e-3, k)


# ---------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16], ids=["f32", "bf16"])
def test_mygan_step_small(dt, dev, tmp_path):
    from vfd_gan_amd import functional as F
    from vfd_gan_amd.models import mygannet as HM
    from vfd_oracle import mygannet as OM
    from vfd_oracle.weights import fill_module, seeded_tensor
    F.set_compute_dtype(dt)
    B, T, S = 2, 16, 64
    og, od = fill_module(OM.NetG(), 3).train(), fill_module(OM.NetD(OM.make_args(T, S)), 4).train()
    _p0(og)
    model = HM.MyGAN(_args(tmp_path, "mygan", B, T, S), None)
    model.netg.load_state_dict(og.state_dict())
    model.netd.load_state_dict(od.state_dict())
    _p0(model.netg)
    F.invalidate_weight_cache()
    from vfd_oracle import bf16 as OB
    opt_g, opt_d = OM.make_optimizers(og, od)
    f32 = dt == torch.float32
    ng, nd = (og, od) if f32 else (OB.Faithful(og), OB.Faithful(od))      # bf16: the bf16-faithful oracle
    for it in range(2):
        inp = seeded_tensor((B, 3, T, S, S), 30 + it)
        gt = (seeded_tensor((B, 1, T, S, S), 40 + it, 0.0, 1.0) > 0.97).float()
        gf, pf = seeded_tensor((B, 3, T, S, S), 50 + it), seeded_tensor((B, 3, T, S, S), 60 + it)
        if f32:
            ref, pred_ref = OM.step(ng, nd, opt_g, opt_d, inp, gt, gf, pf)
        else:
            ref, pred_ref = OM.step(ng, nd, opt_g, opt_d, OB.rbf(inp), gt, OB.rbf(gf), OB.rbf(pf))
        model.set_input((inp, inp, gt, torch.ones(B, T)), gt_flow=gf, pre_flow=pf)
        model.optimize_params()
        got = model.errors()
        for k, v in ref.items():
            g = got["%s/%s/train" % (k[4], k)]
            assert abs(g - v) <= ((1e-4 if it == 0 else 2e-4) if f32 else (1.5e-2 if it == 0 else 6e-2)) * max(abs(v), 1e-3), (it, k, g, v)
        if f32 and it == 0:
            assert relerr(model.predict.to_torch(), pred_ref) < 5e-4, it
        else:
            assert relrms(model.predict.to_torch(), pred_ref) < (5e-3 if f32 else (BF16_OUT_TOL if it == 0 else 4e-2)), (it, relrms(model.predict.to_torch(), pred_ref))
    # (2+1)D conv biases all feed a BatchNorm: zero true gradient, see test_anogan_step_small
    _compare_state(model.netg.state_dict(), og.state_dict(), f32, 2e-5, 2, skip=("_conv.bias",))
    _compare_state(model.netd.state_dict(), od.state_dict(), f32, 2e-5, 2, skip=("_conv.bias",))


def test_mygan_reference_geometry_golden(dev, tmp_path):
    """16x128x128, B=2, float32: HIP nets / step against the vectors of the reference's own NetG / NetD / losses."""
    from vfd_gan_amd import functional as F
    from vfd_gan_amd.models import mygannet as HM
    from vfd_oracle.weights import fill_module, seeded_tensor
    F.set_compute_dtype(torch.float32)
    R = JS["mygan"]
    s = R["seeds"]
    model = HM.MyGAN(_args(tmp_path, "mygan", 2, 16, 128), None)
    assert list(model.netg.state_dict().keys()) == R["keys_g"] and list(model.netd.state_dict().keys()) == R["keys_d"]
    fill_module(model.netg, s["g"])
    fill_module(model.netd, s["d"])
    F.invalidate_weight_cache()
    inp = seeded_tensor((2, 3, 16, 128, 128), s["inp"])
    gt = (seeded_tensor((2, 1, 16, 128, 128), s["gt"], 0.0, 1.0) > 0.97).float()
    gf, pf = seeded_tensor((2, 3, 16, 128, 128), s["gt_flow"]), seeded_tensor((2, 3, 16, 128, 128), s["pre_flow"])
    F.set_dropout_mask_provider(_mask_provider())
    try:
        out = model.netg(F.to_cl(inp[:1].to(dev)))
    finally:
        F.set_dropout_mask_provider(None)
    check_summary(out.to_torch(), R["fwd_masked"]["predict"], 5e-4, "masked predict")
    # the U-Net is size-agnostic: 16x112x112 matches the reference too
    fill_module(model.netg, s["g"])
    F.invalidate_weight_cache()
    _p0(model.netg)
    out112 = model.netg(F.to_cl(seeded_tensor((1, 3, 16, 112, 112), 67).to(dev)))
    check_summary(out112.to_torch(), R["fwd_112"]["predict"], 5e-4, "predict@112")
    fill_module(model.netg, s["g"])
    F.invalidate_weight_cache()
    model.set_input((inp, inp, gt, torch.ones(2, 16)), gt_flow=gf, pre_flow=pf)
    model.optimize_params()
    got = model.errors()
    check_errs({k: got["%s/%s/train" % (k[4], k)] for k in R["step_p0"]["errs"]}, R["step_p0"]["errs"], 1e-4)
    check_summary(model.predict.to_torch(), R["step_p0"]["predict"], 1e-3, "predict")


def test_ganomaly_golden(dev, tmp_path):
    """ganomaly at a power-of-two size against the reference's own Encoder/Decoder/NetG/NetD vectors (3 steps)."""
    from vfd_gan_amd import functional as F
    from vfd_gan_amd.models import ganomaly as HG
    from vfd_oracle.weights import fill_module, seeded_tensor
    F.set_compute_dtype(torch.float32)
    R = JS["ganomaly"]
    cfg = R["cfg"]
    args = _args(tmp_path, "ganomaly", 2, 4, cfg["isize"], lr=2e-4, w_con=50)
    model = HG.Ganomaly(args, None, opt=HG.make_opt(isize=cfg["isize"], ngf=cfg["ngf"]))
    assert list(model.netg.state_dict().keys()) == R["keys_g"] and list(model.netd.state_dict().keys()) == R["keys_d"]
    fill_module(model.netg, R["seeds"]["g"])
    fill_module(model.netd, R["seeds"]["d"])
    F.invalidate_weight_cache()
    for it in range(3):
        x = seeded_tensor((8, 3, 32, 32), 30 + it)
        clip = x.view(2, 4, 3, 32, 32).permute(0, 2, 1, 3, 4).contiguous()      # 8 frames = 2 clips x 4 frames
        model.set_input((clip, clip, clip[:, :1], torch.ones(2, 4)))
        model.optimize_params(check_collapse=False)
        got = model.errors()
        check_errs({k: got["%s/%s/train" % (k[4], k)] for k in R["steps"][it]["errs"]}, R["steps"][it]["errs"], 2e-4, "step %d" % it)
        check_summary(model.fake.to_torch(), R["steps"][it]["fake"], 1e-3, "fake %d" % it)


def test_anogan_generalised_112(dev, tmp_path):
    """BASELINE configs[2] geometry: anogan at 16x112x112 (seed volume (512,2,14,14), Linear(256*2*14*14, 1)), B=2.
    The reference's NetD does not exist at 112 (fixture anogan.netd112_raises), so the check is HIP vs the oracle's
    generalisation, which reduces to the reference modules at 128 (test_anogan_reference_geometry_golden): ONE oracle
    step, compared with the float32 path (losses 1e-4 relative: north_star's tolerance) and with the bf16 path as
    benchmarked (5e-2) on the same weights, noise and clips."""
    from vfd_gan_amd import functional as F
    from vfd_gan_amd.models import anogan as HA
    from vfd_oracle import anogan as OA
    from vfd_oracle.weights import fill_module, seeded_normal, seeded_tensor
    assert JS["anogan"]["netd112_raises"]
    B, T, S = 2, 16, 112
    og, od = fill_module(OA.NetG(T, S), 71).train(), fill_module(OA.NetD(T, S), 72).train()
    _p0(og)
    sd_g, sd_d = {k: v.clone() for k, v in og.state_dict().items()}, {k: v.clone() for k, v in od.state_dict().items()}
    z, real = seeded_normal((B, 100), 73), seeded_tensor((B, 3, T, S, S), 74)
    from vfd_oracle import bf16 as OB
    g_opt, d_opt = OA.make_optimizers(og, od, 2e-5)
    ref32, fake32 = OA.step(og, od, g_opt, d_opt, real, z)
    # bf16 as benchmarked: against the bf16-faithful oracle on the same starting weights (2e-2: two clips only — the BCE terms are
    # means over 2 sigmoid outputs of a 100k-feature Linear; measured 9.4e-3; frames 1e-2, measured 3.0e-3)
    og.load_state_dict(sd_g)
    od.load_state_dict(sd_d)
    ref16, fake16 = OA.step(OB.Faithful(og), OB.Faithful(od), *OA.make_optimizers(og, od, 2e-5), OB.rbf(real), OB.rbf(z))
    for dt, tol in ((torch.float32, 1e-4), (torch.bfloat16, 2e-2)):
        ref, fake_ref = (ref32, fake32) if dt == torch.float32 else (ref16, fake16)
        F.set_compute_dtype(dt)
        model = HA.AnoGAN(_args(tmp_path, "anogan", B, T, S), None)
        assert model.netg.seed_shape == (512, 2, 14, 14) and model.netd.fc[0].in_features == 256 * 2 * 14 * 14
        model.netg.load_state_dict(sd_g)
        model.netd.load_state_dict(sd_d)
        _p0(model.netg)
        F.invalidate_weight_cache()
        model.set_input((real, real, real[:, :1], torch.ones(B, T)))
        model.z = z.to(dev)
        model.optimize_params()
        got = model.errors()
        for k, v in ref.items():
            g = got["%s/%s/train" % (k[4], k)]
            assert abs(g - v) <= tol * max(abs(v), 1e-3), (dt, k, g, v)
        if dt == torch.float32:
            assert relerr(model.gen_fake.to_torch(), fake_ref) < 5e-4
        else:
            assert relrms(model.gen_fake.to_torch(), fake_ref) < BF16_OUT_TOL, relrms(model.gen_fake.to_torch(), fake_ref)
        del model
        torch.cuda.empty_cache()
    F.set_compute_dtype(torch.bfloat16)


def test_mygan_netg_224_golden(dev, tmp_path):
    """BASELINE configs[3] geometry: NetG at 16x224x224 (B=1) on the HIP path against the vector the REFERENCE's own
    NetG produced there (fixture mygan224); float32 tight, bf16 (as benchmarked) at the stated tolerance.  The reference's
    NetD does not exist at 224 (it is locked to 16x128x128), so the discriminators at 224 are covered by the oracle's
    generalisation only (bench.py --model mygan runs them; the 128 geometry is pinned by test_mygan_reference_geometry_golden)."""
    from vfd_gan_amd import functional as F
    from vfd_gan_amd.models import mygannet as HM
    from vfd_oracle.weights import fill_module, seeded_tensor
    R = JS["mygan224"]
    x = seeded_tensor((1, 3, 16, 224, 224), R["seeds"]["inp"])
    outs = {}
    for dt, tol in ((torch.float32, 5e-4), (torch.bfloat16, 4e-2)):
        F.set_compute_dtype(dt)
        netg = HM.NetG(3).to(dev).train()
        fill_module(netg, R["seeds"]["g"])
        _p0(netg)
        F.invalidate_weight_cache()
        with torch.no_grad():
            out = netg(F.to_cl(x.to(dev))).to_torch()
        assert tuple(out.shape) == (1, 1, 16, 224, 224)
        outs[dt] = out.cpu()
        if dt == torch.float32:
            check_summary(out, R["predict"], tol, "predict@224")
        else:
            assert relrms(out, outs[torch.float32]) < tol, relrms(out, outs[torch.float32])
        del netg
        torch.cuda.empty_cache()
    F.set_compute_dtype(torch.bfloat16)


@pytest.mark.parametrize("which", ["anogan", "mygan"])
def test_graph_replay_equals_eager_3d(which, dev, tmp_path):
    """hipGraph replay of the anogan / mygan step is the same arithmetic as the eager step (float32, no atomics: bit for bit),
    over several replays — in particular the packed filter copies a net uses at the START of a step must be the ones its
    Adam update at the END of the previous replay produced (AnoGAN's netD), not copies cached before the capture."""
    from vfd_gan_amd import functional as F
    from vfd_gan_amd.graph import GraphedStep
    from vfd_gan_amd.lib.data import synthetic_batch
    F.set_compute_dtype(torch.float32)
    if which == "anogan":
        from vfd_gan_amd.models.anogan import AnoGAN as M
        B, T, S = 2, 8, 16
    else:
        from vfd_gan_amd.models.mygannet import MyGAN as M
        B, T, S = 2, 16, 64
    models = []
    for i in range(2):
        torch.manual_seed(21)
        torch.cuda.manual_seed(21)
        m = M(_args(tmp_path / str(i), which, B, T, S), None)
        _p0(m.netg)
        if which == "anogan":
            m.z = torch.randn(B, 100, generator=torch.Generator().manual_seed(9)).to(dev)
        models.append(m)
    a, b = models
    for (ka, va), (kb, vb) in zip(a.netg.state_dict().items(), b.netg.state_dict().items()):
        assert torch.equal(va, vb), ka
    batch0, batch1 = synthetic_batch(B, T, S, 3, seed=300), synthetic_batch(B, T, S, 3, seed=301)
    a.set_input(batch0)
    for _ in range(2):
        a.optimize_params()
    a.set_input(batch1)
    for _ in range(3):
        a.optimize_params()
    b.set_input(batch0)
    step = GraphedStep(b, warmup=2).capture()
    step.load_input(batch1)
    for _ in range(3):
        step.replay()
    ea, eb = a.errors(), b.errors()
    for k in ea:
        assert ea[k] == eb[k], (k, ea[k], eb[k])
    for net in ("netg", "netd"):
        for (k, v), (_, r) in zip(getattr(a, net).state_dict().items(), getattr(b, net).state_dict().items()):
            assert torch.equal(v, r), (net, k)


def test_mygan_step_224_configs3(dev, tmp_path):
    """BASELINE configs[3] AS A STEP: one full MyGAN.optimize_params at 16x224x224, B=1 — NetG, the generalised SDisc / TDisc
    (Linear(1024*3*3, 1) / Linear(128*2, 1); the reference's NetD is locked to 128, fixture mygan.netd_locked), every backward
    kernel at this geometry and both Adam updates — against the oracle on identical weights, clips and flow streams:
    float32 to north_star's 1e-4 on the 12 loss scalars (+ NetG / NetD gradients), bf16 (as benchmarked) against the
    bf16-faithful oracle (oracle/vfd_oracle/bf16.py)."""
    from vfd_gan_amd import functional as F
    from vfd_gan_amd.models import mygannet as HM
    from vfd_oracle import bf16 as OB
    from vfd_oracle import mygannet as OM
    from vfd_oracle.weights import fill_module, seeded_tensor
    B, T, S = 1, 16, 224
    inp = seeded_tensor((B, 3, T, S, S), 230)
    gt = (seeded_tensor((B, 1, T, S, S), 240, 0.0, 1.0) > 0.97).float()
    gf, pf = seeded_tensor((B, 3, T, S, S), 250), seeded_tensor((B, 3, T, S, S), 260)
    report = {}
    for dt in (torch.float32, torch.bfloat16):
        f32 = dt == torch.float32
        og, od = fill_module(OM.NetG(), 3).train(), fill_module(OM.NetD(OM.make_args(T, S)), 4).train()
        _p0(og)
        sd_g, sd_d = {k: v.clone() for k, v in og.state_dict().items()}, {k: v.clone() for k, v in od.state_dict().items()}
        opt_g, opt_d = OM.make_optimizers(og, od)
        if f32:
            ref, pred_ref = OM.step(og, od, opt_g, opt_d, inp, gt, gf, pf)
        else:
            ref, pred_ref = OM.step(OB.Faithful(og), OB.Faithful(od), opt_g, opt_d, OB.rbf(inp), gt, OB.rbf(gf), OB.rbf(pf))
        F.set_compute_dtype(dt)
        model = HM.MyGAN(_args(tmp_path, "mygan", B, T, S), None)
        assert model.netd.spatdisc.linear.in_features == 1024 * 3 * 3 and model.netd.tempdisc.linear.in_features == 128 * 2
        model.netg.load_state_dict(sd_g)
        model.netd.load_state_dict(sd_d)
        _p0(model.netg)
        F.invalidate_weight_cache()
        model.set_input((inp, inp, gt, torch.ones(B, T)), gt_flow=gf, pre_flow=pf)
        model.optimize_params()
        got = model.errors()
        bad = {}
        for k, v in ref.items():
            g = got["%s/%s/train" % (k[4], k)]
            # bf16, SDisc terms (3e-2): its input is the sparse 0/1 mask, most positions of its first conv output hold ONE value
            # (the bias) and BatchNorm puts that plateau wherever the few other positions leave the mean — a 1-ulp difference
            # of the plateau moves every position at once (measured 1.2e-2 at 224, 5e-3 at 64)
            tol = 1e-4 if f32 else (3e-2 if k.endswith("_s") or k in ("err_d_real", "err_d_fake", "err_d", "err_g_adv") else BF16_LOSS_TOL)
            if not abs(g - v) <= tol * max(abs(v), 1e-3):
                bad["loss " + k] = (g, v)
        e = relrms(model.predict.to_torch(), pred_ref)
        if not e < (2e-4 if f32 else BF16_OUT_TOL):
            bad["predict"] = e
        # gradients of both nets at this geometry (conv biases that feed a BatchNorm have a zero true gradient: rounding noise).
        # float32: the BatchNorm parameters of the FIRST blocks sum their gradient over 800k positions behind a ReLU / LeakyReLU
        # kink, and SDisc's input is the sparse 0/1 mask, whose first conv output is the same value at most positions: a
        # 1e-6 forward difference moves whole plateaus across a kink (measured 2.7e-3 .. 5.4e-3 there, 1e-5 .. 1e-3 elsewhere).
        # TDisc ends in a global average over 224 x 224 positions: every position receives the SAME upstream gradient, which
        # the BatchNorm backward (g - mean(g) - xh mean(g xh)) cancels down to the part the LeakyReLU pattern leaves — its
        # gradients are a small difference of large terms in either implementation (measured 1.3e-2 .. 2.6e-2 in float32).
        errs = []
        for (k, p), (_, r) in list(zip(model.netg.named_parameters(), og.named_parameters())) + list(zip(model.netd.named_parameters(), od.named_parameters())):
            if "_conv.bias" in k or float(r.grad.abs().max()) < 1e-9:
                continue
            e = relrms(p.grad, r.grad)
            errs.append(e)
            gate = (5e-2 if k.startswith("tempdisc") else 1e-2) if f32 else (0.7 if k.startswith("spatdisc") else BF16_GRAD_TOL)
            if not f32 and "dconv1.conv.bn." in k:
                # bf16, the BatchNorm INSIDE the first (2+1)D block of NetG / SDisc: its input is the 3-channel clip (resp. the
                # sparse 0/1 mask) through ONE 14-channel conv, i.e. plateaus of near-equal values over 800k positions in
                # front of a ReLU kink; one bf16 ulp of the batch mean moves whole plateaus across the kink.  Measured against
                # the faithful oracle over builds of round 3 that differ only in the float32 rounding of the statistics:
                # 0.27 .. 0.41 (NetG), 0.55 .. 0.70 (SDisc).  No per-parameter bf16 gate for these four tensors: the
                # float32 pass of this test gates them at 1e-2 and the median over all parameters below still counts them.
                continue
            if not e < gate:
                bad["grad " + k] = e
        errs.sort()
        if not errs[len(errs) // 2] < (2e-3 if f32 else BF16_GRAD_MEDIAN_TOL):
            bad["median gradient error"] = errs[len(errs) // 2]
        report[str(dt)] = bad
        del model
        torch.cuda.empty_cache()
    F.set_compute_dtype(torch.bfloat16)
    assert not any(report.values()), report
