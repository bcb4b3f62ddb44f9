"""CPU tests: the oracle (oracle/vfd_oracle, the CPU restatement of the reference hot path) against the golden
vectors produced by the reference's own classes (tests/golden/make_fixtures.py).  This is what pins parity."""
import types

import pytest
import torch
import torch.nn as nn

from golden_util import check_errs, check_summary, load_golden
from vfd_oracle import anogan as OA, ganomaly as OG, losses as OL, mygannet as OM, spatiotempconv as OS
from vfd_oracle.weights import fill_module, seeded_normal, seeded_tensor

JS, NPZ = load_golden()
RT = 2e-5   # same torch, same float32 ops: the restatement should agree to rounding


def test_loss_known_answers():
    k = JS["kat_losses"]
    x, t = torch.tensor(k["x"]), torch.tensor(k["t"])
    assert abs(OL.l2_loss(x, t).item() - k["l2_loss"]) < 1e-8 and abs(k["l2_loss"] - 0.1) < 1e-7
    assert abs(OL.weighted_bce(x, t).item() - k["weighted_bce"]) < 1e-8 and abs(k["weighted_bce"] - 0.41493162) < 1e-7
    assert abs(OL.weighted_bce(x, t, pos_weight=None).item() - k["weighted_bce_pw_none"]) < 1e-8
    assert abs(nn.BCELoss()(x, t).item() - k["bce"]) < 1e-8 and abs(k["bce"] - 0.34055042) < 1e-7
    assert list(OL.gray2rgb(torch.zeros(2, 1, 3, 4, 4)).shape) == k["gray2rgb_shape"]


def test_weights_init_touches_only_3d_conv_and_bn():
    mods = {"Conv3d": nn.Conv3d(2, 2, 3), "ConvTranspose3d": nn.ConvTranspose3d(2, 2, 3), "Linear": nn.Linear(4, 4),
            "BatchNorm3d": nn.BatchNorm3d(4), "BatchNorm1d": nn.BatchNorm1d(4), "Conv2d": nn.Conv2d(2, 2, 3),
            "ConvTranspose2d": nn.ConvTranspose2d(2, 2, 3), "BatchNorm2d": nn.BatchNorm2d(4)}
    for name, m in mods.items():
        if "BatchNorm" in name:
            m.bias.data.fill_(0.5)
        before = [p.clone() for p in m.parameters()]
        OL.weights_init(m)
        assert [bool((a != b).any()) for a, b in zip(before, m.parameters())] == JS["weights_init_touched"][name], name


def test_spatiotemporal_conv():
    for key, m in JS["spatiotemp_intermed"].items():
        i, o, k = key.split(",", 2)
        assert OS.intermed_channels(int(i), int(o), eval(k)) == m, key
    mod = fill_module(OS.SpatioTemporalConv(3, 8, 3, padding=1), JS["spatiotemp"]["seed_w"]).train()
    assert list(mod.state_dict().keys()) == JS["spatiotemp"]["keys"]
    x = seeded_tensor((2, 3, 4, 8, 8), JS["spatiotemp"]["seed_x"]).requires_grad_()
    y = mod(x)
    y.pow(2).mean().backward()
    assert torch.allclose(y, torch.from_numpy(NPZ["spatiotemp_y"]), rtol=1e-5, atol=1e-6)
    assert torch.allclose(x.grad, torch.from_numpy(NPZ["spatiotemp_gx"]), rtol=1e-4, atol=1e-8)
    assert torch.allclose(mod.spatial_conv.weight.grad, torch.from_numpy(NPZ["spatiotemp_gw_spatial"]), rtol=1e-4, atol=1e-8)
    assert torch.allclose(mod.bn.running_var, torch.from_numpy(NPZ["spatiotemp_running_var"]), rtol=1e-5)


def test_ganomaly_nets_and_three_steps():
    R = JS["ganomaly"]
    opt = OG.make_opt(**R["cfg"])
    g, d = fill_module(OG.NetG(opt), R["seeds"]["g"]).train(), fill_module(OG.NetD(opt), R["seeds"]["d"]).train()
    assert list(g.state_dict().keys()) == R["keys_g"] and list(d.state_dict().keys()) == R["keys_d"]
    assert sum(p.numel() for p in g.parameters()) == R["n_params_g"] and sum(p.numel() for p in d.parameters()) == R["n_params_d"]
    x = seeded_tensor((8, 3, 32, 32), R["seeds"]["x"])
    fake, li, lo = g(x)
    pred, feat = d(x)
    for name, t in [("fake", fake), ("latent_i", li), ("latent_o", lo), ("pred", pred), ("feat", feat)]:
        check_summary(t, R["fwd"][name], RT, name)
    assert torch.allclose(li, torch.from_numpy(NPZ["ganomaly_latent_i"]), rtol=1e-5, atol=1e-6)
    g, d = fill_module(OG.NetG(opt), R["seeds"]["g"]).train(), fill_module(OG.NetD(opt), R["seeds"]["d"]).train()
    og, od = OG.make_optimizers(g, d, opt)
    for it in range(3):
        errs, fk = OG.step(g, d, og, od, seeded_tensor((8, 3, 32, 32), 30 + it), opt)
        check_errs(errs, R["steps"][it]["errs"], 1e-5, "step %d" % it)
        check_summary(fk, R["steps"][it]["fake"], 1e-4, "fake %d" % it)
    for k, ref in R["after3"]["g"].items():
        check_summary(g.state_dict()[k], ref, 2e-4, k)
    for k, ref in R["after3"]["d"].items():
        check_summary(d.state_dict()[k], ref, 2e-4, k)
    assert R["encoder112_raises"]   # the reference pyramid does not exist at 112 -> generalisation is ours


def test_ganomaly_baseline_config1():
    """BASELINE.json configs[0]: ganomaly on 8x64x64 clips, batch 2 (16 frames), CPU."""
    R = JS["ganomaly_cfg1"]
    opt = OG.make_opt(isize=64)
    g, d = fill_module(OG.NetG(opt), R["seeds"]["g"]).train(), fill_module(OG.NetD(opt), R["seeds"]["d"]).train()
    assert sum(p.numel() for p in g.parameters()) == R["n_params_g"] and sum(p.numel() for p in d.parameters()) == R["n_params_d"]
    og, od = OG.make_optimizers(g, d, opt)
    errs, fk = OG.step(g, d, og, od, seeded_tensor((16, 3, 64, 64), R["seeds"]["x"]), opt)
    check_errs(errs, R["errs"], 1e-5)
    check_summary(fk, R["fake"], 1e-4, "fake")


def _set_p0(m):
    for mm in m.modules():
        if isinstance(mm, nn.Dropout):
            mm.p = 0.0


class masked_dropout:
    """Same deterministic keep-masks as the fixture generator: U(seed = 1000 + call index) >= p."""

    def __enter__(self):
        self.orig = torch.nn.functional.dropout
        self.calls = 0

        def fn(input, p=0.5, training=True, inplace=False):
            if not training or p == 0:
                return input
            mask = (seeded_tensor(tuple(input.shape), 1000 + self.calls, 0.0, 1.0) >= p).float()
            self.calls += 1
            return input * mask / (1.0 - p)
        torch.nn.functional.dropout = fn
        return self

    def __exit__(self, *a):
        torch.nn.functional.dropout = self.orig


def test_anogan_step():
    R = JS["anogan"]
    g, d = fill_module(OA.NetG(), R["seeds"]["g"]).train(), fill_module(OA.NetD(), R["seeds"]["d"]).train()
    assert list(g.state_dict().keys()) == R["keys_g"] and list(d.state_dict().keys()) == R["keys_d"]
    assert sum(p.numel() for p in g.parameters()) == R["n_params_g"] == 33975353
    assert sum(p.numel() for p in d.parameters()) == R["n_params_d"] == 1849473
    z, real = seeded_normal((2, 100), R["seeds"]["z"]), seeded_tensor((2, 3, 16, 128, 128), R["seeds"]["real"])
    with masked_dropout() as md:
        out = fill_module(OA.NetG(), R["seeds"]["g"]).train()(z)
        assert md.calls == R["fwd_masked"]["mask_calls"] == 4
    check_summary(out, R["fwd_masked"]["fake"], 1e-4, "masked fake")
    _set_p0(g)
    g_opt, d_opt = OA.make_optimizers(g, d, 2e-5)
    errs, fake = OA.step(g, d, g_opt, d_opt, real, z)
    check_errs(errs, R["step_p0"]["errs"], 1e-4)
    check_summary(fake, R["step_p0"]["fake"], 1e-4, "fake")
    for k, ref in R["after1"]["d"].items():
        check_summary(d.state_dict()[k], ref, 5e-4, k)
    for k, ref in R["after1"]["g_bn"].items():
        check_summary(g.state_dict()[k], ref, 5e-4, k)
    assert R["netd112_raises"]
    # generalised geometry (BASELINE config 3): constructible and shape-consistent at 16x112x112
    g112, d112 = OA.NetG(16, 112), OA.NetD(16, 112)
    assert g112.layer1[0].out_features == 512 * 2 * 14 * 14 and d112.fc[0].in_features == 256 * 2 * 14 * 14


def test_mygan_step():
    R = JS["mygan"]
    args = OM.make_args(16, 128)
    g, d = fill_module(OM.NetG(), R["seeds"]["g"]).train(), fill_module(OM.NetD(args), R["seeds"]["d"]).train()
    assert list(g.state_dict().keys()) == R["keys_g"] and list(d.state_dict().keys()) == R["keys_d"]
    assert sum(p.numel() for p in g.parameters()) == R["n_params_g"] == 13527885
    assert sum(p.numel() for p in d.parameters()) == R["n_params_d"] == 6324353
    s = R["seeds"]
    inp = seeded_tensor((2, 3, 16, 128, 128), s["inp"])
    gt = (seeded_tensor((2, 1, 16, 128, 128), s["gt"], 0.0, 1.0) > 0.97).float()
    gt_flow, pre_flow = seeded_tensor((2, 3, 16, 128, 128), s["gt_flow"]), seeded_tensor((2, 3, 16, 128, 128), s["pre_flow"])
    with masked_dropout() as md:
        out = fill_module(OM.NetG(), s["g"]).train()(inp[:1])
        assert md.calls == R["fwd_masked"]["mask_calls"] == 4
    check_summary(out, R["fwd_masked"]["predict"], 1e-4, "masked predict")
    g3 = fill_module(OM.NetG(), s["g"]).train()
    _set_p0(g3)
    check_summary(g3(seeded_tensor((1, 3, 16, 112, 112), 67)), R["fwd_112"]["predict"], 1e-4, "predict@112")
    _set_p0(g)
    og, od = OM.make_optimizers(g, d)
    errs, predict = OM.step(g, d, og, od, inp, gt, gt_flow, pre_flow)
    check_errs(errs, R["step_p0"]["errs"], 1e-4)
    check_summary(predict, R["step_p0"]["predict"], 1e-4, "predict")
    for k, ref in R["after1"]["g_head"].items():
        check_summary(g.state_dict()[k], ref, 5e-4, k)
    for k, ref in R["after1"]["d_head"].items():
        check_summary(d.state_dict()[k], ref, 5e-4, k)
    assert R["netd112_raises"]
    d112 = OM.NetD(OM.make_args(16, 112))
    assert d112.spatdisc.linear.in_features == 1024 * 1 * 1


def test_mygan_netg_224():
    """BASELINE configs[3] geometry (16x224x224): the oracle's NetG against the reference's own NetG output (fixture
    mygan224, generated by tests/golden/make_fixtures.py mygan224)."""
    R = JS["mygan224"]
    g = fill_module(OM.NetG(), R["seeds"]["g"]).train()
    _set_p0(g)
    with torch.no_grad():
        out = g(seeded_tensor((1, 3, 16, 224, 224), R["seeds"]["inp"]))
    check_summary(out, R["predict"], 1e-4, "predict@224")


@pytest.mark.parametrize("name", ["autoencoder", "xception"])
def test_supervised_baselines_step(name):
    """the oracle's restatement of models/mystcnn.py AutoEncoder / models/xception.py Xception and of the
    lib/train_stcnn.py:103-108 step against the vectors of the reference's own classes (16x128x128, B=1)."""
    from vfd_oracle import mystcnn as OMS, xception as OX
    R = JS["baselines"][name]
    m = fill_module({"autoencoder": OMS.AutoEncoder, "xception": OX.Xception}[name](), R["seeds"]["net"]).train()
    assert list(m.state_dict().keys()) == R["keys"] and sum(p.numel() for p in m.parameters()) == R["n_params"]
    for mm in m.modules():
        if isinstance(mm, nn.Dropout):
            mm.p = 0.0
    inp = seeded_tensor((1, 3, 16, 128, 128), R["seeds"]["inp"])
    gt = (seeded_tensor((1, 1, 16, 128, 128), R["seeds"]["gt"], 0.0, 1.0) > 0.97).float()
    errs, predict = OMS.step(m, OMS.make_optimizer(m), inp, gt)
    check_errs(errs, R["step_p0"]["errs"], RT, name)
    check_summary(predict, R["step_p0"]["predict"], 1e-4, name + " predict")
    sd = m.state_dict()
    for k, ref in R["after1"].items():
        check_summary(sd[k].float(), ref, 1e-4, k)
