#!/usr/bin/env python3
"""Generate tests/golden/*.json|npz by running the REFERENCE's own nn.Module classes and loss functions.

    cd /tmp && python /root/repo/tests/golden/make_fixtures.py        # needs /root/reference (not on the GPU box)

The reference's hot-path modules import cv2 / torchvision / tensorboard / skimage (absent in this image) and three
lib.* modules that do not exist in its tree; none of them is touched by any nn.Module.__init__/forward or by
l2_loss / weighted_bce / weights_init, so inert stubs are registered in sys.modules first.
The trainer classes hard-code 'cuda' and call cv2 inside the step, so the step SEQUENCE comes from
oracle/vfd_oracle/*.step — driven here with the reference's module classes and the reference's loss functions.
Weights and inputs come from numpy PCG64 (vfd_oracle.weights), never from torch's init RNG.
Only data (inputs' seeds, outputs, loss scalars) is written; no reference source is copied.
"""
import json
import os
import sys
import types

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.dont_write_bytecode = True


def _stub(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


_stub("cv2")
_tv = _stub("torchvision")
_tv.utils = _stub("torchvision.utils", save_image=lambda *a, **k: None, make_grid=lambda *a, **k: None)
_tv.transforms = _stub("torchvision.transforms")
_sk = _stub("skimage")
_sk.transform = _stub("skimage.transform", resize=lambda *a, **k: None)
_stub("tensorboard")
import torch  # noqa: E402
import torch.utils  # noqa: E402
torch.utils.tensorboard = _stub("torch.utils.tensorboard", SummaryWriter=object)
_stub("lib.networks", NetG=None, NetD=None, weights_init=None)
_stub("lib.visualizer", Visualizer=None)
_stub("lib.loss", l2_loss=None)

sys.path.insert(0, os.path.join(REPO, "oracle"))     # vfd_oracle.weights / step sequences (package name is unique)
sys.path.insert(0, "/root/reference")
import numpy as np  # noqa: E402
import torch.nn as nn  # noqa: E402
import lib.utils as RU  # noqa: E402  (reference)
import models.anogan as RA  # noqa: E402
import models.ganomaly as RG  # noqa: E402
import models.mygannet as RM  # noqa: E402
import models.spatiotempconv as RS  # noqa: E402
from vfd_oracle import anogan as OA, ganomaly as OG, mygannet as OM  # noqa: E402
from vfd_oracle.weights import fill_module, seeded_normal, seeded_tensor, summarize  # noqa: E402

torch.set_num_threads(8)
OUT_JSON = {}
OUT_NPZ = {}


def sd_summary(module):
    return {k: summarize(v.float()) for k, v in module.state_dict().items() if v.dtype.is_floating_point}


# deterministic dropout: keep-mask = U(seed = 1000 + call index) >= p  (same provider in the tests)
_DROP = {"calls": 0, "on": False}
_orig_dropout = torch.nn.functional.dropout


def _dropout(input, p=0.5, training=True, inplace=False):
    if not _DROP["on"] or not training or p == 0:
        return _orig_dropout(input, p, training, inplace)
    mask = (seeded_tensor(tuple(input.shape), 1000 + _DROP["calls"], 0.0, 1.0) >= p).float()
    _DROP["calls"] += 1
    return input * mask / (1.0 - p)


torch.nn.functional.dropout = _dropout


def set_dropout_p(module, p):
    for m in module.modules():
        if isinstance(m, nn.Dropout):
            m.p = p


# ---------------------------------------------------------------------------------------------------------------
def kats():
    x, t = torch.tensor([.2, .9, .5]), torch.tensor([0., 1., 1.])
    OUT_JSON["kat_losses"] = {"x": x.tolist(), "t": t.tolist(), "l2_loss": RU.l2_loss(x, t).item(),
                              "weighted_bce": RU.weighted_bce(x, t).item(), "bce": nn.BCELoss()(x, t).item(),
                              "weighted_bce_pw_none": RU.weighted_bce(x, t, pos_weight=None).item(),
                              "gray2rgb_shape": list(RU.gray2rgb(torch.zeros(2, 1, 3, 4, 4)).shape)}
    # weights_init touches only Conv3d / BatchNorm3d instances (lib/utils.py:51-56)
    touched = {}
    for name, m in [("Conv3d", nn.Conv3d(2, 2, 3)), ("ConvTranspose3d", nn.ConvTranspose3d(2, 2, 3)), ("Linear", nn.Linear(4, 4)),
                    ("BatchNorm3d", nn.BatchNorm3d(4)), ("BatchNorm1d", nn.BatchNorm1d(4)), ("Conv2d", nn.Conv2d(2, 2, 3)),
                    ("ConvTranspose2d", nn.ConvTranspose2d(2, 2, 3)), ("BatchNorm2d", nn.BatchNorm2d(4))]:
        before = [p.clone() for p in m.parameters()]
        if isinstance(m, (nn.BatchNorm1d, nn.BatchNorm2d, nn.BatchNorm3d)):
            m.bias.data.fill_(0.5)
            before = [p.clone() for p in m.parameters()]
        torch.manual_seed(0)
        RU.weights_init(m)
        touched[name] = [bool((a != b).any()) for a, b in zip(before, m.parameters())]
    OUT_JSON["weights_init_touched"] = touched
    OUT_JSON["spatiotemp_intermed"] = {"%d,%d,%s" % (i, o, k): RS.SpatioTemporalConv(i, o, k).spatial_conv.out_channels
                                        for i, o, k in [(3, 32, 3), (32, 64, 3), (64, 128, 3), (128, 256, 3), (256, 512, 3),
                                                        (512, 256, 3), (96, 32, 3), (3, 32, (1, 3, 3)), (3, 32, (3, 1, 1)),
                                                        (32, 64, (3, 1, 1)), (512, 1024, (1, 3, 3))]}


def args_defaults():
    import lib.args as RAr  # reference flag system; parse() itself needs a CUDA device (lib/args.py:52), the parser does not
    ns = RAr.Args().parser.parse_args([])
    OUT_JSON["args_defaults"] = {k: v for k, v in vars(ns).items() if k not in ("tr_plist", "ts_plist", "result_root")}


def spatiotemp():
    m = fill_module(RS.SpatioTemporalConv(3, 8, 3, padding=1), 11).train()
    x = seeded_tensor((2, 3, 4, 8, 8), 12).requires_grad_()
    y = m(x)
    y.pow(2).mean().backward()
    OUT_JSON["spatiotemp"] = {"keys": list(m.state_dict().keys()), "in": [3, 8, 3], "seed_w": 11, "seed_x": 12}
    OUT_NPZ["spatiotemp_y"] = y.detach().numpy()
    OUT_NPZ["spatiotemp_gx"] = x.grad.numpy()
    OUT_NPZ["spatiotemp_gw_spatial"] = m.spatial_conv.weight.grad.numpy()
    OUT_NPZ["spatiotemp_running_var"] = m.bn.running_var.numpy()


def ganomaly():
    cfg = dict(isize=32, ngf=16, nz=100, nc=3, ngpu=1, extralayers=0)
    opt = OG.make_opt(**cfg)
    g, d = fill_module(RG.NetG(opt), 21).train(), fill_module(RG.NetD(opt), 22).train()
    x = seeded_tensor((8, 3, 32, 32), 23)
    fake, li, lo = g(x)
    pred, feat = d(x)
    rec = {"cfg": cfg, "seeds": {"g": 21, "d": 22, "x": 23}, "keys_g": list(g.state_dict().keys()),
           "keys_d": list(d.state_dict().keys()), "n_params_g": sum(p.numel() for p in g.parameters()),
           "n_params_d": sum(p.numel() for p in d.parameters()),
           "fwd": {"fake": summarize(fake), "latent_i": summarize(li), "latent_o": summarize(lo), "pred": summarize(pred),
                   "feat": summarize(feat)}}
    OUT_NPZ["ganomaly_latent_i"] = li.detach().numpy()
    OUT_NPZ["ganomaly_pred"] = pred.detach().numpy()
    # three restated steps with the reference's nets and the reference's l2_loss
    g, d = fill_module(RG.NetG(opt), 21).train(), fill_module(RG.NetD(opt), 22).train()
    og, od = OG.make_optimizers(g, d, opt)
    steps = []
    for it in range(3):
        errs, fk = OG.step(g, d, og, od, seeded_tensor((8, 3, 32, 32), 30 + it), opt, l2=RU.l2_loss)
        steps.append({"errs": errs, "fake": summarize(fk)})
    rec["steps"] = steps
    rec["after3"] = {"g": sd_summary(g), "d": sd_summary(d)}
    # the reference pyramid only exists for power-of-two frame sizes
    try:
        RG.Encoder(112, 100, 3, 64, 1)(torch.zeros(1, 3, 112, 112))
        rec["encoder112_raises"] = False
    except Exception as e:  # noqa: BLE001
        rec["encoder112_raises"] = type(e).__name__
    rec["decoder112"] = "not constructed: `while tisize != isize` never terminates (models/ganomaly.py:88-91)"
    OUT_JSON["ganomaly"] = rec
    # config 1 of BASELINE.json: 8x64x64 clips, batch 2 -> 16 frames, upstream defaults ngf=64
    opt64 = OG.make_opt(isize=64)
    g, d = fill_module(RG.NetG(opt64), 41).train(), fill_module(RG.NetD(opt64), 42).train()
    og, od = OG.make_optimizers(g, d, opt64)
    errs, fk = OG.step(g, d, og, od, seeded_tensor((16, 3, 64, 64), 43), opt64, l2=RU.l2_loss)
    OUT_JSON["ganomaly_cfg1"] = {"cfg": dict(isize=64), "seeds": {"g": 41, "d": 42, "x": 43}, "errs": errs, "fake": summarize(fk),
                                 "n_params_g": sum(p.numel() for p in g.parameters()),
                                 "n_params_d": sum(p.numel() for p in d.parameters())}


def anogan():
    g, d = fill_module(RA.NetG(), 51).train(), fill_module(RA.NetD(), 52).train()
    set_dropout_p(g, 0.0)
    rec = {"seeds": {"g": 51, "d": 52, "z": 53, "real": 54}, "keys_g": list(g.state_dict().keys()),
           "keys_d": list(d.state_dict().keys()), "n_params_g": sum(p.numel() for p in g.parameters()),
           "n_params_d": sum(p.numel() for p in d.parameters())}
    z, real = seeded_normal((2, 100), 53), seeded_tensor((2, 3, 16, 128, 128), 54)
    g_opt, d_opt = OA.make_optimizers(g, d, 2e-5)
    errs, fake = OA.step(g, d, g_opt, d_opt, real, z)
    rec["step_p0"] = {"errs": errs, "fake": summarize(fake)}
    rec["after1"] = {"d": sd_summary(d), "g_bn": {k: v for k, v in sd_summary(g).items() if "running" in k or k.startswith("layer3")}}
    # masked dropout forward
    g2 = fill_module(RA.NetG(), 51).train()
    _DROP.update(on=True, calls=0)
    out = g2(z)
    _DROP.update(on=False)
    rec["fwd_masked"] = {"fake": summarize(out), "mask_calls": _DROP["calls"]}
    try:
        RA.NetD()(torch.zeros(2, 3, 16, 112, 112))
        rec["netd112_raises"] = False
    except Exception as e:  # noqa: BLE001
        rec["netd112_raises"] = type(e).__name__
    OUT_JSON["anogan"] = rec


def mygan():
    args = types.SimpleNamespace(nfr=16, isize=128)
    g, d = fill_module(RM.NetG(), 61).train(), fill_module(RM.NetD(args), 62).train()
    set_dropout_p(g, 0.0)
    rec = {"seeds": {"g": 61, "d": 62, "inp": 63, "gt": 64, "gt_flow": 65, "pre_flow": 66},
           "keys_g": list(g.state_dict().keys()), "keys_d": list(d.state_dict().keys()),
           "n_params_g": sum(p.numel() for p in g.parameters()), "n_params_d": sum(p.numel() for p in d.parameters())}
    B = 2
    inp = seeded_tensor((B, 3, 16, 128, 128), 63)
    gt = (seeded_tensor((B, 1, 16, 128, 128), 64, 0.0, 1.0) > 0.97).float()
    gt_flow, pre_flow = seeded_tensor((B, 3, 16, 128, 128), 65), seeded_tensor((B, 3, 16, 128, 128), 66)
    og, od = OM.make_optimizers(g, d)
    errs, predict = OM.step(g, d, og, od, inp, gt, gt_flow, pre_flow, fns=RU)
    rec["step_p0"] = {"errs": errs, "predict": summarize(predict)}
    rec["after1"] = {"g_head": {k: v for k, v in sd_summary(g).items() if k.startswith(("dconv1", "conv_last", "uconv1.bn"))},
                     "d_head": {k: v for k, v in sd_summary(d).items() if k.startswith(("spatdisc.dconv1", "tempdisc.dconv3", "spatdisc.linear"))}}
    # forward with imposed dropout masks, B=1, plus a size the U-Net accepts but the reference NetD does not
    g2 = fill_module(RM.NetG(), 61).train()
    _DROP.update(on=True, calls=0)
    out = g2(inp[:1])
    _DROP.update(on=False)
    rec["fwd_masked"] = {"predict": summarize(out), "mask_calls": _DROP["calls"]}
    g3 = fill_module(RM.NetG(), 61).train()
    set_dropout_p(g3, 0.0)
    rec["fwd_112"] = {"predict": summarize(g3(seeded_tensor((1, 3, 16, 112, 112), 67)))}
    try:
        RM.NetD(types.SimpleNamespace(nfr=16, isize=112))(torch.zeros(2, 3, 16, 112, 112), torch.zeros(2, 3, 16, 112, 112))
        rec["netd112_raises"] = False
    except Exception as e:  # noqa: BLE001
        rec["netd112_raises"] = type(e).__name__
    OUT_JSON["mygan"] = rec


def mygan224():
    """BASELINE configs[3] geometry: the reference's size-agnostic U-Net NetG at 16x224x224 (B=1, dropout p=0, training-mode
    BatchNorm).  Its NetD is locked to 16x128x128 (models/mygannet.py:134,176), so only the generator is pinned here."""
    g = fill_module(RM.NetG(), 61).train()
    set_dropout_p(g, 0.0)
    with torch.no_grad():
        out = g(seeded_tensor((1, 3, 16, 224, 224), 68))
    OUT_JSON["mygan224"] = {"seeds": {"g": 61, "inp": 68}, "predict": summarize(out)}


def baselines():
    """the supervised baselines' nets (reference models/mystcnn.py AutoEncoder, models/xception.py Xception) at the
    reference's 16x128x128, B=1, Dropout p=0, training-mode BatchNorm: forward, and ONE step of lib/train_stcnn.py:103-108
    (BCELoss against the mask, Adam(2e-5, (0.5, 0.999))) restated by vfd_oracle.mystcnn.step around the reference's modules."""
    import models.mystcnn as RMS
    import models.xception as RX
    from vfd_oracle import mystcnn as OMS
    rec = {}
    inp = seeded_tensor((1, 3, 16, 128, 128), 73)
    gt = (seeded_tensor((1, 1, 16, 128, 128), 74, 0.0, 1.0) > 0.97).float()
    for name, cls, seed in (("autoencoder", RMS.AutoEncoder, 71), ("xception", RX.Xception, 72)):
        m = fill_module(cls(), seed).train()
        set_dropout_p(m, 0.0)
        opt = OMS.make_optimizer(m)
        errs, predict = OMS.step(m, opt, inp, gt)
        sd = sd_summary(m)
        keys = list(m.state_dict().keys())
        pick = [k for k in keys if k.endswith(("running_mean", "running_var"))][:4] + [k for k in keys if k.endswith("weight")][:3] + \
               [k for k in keys if k.endswith("weight")][-3:]
        rec[name] = {"seeds": {"net": seed, "inp": 73, "gt": 74}, "keys": keys, "n_params": sum(p.numel() for p in m.parameters()),
                     "step_p0": {"errs": errs, "predict": summarize(predict)}, "after1": {k: sd[k] for k in pick}}
    OUT_JSON["baselines"] = rec


if __name__ == "__main__":
    which = sys.argv[1:] or ["kats", "args_defaults", "spatiotemp", "ganomaly", "anogan", "mygan"]
    jp, npz = os.path.join(HERE, "reference_vectors.json"), os.path.join(HERE, "reference_vectors.npz")
    if os.path.exists(jp):
        OUT_JSON.update(json.load(open(jp)))
    if os.path.exists(npz):
        OUT_NPZ.update(dict(np.load(npz)))
    for w in which:
        print("generating", w, flush=True)
        globals()[w]()
    OUT_JSON["_meta"] = {"torch": torch.__version__, "generator": "tests/golden/make_fixtures.py",
                         "note": "outputs of the reference's own nn.Module classes / loss functions on torch CPU float32"}
    with open(jp, "w") as f:
        json.dump(OUT_JSON, f, indent=1, sort_keys=True)
    np.savez_compressed(npz, **OUT_NPZ)
    print("wrote", jp, npz)
