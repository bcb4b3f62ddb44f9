"""bf16 HIP step vs the float32 oracle AND vs the bf16-faithful oracle (oracle/vfd_oracle/bf16.py), all three models: the
numbers the tightened gates of the GPU suite are set from.  Usage: python tools/probe/bf16_parity.py [ganomaly anogan mygan]"""
import os
import sys
import tempfile
import types

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402

from util import relerr, relrms  # noqa: E402
from vfd_gan_amd import functional as F  # noqa: E402
from vfd_gan_amd.lib.data import synthetic_batch  # noqa: E402
from vfd_oracle import bf16 as OB  # noqa: E402
from vfd_oracle.weights import fill_module, seeded_normal, seeded_tensor  # noqa: E402

torch.set_num_threads(min(16, os.cpu_count() or 1))
dev = torch.device("cuda", 0)


def p0(m):
    for mm in m.modules():
        if isinstance(mm, torch.nn.Dropout):
            mm.p = 0.0


def report(tag, got, ref, out, out_ref, grads):
    worst = max(abs(got[k] - v) / max(abs(v), 1e-3) for k, v in ref.items())
    print("  [%s] worst loss rel err %.3e   output relrms %.3e" % (tag, worst, relrms(out, out_ref)))
    for k, v in ref.items():
        print("      %-14s got %.6f ref %.6f  rel %.2e" % (k, got[k], v, abs(got[k] - v) / max(abs(v), 1e-3)))
    rm = sorted(((relrms(g, r), relerr(g, r), k) for k, g, r in grads if float(r.abs().max()) > 1e-7), reverse=True)
    print("      grads: worst relrms %.3e (%s), median %.3e; worst max-norm %.3e" % (rm[0][0], rm[0][2], rm[len(rm) // 2][0], max(x[1] for x in rm)))
    for x in rm[:6]:
        print("        %-50s rms %.3e max %.3e" % (x[2], x[0], x[1]))


def args_ns(model, B, T, S, **kw):
    d = dict(batchsize=B, nfr=T, isize=S, ich=3, lr=2e-5, beta1=0.5, w_adv=1, w_con=10, pos_weight=2, freq=10 ** 9, ep=1,
             model=model, result_root=tempfile.mkdtemp(), gpu=[0], ae=False, steps_per_epoch=1)
    d.update(kw)
    return types.SimpleNamespace(**d)


def run_ganomaly(T=16, S=112, ngf=64):
    from vfd_gan_amd.models import ganomaly as HG
    from vfd_oracle import ganomaly as OG
    print("ganomaly %dx%dx%d ngf=%d" % (T, S, S, ngf))
    opt = OG.make_opt(isize=S, ngf=ngf)
    batch = synthetic_batch(1, T, S, 3, seed=321)
    x = OG.fold_frames(batch[0])
    sd = None
    res = {}
    for tag in ("f32", "bf16"):
        og, od = fill_module(OG.NetG(opt), 7), fill_module(OG.NetD(opt), 8)
        sd = ({k: v.clone() for k, v in og.state_dict().items()}, {k: v.clone() for k, v in od.state_dict().items()})
        if tag == "f32":
            ref, fake = OG.step(og, od, *OG.make_optimizers(og, od, opt), x, opt)
        else:
            ref, fake = OG.step(OB.Faithful(og), OB.Faithful(od), *OG.make_optimizers(og, od, opt), OB.rbf(x), opt)
        res[tag] = (ref, fake, [(k, p.grad.clone()) for k, p in list(og.named_parameters()) + list(od.named_parameters())])
    F.set_compute_dtype(torch.bfloat16)
    m = HG.Ganomaly(args_ns("ganomaly", 1, T, S, lr=2e-4, w_con=50), None, opt=HG.make_opt(isize=S, ngf=ngf))
    m.netg.load_state_dict(sd[0])
    m.netd.load_state_dict(sd[1])
    F.invalidate_weight_cache()
    m.set_input(batch)
    m.optimize_params(check_collapse=False)
    got = {k.split("/")[1]: v for k, v in m.errors().items()}
    hg = [p.grad for _, p in list(m.netg.named_parameters()) + list(m.netd.named_parameters())]
    for tag in ("f32", "bf16"):
        ref, fake, gr = res[tag]
        report("vs %s oracle" % tag, got, ref, m.fake.to_torch(), fake, [(k, g, r) for (k, r), g in zip(gr, hg)])


def run_anogan(B=2, T=16, S=112):
    from vfd_gan_amd.models import anogan as HA
    from vfd_oracle import anogan as OA
    print("anogan %dx%dx%d B=%d" % (T, S, S, B))
    z, real = seeded_normal((B, 100), 73), seeded_tensor((B, 3, T, S, S), 74)
    res = {}
    for tag in ("f32", "bf16"):
        og, od = fill_module(OA.NetG(T, S), 71).train(), fill_module(OA.NetD(T, S), 72).train()
        p0(og)
        sd = ({k: v.clone() for k, v in og.state_dict().items()}, {k: v.clone() for k, v in od.state_dict().items()})
        g_opt, d_opt = OA.make_optimizers(og, od, 2e-5)
        if tag == "f32":
            ref, fake = OA.step(og, od, g_opt, d_opt, real, z)
        else:
            ref, fake = OA.step(OB.Faithful(og), OB.Faithful(od), g_opt, d_opt, OB.rbf(real), OB.rbf(z))
        res[tag] = (ref, fake, [(k, p.grad.clone()) for k, p in og.named_parameters()])      # netG's (netD's are re-zeroed mid-step)
    F.set_compute_dtype(torch.bfloat16)
    m = HA.AnoGAN(args_ns("anogan", B, T, S), None)
    m.netg.load_state_dict(sd[0])
    m.netd.load_state_dict(sd[1])
    p0(m.netg)
    F.invalidate_weight_cache()
    m.set_input((real, real, real[:, :1], torch.ones(B, T)))
    m.z = z.to(dev)
    m.optimize_params()
    got = {k.split("/")[1]: v for k, v in m.errors().items()}
    hg = [p.grad for _, p in m.netg.named_parameters()]
    for tag in ("f32", "bf16"):
        ref, fake, gr = res[tag]
        report("vs %s oracle" % tag, got, ref, m.gen_fake.to_torch(), fake, [(k, g, r) for (k, r), g in zip(gr, hg)])


def run_mygan(B=1, T=16, S=64):
    from vfd_gan_amd.models import mygannet as HM
    from vfd_oracle import mygannet as OM
    print("mygan %dx%dx%d B=%d" % (T, S, S, B))
    inp = seeded_tensor((B, 3, T, S, S), 30)
    gt = (seeded_tensor((B, 1, T, S, S), 40, 0.0, 1.0) > 0.97).float()
    gf, pf = seeded_tensor((B, 3, T, S, S), 50), seeded_tensor((B, 3, T, S, S), 60)
    res = {}
    for tag in ("f32", "bf16"):
        og, od = fill_module(OM.NetG(), 3).train(), fill_module(OM.NetD(OM.make_args(T, S)), 4).train()
        p0(og)
        sd = ({k: v.clone() for k, v in og.state_dict().items()}, {k: v.clone() for k, v in od.state_dict().items()})
        opt_g, opt_d = OM.make_optimizers(og, od)
        if tag == "f32":
            ref, pred = OM.step(og, od, opt_g, opt_d, inp, gt, gf, pf)
        else:
            ref, pred = OM.step(OB.Faithful(og), OB.Faithful(od), opt_g, opt_d, OB.rbf(inp), gt, OB.rbf(gf), OB.rbf(pf))
        res[tag] = (ref, pred, [(k, p.grad.clone()) for k, p in list(og.named_parameters()) + list(od.named_parameters())])
    F.set_compute_dtype(torch.bfloat16)
    m = HM.MyGAN(args_ns("mygan", B, T, S), None)
    m.netg.load_state_dict(sd[0])
    m.netd.load_state_dict(sd[1])
    p0(m.netg)
    F.invalidate_weight_cache()
    m.set_input((inp, inp, gt, torch.ones(B, T)), gt_flow=gf, pre_flow=pf)
    m.optimize_params()
    got = {k.split("/")[1]: v for k, v in m.errors().items()}
    hg = [p.grad for _, p in list(m.netg.named_parameters()) + list(m.netd.named_parameters())]
    for tag in ("f32", "bf16"):
        ref, pred, gr = res[tag]
        report("vs %s oracle" % tag, got, ref, m.predict.to_torch(), pred, [(k, g, r) for (k, r), g in zip(gr, hg)])


if __name__ == "__main__":
    which = sys.argv[1:] or ["ganomaly", "anogan", "mygan"]
    for w in which:
        {"ganomaly": run_ganomaly, "anogan": run_anogan, "mygan": run_mygan}[w]()
