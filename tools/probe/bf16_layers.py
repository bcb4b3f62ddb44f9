"""Where does the bf16 HIP forward leave the bf16-faithful oracle (oracle/vfd_oracle/bf16.py)?  Stage-by-stage comparison of
stored tensors (fraction of elements that differ at all, relative RMS) for ganomaly's nets (prefixes of the Sequentials, cut
after each fused group) and mygan's NetG (block by block).  A plan mismatch shows as a jump at one stage; float32 summation
order alone shows as a sprinkle of single-ulp differences."""
import os
import sys
import types
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402
import torch.nn as nn  # noqa: E402

from util import relrms  # noqa: E402
from vfd_gan_amd import functional as F, nn as hnn  # noqa: E402
from vfd_oracle import bf16 as OB  # noqa: E402
from vfd_oracle.weights import fill_module, seeded_tensor  # noqa: E402

dev = torch.device("cuda", 0)
F.set_compute_dtype(torch.bfloat16)


def cmp(tag, h, o):
    h = h.to_torch().cpu() if isinstance(h, F.ClTensor) else h.cpu()
    d = (h != o)
    print("  %-44s differ %7.3f %%   relrms %.3e   max|o| %.3g" % (tag, 100.0 * d.float().mean().item(), relrms(h, o), o.abs().max().item()))


def groups(mods):
    """cut points of a Sequential after each fused group (conv[+act] | conv,bn[,act][,pool] | other)"""
    cuts, i, n = [], 0, len(mods)
    while i < n:
        m = mods[i]
        if isinstance(m, OB._CONVS):
            j = i + 1
            if j < n and isinstance(mods[j], OB._ACTS):
                j += 1
            elif j < n and isinstance(mods[j], OB._BNS):
                j += 1
                if j < n and isinstance(mods[j], OB._ACTS):
                    j += 1
                if j < n and isinstance(mods[j], nn.AvgPool3d):
                    j += 1
            i = j
        elif isinstance(m, OB._BNS):
            j = i + 1
            if j < n and isinstance(mods[j], OB._ACTS):
                j += 1
            i = j
        else:
            i += 1
        cuts.append(i)
    return cuts


def seq_prefixes(name, hseq, oseq, x):
    hm, om = list(hseq), list(oseq)
    xq = OB.rbf(x)
    for c in groups(om):
        for m in om:
            m.train()
        with torch.no_grad():
            o = OB.run_seq(om[:c], xq)
            h = hnn.run_fused(hm[:c], F.to_cl(x.to(dev)))
        cmp("%s[:%d] %s" % (name, c, type(om[c - 1]).__name__), h, o)


def ganomaly():
    from vfd_gan_amd.models import ganomaly as HG
    from vfd_oracle import ganomaly as OG
    S, ngf, N = 112, 64, 16
    opt = OG.make_opt(isize=S, ngf=ngf)
    og, od = fill_module(OG.NetG(opt), 7), fill_module(OG.NetD(opt), 8)
    args = types.SimpleNamespace(batchsize=1, nfr=N, isize=S, ich=3, lr=2e-4, beta1=0.5, w_adv=1, w_con=50, freq=10 ** 9, ep=1,
                                 model="ganomaly", result_root=tempfile.mkdtemp(), gpu=[0], steps_per_epoch=1)
    m = HG.Ganomaly(args, None, opt=HG.make_opt(isize=S, ngf=ngf))
    m.netg.load_state_dict(og.state_dict())
    m.netd.load_state_dict(od.state_dict())
    F.invalidate_weight_cache()
    x = seeded_tensor((N, 3, S, S), 5)
    print("ganomaly netD.features")
    seq_prefixes("features", m.netd.features, od.features, x)
    print("ganomaly encoder1 -> decoder")
    seq_prefixes("enc1", m.netg.encoder1.main, og.encoder1.main, x)
    with torch.no_grad():
        z = OB.run_seq(og.encoder1.main, OB.rbf(x))
    seq_prefixes("dec", m.netg.decoder.main, og.decoder.main, z)


def mygan():
    from vfd_gan_amd.models import mygannet as HM
    from vfd_oracle import mygannet as OM
    T, S = 16, 64
    og = fill_module(OM.NetG(), 3).train()
    hg = HM.NetG(3).to(dev).train()
    hg.load_state_dict(og.state_dict())
    for net in (og, hg):
        for mm in net.modules():
            if isinstance(mm, nn.Dropout):
                mm.p = 0.0
    F.invalidate_weight_cache()
    x = seeded_tensor((1, 3, T, S, S), 30)
    xq = OB.rbf(x)
    print("mygan NetG")
    with torch.no_grad():
        xc = F.to_cl(x.to(dev))
        # stconv of block 1, then block 1
        cmp("dconv1.conv (SpatioTemporalConv)", hg.dconv1.conv(xc), OB.stconv(og.dconv1.conv, xq))
        hp, hf = hg.dconv1(xc, pool=hg.avgpool)
        op, of = OB.conv_bn_act(og.dconv1, xq, og.avgpool, True)
        cmp("dconv1 full", hf, of)
        cmp("dconv1 pooled", hp, op)
        hs, os_ = [hf], [of]
        for k in (2, 3, 4):
            hp, hf = getattr(hg, "dconv%d" % k)(hp, pool=hg.avgpool)
            op, of = OB.conv_bn_act(getattr(og, "dconv%d" % k), op, og.avgpool, True)
            cmp("dconv%d full" % k, hf, of)
            cmp("dconv%d pooled" % k, hp, op)
            hs.append(hf)
            os_.append(of)
        hl, ol = hg.dconv5(hp), OB.conv_bn_act(og.dconv5, op)
        cmp("dconv5 (latent)", hl, ol)
        hx, ox = hg.uconv5(hl), OB.conv_bn_act(og.uconv5, ol)
        cmp("uconv5", hx, ox)
        for k in (4, 3, 2, 1):
            hc = F.upsample_cat(hx, hs[k - 1])
            oc = OB.upsample_cat(og.upsamp, ox, os_[k - 1])
            cmp("upsample_cat -> uconv%d input" % k, hc, oc)
            hx, ox = getattr(hg, "uconv%d" % k)(hc), OB.conv_bn_act(getattr(og, "uconv%d" % k), oc)
            cmp("uconv%d" % k, hx, ox)
        cmp("conv_last + sigmoid", hg.conv_last(hx, act=2), OB.run_seq([og.conv_last, og.sigmoid], ox))


def anogan():
    from vfd_gan_amd.models import anogan as HA
    from vfd_oracle import anogan as OA
    T, S = 16, 112
    od = fill_module(OA.NetD(T, S), 72).train()
    hd = HA.NetD(T, S).to(dev).train()
    hd.load_state_dict(od.state_dict())
    F.invalidate_weight_cache()
    x = seeded_tensor((2, 3, T, S, S), 74)
    print("anogan NetD (real)")
    seq_prefixes("layer1", hd.layer1, od.layer1, x)
    with torch.no_grad():
        z = OB.run_seq(od.layer1, OB.rbf(x))
    seq_prefixes("layer2", hd.layer2, od.layer2, z)
    # per-channel |mean| / sigma of every conv output that feeds a BatchNorm (where float32 statistic sums lose digits)
    with torch.no_grad():
        h = OB.rbf(x)
        mods = list(od.layer1) + list(od.layer2)
        for i, m in enumerate(mods):
            h2 = m(h)
            if isinstance(m, nn.Conv3d) and i + 1 < len(mods) and isinstance(mods[i + 1], nn.BatchNorm3d):
                mu, sd = h2.mean((0, 2, 3, 4)), h2.std((0, 2, 3, 4))
                print("  conv %d -> BatchNorm: max |mean|/sigma %.1f (median %.2f)" % (i, float((mu.abs() / sd).max()), float((mu.abs() / sd).median())))
            h = h2


if __name__ == "__main__":
    for w in (sys.argv[1:] or ["mygan", "ganomaly", "anogan"]):
        {"mygan": mygan, "ganomaly": ganomaly, "anogan": anogan}[w]()
