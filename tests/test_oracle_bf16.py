"""CPU checks of the bf16-faithful oracle mode (oracle/vfd_oracle/bf16.py): the rounding primitive, that the mode changes
nothing but roundings (it stays within bf16 distance of the float32 oracle and reaches the float32 master weights with a
full-precision gradient), and that every tensor the HIP path stores comes out bf16-representable."""
import torch

from vfd_oracle import anogan as OA
from vfd_oracle import bf16 as OB
from vfd_oracle import ganomaly as OG
from vfd_oracle import mygannet as OM
from vfd_oracle.weights import fill_module, seeded_normal, seeded_tensor


def _is_bf16(t):
    return torch.equal(t, OB.rbf(t))


def test_round_primitive_is_rne_and_straight_through():
    # 1 + 2^-8 is a tie between 1 and 1 + 2^-7 -> even (1.0); 1 + 3*2^-8 ties to 1 + 2^-6 ... checked against torch's own cast
    x = torch.tensor([1.0 + 2.0 ** -8, 1.0 + 3 * 2.0 ** -8, -1.0 - 2.0 ** -8, 3.14159, 1e-30, 65504.0])
    assert torch.equal(OB.rbf(x), x.to(torch.bfloat16).float())
    assert float(OB.rbf(x)[0]) == 1.0 and float(OB.rbf(x)[1]) == 1.0 + 2.0 ** -6
    w = torch.nn.Parameter(torch.tensor([0.1234567, -2.7182818]))
    y = (OB.wq(w) * torch.tensor([3.0, 5.0])).sum()
    y.backward()
    assert _is_bf16(OB.wq(w).detach()) and torch.equal(w.grad, torch.tensor([3.0, 5.0]))      # gradient reaches the master unrounded
    a = torch.tensor([1.0 + 2.0 ** -10], requires_grad=True)
    (OB.R(a) * (1.0 + 2.0 ** -10)).backward()
    assert float(a.grad) == 1.0 and float(OB.RF(a).detach()) == 1.0       # forward and gradient both rounded
    b = torch.tensor([1.0 + 2.0 ** -10], requires_grad=True)
    (OB.RB(b) * (1.0 + 2.0 ** -10)).backward()
    assert float(OB.RB(b).detach()) == 1.0 + 2.0 ** -10 and float(b.grad) == 1.0


def test_ganomaly_faithful_step_is_float32_step_up_to_bf16_rounding():
    opt = OG.make_opt(isize=32, ngf=8)
    x = seeded_tensor((4, 3, 32, 32), 5)
    out = {}
    for tag in ("f32", "bf16"):
        og, od = fill_module(OG.NetG(opt), 7), fill_module(OG.NetD(opt), 8)
        if tag == "f32":
            out[tag] = OG.step(og, od, *OG.make_optimizers(og, od, opt), x, opt)
        else:
            out[tag] = OG.step(OB.Faithful(og), OB.Faithful(od), *OG.make_optimizers(og, od, opt), OB.rbf(x), opt)
            assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in list(og.parameters()) + list(od.parameters()))
    (e32, f32), (e16, f16) = out["f32"], out["bf16"]
    assert _is_bf16(f16) and not _is_bf16(f32)
    for k, v in e32.items():
        assert e16[k] != v and abs(e16[k] - v) <= 5e-2 * max(abs(v), 1e-3), (k, e16[k], v)
    assert float((f16 - f32).pow(2).mean().sqrt() / f32.pow(2).mean().sqrt()) < 3e-2


def test_fusion_plan_rounding_points():
    """run_seq rounds once per fused HIP kernel: conv+act (no rounding in between), conv -> BN(+act) (conv output stored, then
    one rounding after the activation; the batch statistics are those of the conv output BEFORE it was rounded, as the HIP conv
    epilogue sums them), BN+act+AvgPool3d (pooled from the UNROUNDED activation, rounded once)."""
    torch.manual_seed(0)
    conv, act = torch.nn.Conv3d(8, 8, 3, 1, 1), torch.nn.LeakyReLU(0.2)
    bn, pool = torch.nn.BatchNorm3d(8).train(), torch.nn.AvgPool3d(2)
    x = OB.rbf(torch.randn(2, 8, 4, 8, 8))
    y = OB.run_seq([conv, act], x)
    assert torch.equal(y, OB.rbf(act(OB.conv_q(conv, x))))
    bn2 = torch.nn.BatchNorm3d(8).train()
    y = OB.run_seq([conv, bn, act, pool], x)
    t = OB.conv_q(conv, x)
    c = OB.rbf(t)
    want = OB.rbf(pool(act(OB.bn_from(bn2, c, t))))
    assert torch.equal(y, want) and not torch.equal(y, OB.rbf(pool(OB.rbf(act(OB.bn_from(torch.nn.BatchNorm3d(8).train(), c, t))))))
    assert torch.allclose(bn.running_mean, bn2.running_mean) and torch.allclose(bn.running_var, bn2.running_var) and int(bn.num_batches_tracked) == 1
    # a pool the BatchNorm pass cannot absorb (kernel 4) stays a pass of its own: two roundings
    bn3 = torch.nn.BatchNorm3d(8).train()
    pool3 = torch.nn.AvgPool3d((1, 4, 4))
    y = OB.run_seq([conv, bn3, act, pool3], x)
    assert torch.equal(y, OB.rbf(pool3(OB.rbf(act(OB.bn_from(torch.nn.BatchNorm3d(8).train(), c, t))))))
    # a BatchNorm that no conv epilogue feeds (Linear -> BatchNorm1d, anogan NetG.layer1): statistics of the stored tensor
    lin, bn1 = torch.nn.Linear(8, 16), torch.nn.BatchNorm1d(16).train()
    z = OB.rbf(torch.randn(6, 8))
    assert torch.equal(OB.run_seq([lin, bn1, torch.nn.ReLU()], z),
                       OB.rbf(torch.relu(torch.nn.BatchNorm1d(16).train()(OB.rbf(OB.conv_q(lin, z))))))


def test_bn_from_equals_torch_batchnorm_when_the_statistics_are_its_own():
    """bn_from(bn, x, t = x) is torch's training-mode BatchNorm, forward, backward and running statistics (its backward is the
    formula of csrc/bn.hip restated)."""
    torch.manual_seed(1)
    x = torch.randn(3, 5, 2, 6, 6) * 2 + 0.7
    a, b = torch.nn.BatchNorm3d(5).train(), torch.nn.BatchNorm3d(5).train()
    with torch.no_grad():
        for m in (a, b):
            m.weight.copy_(torch.tensor([1.0, 0.5, 2.0, 1.5, 0.1]))
            m.bias.copy_(torch.tensor([0.0, 0.3, -0.2, 1.0, 0.5]))
    xa, xb = x.clone().requires_grad_(), x.clone().requires_grad_()
    ya, yb = a(xa), OB.bn_from(b, xb, xb)
    g = torch.randn_like(x)
    ya.backward(g)
    yb.backward(g)
    assert torch.allclose(ya, yb, atol=2e-6) and torch.allclose(xa.grad, xb.grad, atol=2e-6)
    assert torch.allclose(a.weight.grad, b.weight.grad, rtol=1e-5, atol=1e-5) and torch.allclose(a.bias.grad, b.bias.grad, rtol=1e-5, atol=1e-5)
    assert torch.allclose(a.running_mean, b.running_mean, atol=1e-6) and torch.allclose(a.running_var, b.running_var, atol=1e-6)


def test_anogan_and_mygan_faithful_forward_run_and_store_bf16():
    T, S = 8, 16
    og, od = fill_module(OA.NetG(T, S), 1).train(), fill_module(OA.NetD(T, S), 2).train()
    for m in og.modules():
        if isinstance(m, torch.nn.Dropout):
            m.p = 0.0
    z, real = seeded_normal((2, 100), 10), seeded_tensor((2, 3, T, S, S), 20)
    g_opt, d_opt = OA.make_optimizers(og, od, 2e-5)
    ref, fake = OA.step(OB.Faithful(og), OB.Faithful(od), g_opt, d_opt, OB.rbf(real), OB.rbf(z))
    assert _is_bf16(fake) and all(v == v for v in ref.values())
    T, S = 16, 64
    ng, nd = fill_module(OM.NetG(), 3).train(), fill_module(OM.NetD(OM.make_args(T, S)), 4).train()
    for m in ng.modules():
        if isinstance(m, torch.nn.Dropout):
            m.p = 0.0
    inp = seeded_tensor((1, 3, T, S, S), 30)
    gt = (seeded_tensor((1, 1, T, S, S), 40, 0.0, 1.0) > 0.97).float()
    gf, pf = seeded_tensor((1, 3, T, S, S), 50), seeded_tensor((1, 3, T, S, S), 60)
    with torch.no_grad():
        p16 = OB.Faithful(ng)(inp)
        p32 = ng(inp)
        s_cls, s_feat, t_cls, t_feat = OB.Faithful(nd)(OB.rbf(gf), OB.rbf(pf))
    assert _is_bf16(p16) and _is_bf16(s_feat) and _is_bf16(t_feat)
    assert float((p16 - p32).pow(2).mean().sqrt() / p32.pow(2).mean().sqrt()) < 4e-2
