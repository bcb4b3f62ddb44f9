"""GPU parity of the whole ganomaly training step (G fwd, 4 D fwd, backward_g, Adam(G), backward_d, Adam(D))
against the CPU oracle on identical clips and weights: losses, generated frames, BatchNorm running statistics
(the step runs netD four times -> four momentum updates) and post-Adam parameters after 1 and 3 steps."""
import types

import pytest
import torch

from util import relerr, relrms

pytestmark = pytest.mark.gpu


def _args(tmp, B, T, S):
    return types.SimpleNamespace(batchsize=B, nfr=T, isize=S, ich=3, lr=2e-4, beta1=0.5, w_adv=1, w_con=50, freq=10 ** 9,
                                 ep=1, model="ganomaly", result_root=str(tmp), gpu=[0], steps_per_epoch=1)


def _build(tmp, dev, dt, B, T, S, ngf, extralayers=0, seed=7):
    from vfd_gan_amd import functional as F
    from vfd_gan_amd.models import ganomaly as HG
    from vfd_oracle import ganomaly as OG
    from vfd_oracle.weights import fill_module
    F.set_compute_dtype(dt)
    opt = OG.make_opt(isize=S, ngf=ngf, extralayers=extralayers)
    og, od = OG.NetG(opt), OG.NetD(opt)
    fill_module(og, seed)
    fill_module(od, seed + 1)
    model = HG.Ganomaly(_args(tmp, B, T, S), None, opt=HG.make_opt(isize=S, ngf=ngf, extralayers=extralayers))
    # identical state_dict keys -> reference-format checkpoints load
    assert list(model.netg.state_dict().keys()) == list(og.state_dict().keys())
    assert list(model.netd.state_dict().keys()) == list(od.state_dict().keys())
    model.netg.load_state_dict(og.state_dict())
    model.netd.load_state_dict(od.state_dict())
    F.invalidate_weight_cache()
    return model, og, od, opt


@pytest.mark.parametrize("dt,S,ngf,extra", [(torch.float32, 32, 16, 0), (torch.float32, 48, 8, 1), (torch.bfloat16, 32, 16, 0)],
                         ids=["f32_s32", "f32_s48_extra", "bf16_s32"])
def test_ganomaly_step_parity(dt, S, ngf, extra, dev, tmp_path):
    from vfd_gan_amd.lib.data import synthetic_batch
    from vfd_oracle import ganomaly as OG
    B, T = 2, 4
    model, og, od, opt = _build(tmp_path, dev, dt, B, T, S, ngf, extra)
    opt_g, opt_d = OG.make_optimizers(og, od, opt)
    f32 = dt == torch.float32
    for it in range(3):
        batch = synthetic_batch(B, T, S, 3, seed=100 + it)
        x = OG.fold_frames(batch[0])
        errs_ref, fake_ref = OG.step(og, od, opt_g, opt_d, x, opt)
        model.set_input(batch)
        model.optimize_params(check_collapse=False)
        errs = model.errors()
        tol_l = 1e-4 if f32 else 5e-2
        for k, v in errs_ref.items():
            got = errs["%s/%s/train" % (k[4], k)]
            assert abs(got - v) <= tol_l * max(abs(v), 1e-3), (it, k, got, v)
        if f32:
            assert relerr(model.fake.to_torch(), fake_ref) < 2e-4, it
        else:  # bf16 storage: stated tolerance 3e-2 relative RMS, 0.2 max-norm
            assert relrms(model.fake.to_torch(), fake_ref) < 3e-2 and relerr(model.fake.to_torch(), fake_ref) < 0.2, it
        if it in (0, 2):
            sdg, sdd = model.netg.state_dict(), model.netd.state_dict()
            lr = opt.lr
            for (k, v), (_, r) in list(zip(sdg.items(), og.state_dict().items())) + list(zip(sdd.items(), od.state_dict().items())):
                if "num_batches_tracked" in k:
                    assert int(v) == int(r), k
                elif "running_" in k:
                    assert relerr(v, r) < (5e-4 if f32 else 5e-2), (it, k, relerr(v, r))
                else:
                    # Adam's early steps move every weight by ~lr whatever the gradient's size, so a weight whose
                    # gradient is ~0 amplifies rounding noise: state the tolerance in units of lr per step taken
                    d = (v.detach().cpu().double() - r.detach().double()).abs()
                    assert float(d.max()) <= (0.5 if f32 else 6.0) * lr * (it + 1), (it, k, float(d.max()))
                    assert float(d.mean()) <= (0.01 if f32 else 0.35) * lr * (it + 1), (it, k, float(d.mean()))


def test_ganomaly_generalised_pyramid_112(dev, tmp_path):
    """isize=112 (BASELINE config 2): 112->56->28->14->7, final kernel 7; per-net forward/backward parity vs the oracle."""
    from vfd_gan_amd import functional as F
    from vfd_oracle import ganomaly as OG
    model, og, od, opt = _build(tmp_path, dev, torch.float32, 1, 2, 112, 8)
    torch.manual_seed(112)
    x = torch.rand(2, 3, 112, 112) * 2 - 1
    fr, li, lo = og(x)
    pr, ft = od(x)
    (fr.mean() + li.pow(2).mean() + lo.mean() + pr.mean() + ft.pow(2).mean()).backward()
    xc = F.to_cl(x.to(dev))
    fh, lih, loh = model.netg(xc)
    ph, fth = model.netd(xc)
    assert tuple(fh.shape) == (2, 3, 112, 112) and tuple(lih.shape) == (2, 100, 1, 1)
    loss = fh.to_torch().mean() + lih.to_torch().pow(2).mean() + loh.to_torch().mean() + ph.to_torch().mean() + fth.to_torch().pow(2).mean()
    loss.backward()
    assert relerr(fh.to_torch(), fr) < 1e-4 and relerr(loh.to_torch(), lo) < 1e-4 and relerr(fth.to_torch(), ft) < 1e-4
    errs = {k: relerr(p.grad, r.grad) for (k, p), (_, r) in list(zip(model.netg.named_parameters(), og.named_parameters())) +
            list(zip(model.netd.named_parameters(), od.named_parameters()))}
    bad = {k: v for k, v in errs.items() if not v < 2e-3}
    assert not bad, bad


def test_graph_replay_equals_eager(dev, tmp_path):
    """The hipGraph-captured step (vfd_gan_amd.graph.GraphedStep) is the same arithmetic as the eager step: f32 mode
    has no atomics, so losses and parameters agree bit for bit over several steps (incl. the device-side Adam counter)."""
    from vfd_gan_amd.graph import GraphedStep
    from vfd_gan_amd.lib.data import synthetic_batch
    B, T, S = 2, 4, 32
    a, og, od, _ = _build(tmp_path, dev, torch.float32, B, T, S, 16)
    b, _, _, _ = _build(tmp_path, dev, torch.float32, B, T, S, 16)
    batch0, batch1 = synthetic_batch(B, T, S, 3, seed=200), synthetic_batch(B, T, S, 3, seed=201)
    # eager reference: 2 steps on batch0 (= the 2 warm-up steps; capturing records the step without running it)
    # then 3 steps on batch1 (= 3 replays)
    a.set_input(batch0)
    for _ in range(2):
        a.optimize_params(check_collapse=False)
    a.set_input(batch1)
    for _ in range(3):
        a.optimize_params(check_collapse=False)
    b.set_input(batch0)
    step = GraphedStep(b, warmup=2).capture()
    step.load_input(batch1)                 # new clips go into the static buffers the graph reads
    for _ in range(3):
        step.replay()
    ea, eb = a.errors(), b.errors()
    for k in ea:
        assert ea[k] == eb[k], (k, ea[k], eb[k])
    for (k, v), (_, r) in zip(a.netg.state_dict().items(), b.netg.state_dict().items()):
        assert torch.equal(v, r), k
    for (k, v), (_, r) in zip(a.netd.state_dict().items(), b.netd.state_dict().items()):
        assert torch.equal(v, r), k
    assert int(b.optimizer_g._step_dev.item()) == 5
