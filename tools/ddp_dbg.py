import os, sys, tempfile, types
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from vfd_gan_amd import dist as vdist, functional as F
from vfd_gan_amd.lib.data import synthetic_batch
from vfd_gan_amd.models import ganomaly as HG
rank, world = vdist.init_from_env(backend="gloo")
torch.cuda.set_device(0)
F.set_compute_dtype(torch.float32)
B, T, S = 2, 2, 32
args = types.SimpleNamespace(batchsize=B, nfr=T, isize=S, ich=3, lr=2e-4, beta1=0.5, w_adv=1, w_con=50, freq=10**9, ep=1, model="ganomaly", result_root=tempfile.mkdtemp(), gpu=[0])
torch.manual_seed(3)
m = HG.Ganomaly(args, None, opt=HG.make_opt(isize=S, ngf=8))
m.set_input(synthetic_batch(B, T, S, 3, seed=77))
def grads(reduce):
    for r in (m.reducer_g, m.reducer_d):
        r.world = world if reduce else 1
    m.forward_g(); m.forward_d()
    m.optimizer_g.zero_grad(); m.backward_g()
    gg = m.optimizer_g.grad_arena.clone()
    m.optimizer_d.zero_grad(); m.backward_d()
    gd = m.optimizer_d.grad_arena.clone()
    return gg, gd
g0, d0 = grads(False)
g1, d1 = grads(True)
torch.cuda.synchronize()
if rank == 0:
    print("G local vs reduced/2: max abs diff", float((g0 - g1 / world).abs().max()), "ref max", float(g0.abs().max()))
    print("D local vs reduced/2: max abs diff", float((d0 - d1 / world).abs().max()), "ref max", float(d0.abs().max()))
    bad = ((d0 - d1 / world).abs() > 1e-6 * d0.abs().max()).nonzero().flatten()
    print("D bad count", bad.numel(), "of", d0.numel(), "first", bad[:5].tolist(), "buckets", m.reducer_d.buckets[:3], "nb", len(m.reducer_d.buckets))
    badg = ((g0 - g1 / world).abs() > 1e-6 * g0.abs().max()).nonzero().flatten()
    print("G samples local/reduced:", [(round(float(g0[i]), 5), round(float(g1[i]), 5)) for i in badg[:6].tolist()])
    print("G slices", m.optimizer_g.slices()[:6], "bad idx range", int(badg.min()), int(badg.max()))
    good = ((g0 - g1 / world).abs() <= 1e-6 * g0.abs().max()).nonzero().flatten()
    nz = good[(g0[good] != 0)]
    print("G good nonzero count", nz.numel(), "range", (int(nz.min()), int(nz.max())) if nz.numel() else None)
    print("G bad count", badg.numel(), "of", g0.numel(), "first", badg[:5].tolist(), "nb", len(m.reducer_g.buckets))
torch.distributed.barrier(); torch.distributed.destroy_process_group()
