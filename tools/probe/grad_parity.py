"""Per-parameter gradient deviation of one ganomaly optimize_params (HIP, f32 and bf16) from the CPU oracle at the
benchmarked network (ngf=64, isize 112): tells rounding noise (bf16 only, spread over all layers) from a wrong kernel
(f32 too, or one layer only)."""
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
from vfd_gan_amd.models import ganomaly as HG  # noqa: E402
from vfd_oracle import ganomaly as OG  # noqa: E402
from vfd_oracle.weights import fill_module  # noqa: E402

B, T, S, ngf = 1, int(os.environ.get("T", "16")), 112, 64
opt = OG.make_opt(isize=S, ngf=ngf)
og, od = fill_module(OG.NetG(opt), 7), fill_module(OG.NetD(opt), 8)
sdg = {k: v.clone() for k, v in og.state_dict().items()}
sdd = {k: v.clone() for k, v in od.state_dict().items()}
batch = synthetic_batch(B, T, S, 3, seed=321)
ref, fake_ref = OG.step(og, od, *OG.make_optimizers(og, od, opt), OG.fold_frames(batch[0]), opt)
for dt in (torch.float32, torch.bfloat16):
    F.set_compute_dtype(dt)
    args = types.SimpleNamespace(batchsize=B, nfr=T, isize=S, ich=3, lr=2e-4, beta1=0.5, w_adv=1, w_con=50, freq=10 ** 9, ep=1,
                                 model="ganomaly", result_root=tempfile.mkdtemp(), gpu=[0], steps_per_epoch=1)
    m = HG.Ganomaly(args, None, opt=HG.make_opt(isize=S, ngf=ngf))
    m.netg.load_state_dict(sdg)
    m.netd.load_state_dict(sdd)
    F.invalidate_weight_cache()
    m.set_input(batch)
    m.optimize_params(check_collapse=False)
    errs = m.errors()
    print(dt, {k: (round(errs["%s/%s/train" % (k[4], k)], 6), round(v, 6)) for k, v in ref.items()})
    print("  fake relrms %.3e" % relrms(m.fake.to_torch(), fake_ref))
    for (k, p), (_, r) in list(zip(m.netg.named_parameters(), og.named_parameters())) + list(zip(m.netd.named_parameters(), od.named_parameters())):
        print("  %-52s rms %.3e  max %.3e  |g|max %.3e" % (k, relrms(p.grad, r.grad), relerr(p.grad, r.grad), float(r.grad.abs().max())))
