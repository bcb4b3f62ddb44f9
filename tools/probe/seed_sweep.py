import os, sys, tempfile, pathlib
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import torch
from util import relerr
import test_ganomaly_step as T
from vfd_gan_amd import functional as F
dev = torch.device("cuda", 0)
for seed in range(int(sys.argv[1]) if len(sys.argv) > 1 else 8):
    torch.manual_seed(1000 + seed)
    model, og, od, opt = T._build(pathlib.Path(tempfile.mkdtemp()), dev, torch.float32, 1, 2, 112, 8)
    x = torch.rand(2, 3, 112, 112) * 2 - 1
    fr, li, lo = og(x); pr, ft = od(x)
    (fr.mean() + li.pow(2).mean() + lo.mean() + pr.mean() + ft.pow(2).mean()).backward()
    xc = F.to_cl(x.to(dev))
    fh, lih, loh = model.netg(xc); ph, fth = model.netd(xc)
    loss = fh.to_torch().mean() + lih.to_torch().pow(2).mean() + loh.to_torch().mean() + ph.to_torch().mean() + fth.to_torch().pow(2).mean()
    loss.backward()
    worst = max(((relerr(p.grad, r.grad), k) for (k, p), (_, r) in list(zip(model.netg.named_parameters(), og.named_parameters())) + list(zip(model.netd.named_parameters(), od.named_parameters()))))
    print("seed %d: fwd %.2e  worst grad relerr %.2e (%s)" % (seed, relerr(fh.to_torch(), fr), worst[0], worst[1]), flush=True)
