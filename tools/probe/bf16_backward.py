"""Single fused groups, forward + backward, bf16 HIP vs the bf16-faithful oracle: which group's backward is not modelled?"""
import os
import sys

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


def run(name, omods, hmods, xshape, loss="sum"):
    o = nn.Sequential(*omods).train()
    fill_module(o, 5)
    h = hnn.Sequential(*hmods).to(dev).train()
    h.load_state_dict(o.state_dict())
    F.invalidate_weight_cache()
    x = OB.rbf(seeded_tensor(xshape, 9))
    xo = x.clone().requires_grad_()
    yo = OB.run_seq(list(o), xo)
    g = OB.rbf(seeded_tensor(tuple(yo.shape), 11))
    yo.backward(g)
    xh = x.to(dev).requires_grad_()
    yh = h(F.to_cl(xh)).to_torch()
    yh.backward(g.to(dev))
    torch.cuda.synchronize()
    print("%-40s y differ %.3f%% rms %.2e | dx rms %.2e |" % (name, 100 * (yh.cpu() != yo).float().mean().item(), relrms(yh, yo), relrms(xh.grad, xo.grad)),
          "  ".join("%s %.2e" % (k.split(".", 1)[1] if "." in k else k, relrms(p.grad, q.grad)) for (k, p), (_, q) in zip(h.named_parameters(), o.named_parameters())))


run("conv k4s2 64->128, BN, LReLU(.2)", [nn.Conv2d(64, 128, 4, 2, 1, bias=False), nn.BatchNorm2d(128), nn.LeakyReLU(0.2)],
    [hnn.Conv2d(64, 128, 4, 2, 1, bias=False), hnn.BatchNorm2d(128), hnn.LeakyReLU(0.2)], (16, 64, 56, 56))
run("conv k4s2 3->64 + LReLU, conv 64->128", [nn.Conv2d(3, 64, 4, 2, 1, bias=False), nn.LeakyReLU(0.2), nn.Conv2d(64, 128, 4, 2, 1, bias=False)],
    [hnn.Conv2d(3, 64, 4, 2, 1, bias=False), hnn.LeakyReLU(0.2), hnn.Conv2d(64, 128, 4, 2, 1, bias=False)], (16, 3, 112, 112))
run("convT k4s2 256->128, BN, ReLU", [nn.ConvTranspose2d(256, 128, 4, 2, 1, bias=False), nn.BatchNorm2d(128), nn.ReLU()],
    [hnn.ConvTranspose2d(256, 128, 4, 2, 1, bias=False), hnn.BatchNorm2d(128), hnn.ReLU()], (16, 256, 14, 14))
run("convT k4s2 64->3 + Tanh", [nn.ConvTranspose2d(64, 3, 4, 2, 1, bias=False), nn.Tanh()],
    [hnn.ConvTranspose2d(64, 3, 4, 2, 1, bias=False), hnn.Tanh()], (16, 64, 56, 56))
run("conv k7 512->100 (final)", [nn.Conv2d(512, 100, 7, 1, 0, bias=False)], [hnn.Conv2d(512, 100, 7, 1, 0, bias=False)], (16, 512, 7, 7))
run("conv k7 512->1 + Sigmoid", [nn.Conv2d(512, 1, 7, 1, 0, bias=False), nn.Sigmoid()], [hnn.Conv2d(512, 1, 7, 1, 0, bias=False), hnn.Sigmoid()], (16, 512, 7, 7))
run("conv3d 64->64 k3 bias, BN, LReLU(64), AvgPool(2)", [nn.Conv3d(64, 64, 3, 1, 1), nn.BatchNorm3d(64), nn.LeakyReLU(64), nn.AvgPool3d(2)],
    [hnn.Conv3d(64, 64, 3, 1, 1), hnn.BatchNorm3d(64), hnn.LeakyReLU(64), hnn.AvgPool3d(2)], (2, 64, 8, 28, 28))
run("convT3d 128->64 s1 bias, conv3d 64->64 bias, BN, LReLU", [nn.ConvTranspose3d(128, 64, 3, 1, 1), nn.Conv3d(64, 64, 3, 1, 1), nn.BatchNorm3d(64), nn.LeakyReLU()],
    [hnn.ConvTranspose3d(128, 64, 3, 1, 1), hnn.Conv3d(64, 64, 3, 1, 1), hnn.BatchNorm3d(64), hnn.LeakyReLU()], (2, 128, 4, 28, 28))
