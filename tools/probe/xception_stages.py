"""Stage-by-stage float32 deviation of the HIP Xception (models/xception.py) from the oracle's, same weights and clip."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402

from util import relerr, relrms  # noqa: E402
from vfd_gan_amd import functional as F  # noqa: E402
from vfd_gan_amd.models.xception import Xception as HX  # noqa: E402
from vfd_oracle import xception as OX  # noqa: E402
from vfd_oracle.weights import fill_module, seeded_tensor  # noqa: E402

dev = torch.device("cuda", 0)
F.set_compute_dtype(torch.float32)
o = fill_module(OX.Xception(), 72).train()
h = HX().to(dev).train()
h.load_state_dict(o.state_dict())
for net in (o, h):
    for m in net.modules():
        if isinstance(m, torch.nn.Dropout):
            m.p = 0.0
F.invalidate_weight_cache()
x = seeded_tensor((1, 3, 16, 128, 128), 73)
names = ["bn1", "bn2"] + ["block%d" % i for i in range(1, 13)] + ["conv3", "bn3", "conv4", "bn4", "uconv1", "uconv2", "uconv3", "uconv4", "conv_last"]
outs = {"o": {}, "h": {}}
for tag, net in (("o", o), ("h", h)):
    for n in names:
        getattr(net, n).register_forward_hook(lambda m, i, out, n=n, tag=tag: outs[tag].__setitem__(n, out.to_torch().cpu() if hasattr(out, "to_torch") else out.detach()))
with torch.no_grad():
    po = o(x)
    ph = h(F.to_cl(x.to(dev))).to_torch().cpu()
for n in names:
    if n in outs["o"] and n in outs["h"]:
        a, b = outs["h"][n], outs["o"][n]
        if n.startswith("bn"):
            b = b.relu()          # the HIP BatchNorm pass carries the ReLU that follows it
        print("%-10s relerr %.3e relrms %.3e  max|o| %.3g" % (n, relerr(a, b), relrms(a, b), b.abs().max().item()))
print("predict relerr %.3e" % relerr(ph, po))
