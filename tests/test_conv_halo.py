"""GPU parity of the halo-tiled convolution kernel (conv_halo.hip): every eligible layer class — stride-1 3-D / 2-D /
(1,3,3) / (3,1,1) convolutions, stride-1 and strided transposed convolutions (= data gradients of strided ones), partial
tiles in every dimension, partial channel chunks (Cin not a multiple of 32), both channel tiles, fused activation /
BatchNorm statistics / activation-gradient multiply — against the torch CPU float32 op on the same bf16-rounded operands
AND against conv_igemm on identical device inputs (same products, other summation order: agreement to bf16 rounding)."""
import pytest
import torch
import torch.nn.functional as TF

from util import TOL, relerr

pytestmark = pytest.mark.gpu


def _rand(shape, seed, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return ((torch.rand(shape, generator=g) * 2 - 1) * scale).bfloat16().float()


CASES = [
    # name, x shape, Cout, k, s, p, op, transposed, bias
    ("c3d_k3_64to64", (2, 64, 6, 9, 20), 64, 3, 1, 1, 0, False, True),           # partial tiles in d, h, w
    ("c3d_k3_32to64", (1, 32, 8, 8, 32), 64, 3, 1, 1, 0, False, False),
    ("c3d_k3_64to32", (1, 64, 4, 12, 16), 32, 3, 1, 1, 0, False, True),          # 32-channel tile
    ("c3d_k3_86to40", (1, 86, 5, 7, 18), 40, 3, 1, 1, 0, False, True),           # partial last chunk (88 = 2*32 + 24), Cout 40
    ("c3d_k133_96to50", (2, 96, 3, 10, 16), 50, (1, 3, 3), 1, (0, 1, 1), 0, False, True),
    ("c3d_k311_40to64", (1, 40, 9, 6, 16), 64, (3, 1, 1), 1, (1, 0, 0), 0, False, False),
    ("c3d_k111_40to24", (1, 40, 4, 8, 16), 24, 1, 1, 0, 0, False, True),      # (1x1x1 over 16/24/32/48/64 channels is conv_cin8's)
    ("c2d_k3_64to64", (3, 64, 20, 24), 64, 3, 1, 1, 0, False, True),             # frames: 1 x 16 x 16 tile
    ("c2d_k3_48to20", (2, 48, 17, 30), 20, 3, 1, 1, 0, False, False),
    ("t3d_k3s1_128to64", (1, 128, 4, 8, 16), 64, 3, 1, 1, 0, True, True),        # anogan NetG layer3[1]
    ("t2d_k4s2_128to64_rows14", (2, 128, 14, 14), 64, 4, 2, 1, 0, True, False),   # ganomaly decoder pyramid (4 classes, 2x2 taps)
    ("t2d_k4s2_40to24", (2, 40, 13, 15), 24, 4, 2, 1, 0, True, True),
    ("t3d_k3s2_64to32", (1, 64, 3, 7, 12), 32, 3, 2, 1, 1, True, True),          # classes with 1 and 2 taps per dim
    # <= 16 output channels on volumes: the 16-channel tile (anogan NetG's 3-channel ends, models/anogan.py:70-72)
    ("t3d_k3s1_32to3", (1, 32, 5, 9, 18), 3, 3, 1, 1, 0, True, True),
    ("t3d_k3s2_64to3", (1, 64, 3, 7, 12), 3, 3, 2, 1, 1, True, False),
    ("c3d_k3_40to12", (2, 40, 4, 6, 16), 12, 3, 1, 1, 0, False, True),
    # frames under 2 x 2-tap classes (k4 s2 p1 transposed): conv_halo_rows (16 virtual rows x 16 pixels, one shared zero row
    # between frames, 4 workgroups per CU).  ganomaly's decoder / data-gradient shape; a partial channel chunk with partial
    # output channels, a 14-row period and partial width tiles; 5 frames of 13 rows (tiles span frames, 70 virtual rows)
    ("t2d_k4s2_128to64_rows", (3, 128, 28, 28), 64, 4, 2, 1, 0, True, False),
    ("t2d_k4s2_40to50_rows56", (2, 40, 13, 56), 50, 4, 2, 1, 0, True, True),
    ("t2d_k4s2_64to64_rows_frames", (5, 64, 13, 28), 64, 4, 2, 1, 0, True, True),
]
ROWS = {c[0] for c in CASES if "_rows" in c[0]}


def _run(case, halo_mode, dev, act=0, slope=0.0, stats=False, wg_mode=1):
    from vfd_gan_amd import _lib, functional as F
    name, xs, cout, k, s, p, op, tr, has_bias = case
    nd = len(xs) - 2
    cin = xs[1]
    kk = (k,) * nd if isinstance(k, int) else k
    wshape = ((cin, cout) if tr else (cout, cin)) + tuple(kk)
    x, w = _rand(xs, 1), _rand(wshape, 2, 0.2)
    b = _rand((cout,), 3, 0.5) if has_bias else None
    lib = _lib.load()
    prev = lib.vfd_conv_set_halo_mode(halo_mode)
    prev_wg = lib.vfd_wgrad_set_halo_mode(wg_mode)
    try:
        xd = x.to(dev).requires_grad_()
        wd = torch.nn.Parameter(w.to(dev))
        bd = torch.nn.Parameter(b.to(dev)) if has_bias else None
        sums = F.new_stats_buffer(cout, dev) if stats else None
        yc = F.conv(F.to_cl(xd, torch.bfloat16), wd, bd, s, p, op, tr, act, slope, stats=sums)
        names = None
        if halo_mode == 2:
            in_dhw, out_dhw = tuple(F.to_cl(xd.detach(), torch.bfloat16).t.shape[1:4]), tuple(yc.t.shape[1:4])
            desc = F._make_desc(xs[0], in_dhw, cin, out_dhw, cout, F._triple(kk, nd, 1), F._triple(s, nd, 1), F._triple(p, nd, 0),
                                tr, torch.bfloat16)
            names = F._conv_kernel_name(desc)
        y = yc.to_torch()
        gy = _rand(tuple(y.shape), 4)
        y.backward(gy.to(dev))
        torch.cuda.synchronize()
        return dict(y=y.detach().cpu(), raw=yc.t.detach().float().cpu(), gx=xd.grad.cpu(), gw=wd.grad.cpu(),
                    gb=bd.grad.cpu() if has_bias else None, sums=sums.cpu() if stats else None, name=names, x=x, w=w, b=b, gy=gy)
    finally:
        lib.vfd_conv_set_halo_mode(prev)
        lib.vfd_wgrad_set_halo_mode(prev_wg)


@pytest.mark.parametrize("case", CASES, ids=[c[0] for c in CASES])
def test_halo_forward_and_gradients(case, dev):
    name, xs, cout, k, s, p, op, tr, has_bias = case
    nd = len(xs) - 2
    h = _run(case, 2, dev)
    assert h["name"].startswith("conv_halo_rows<bf16" if name in ROWS else "conv_halo<bf16"), h["name"]        # the layer really went to the halo kernel
    g = _run(case, 1, dev)
    xr, wr = h["x"].clone().requires_grad_(), h["w"].clone().requires_grad_()
    br = h["b"].clone().requires_grad_() if has_bias else None
    fn = {(2, False): TF.conv2d, (3, False): TF.conv3d, (2, True): TF.conv_transpose2d, (3, True): TF.conv_transpose3d}[(nd, tr)]
    yr = fn(xr, wr, br, s, p, op) if tr else fn(xr, wr, br, s, p)
    yr.backward(h["gy"])
    tol = TOL[torch.bfloat16]
    assert relerr(h["y"], yr) < tol, ("fwd", relerr(h["y"], yr))
    assert relerr(h["gx"], xr.grad) < tol, ("dgrad", relerr(h["gx"], xr.grad))
    assert relerr(h["gw"], wr.grad) < tol, ("wgrad", relerr(h["gw"], wr.grad))
    # against conv_igemm on the same device inputs: f32 accumulation of identical bf16 products in another order, then
    # one rounding to bf16 -> the two kernels agree to one bf16 ulp of the tensor's scale (2^-8 relative to max |y|)
    assert relerr(h["y"], g["y"]) < 2 ** -8, ("fwd vs igemm", relerr(h["y"], g["y"]))
    assert relerr(h["gx"], g["gx"]) < 2 ** -8, ("dgrad vs igemm", relerr(h["gx"], g["gx"]))
    assert float(h["raw"][..., cout:].abs().sum()) == 0.0             # pad channels stay zero


@pytest.mark.parametrize("act,slope", [(1, 0.2), (2, 0.0), (3, 0.0), (1, 64.0)], ids=["lrelu", "sigmoid", "tanh", "lrelu64"])
def test_halo_fused_activation_and_statistics(act, slope, dev):
    """Epilogue fusions through the halo kernel: bias + activation, BatchNorm sum / sum of squares (replica rows)."""
    case = ("c3d_k3_64to64_act", (2, 64, 5, 9, 20), 64, 3, 1, 1, 0, False, True)
    h = _run(case, 2, dev, act, slope, stats=True)
    assert h["name"].startswith("conv_halo<bf16")
    pre = TF.conv3d(h["x"], h["w"], h["b"], 1, 1)
    ref = {1: TF.leaky_relu(pre, slope), 2: torch.sigmoid(pre), 3: torch.tanh(pre)}[act]
    assert relerr(h["y"], ref) < TOL[torch.bfloat16]
    from vfd_gan_amd import functional as F
    folded = h["sums"].view(F.STATS_REPLICAS, 2, 64).sum(0)
    assert relerr(folded[0], pre.sum(dim=(0, 2, 3, 4))) < 1e-4 and relerr(folded[1], (pre * pre).sum(dim=(0, 2, 3, 4))) < 1e-4


@pytest.mark.parametrize("act,slope", [(1, 0.2), (3, 0.0)], ids=["lrelu", "tanh"])
def test_halo_rows_fused_activation_and_statistics(act, slope, dev):
    """The same epilogue fusions through conv_halo_rows (7-wave tile; the zero rows between frames and the rows past the last
    frame must stay out of the statistics)."""
    case = ("t2d_k4s2_64to64_rows_act", (3, 64, 13, 28), 64, 4, 2, 1, 0, True, True)
    h = _run(case, 2, dev, act, slope, stats=True)
    assert h["name"].startswith("conv_halo_rows<bf16"), h["name"]
    pre = TF.conv_transpose2d(h["x"], h["w"], h["b"], 2, 1)
    ref = {1: TF.leaky_relu(pre, slope), 3: torch.tanh(pre)}[act]
    assert relerr(h["y"], ref) < TOL[torch.bfloat16]
    from vfd_gan_amd import functional as F
    folded = h["sums"].view(F.STATS_REPLICAS, 2, 64).sum(0)
    assert relerr(folded[0], pre.sum(dim=(0, 2, 3))) < 1e-4 and relerr(folded[1], (pre * pre).sum(dim=(0, 2, 3))) < 1e-4


def test_halo_activation_gradient_handover(dev):
    """vfd_conv_forward_mul through the halo kernel: Conv -> LeakyReLU -> Conv(64) -> Tanh -> Conv inside one Sequential;
    the consumers' data gradients (halo kernel, 64- and 32-channel tiles) carry the producers' activation gradients."""
    import vfd_gan_amd.nn as vnn
    from vfd_gan_amd import _lib, functional as F
    lib = _lib.load()
    torch.manual_seed(5)
    ref = torch.nn.Sequential(torch.nn.Conv3d(32, 64, 3, 1, 1), torch.nn.LeakyReLU(0.2), torch.nn.Conv3d(64, 32, 3, 1, 1), torch.nn.Tanh(),
                              torch.nn.Conv3d(32, 40, 3, 1, 1))
    mine = vnn.Sequential(vnn.Conv3d(32, 64, 3, 1, 1), vnn.LeakyReLU(0.2), vnn.Conv3d(64, 32, 3, 1, 1), vnn.Tanh(), vnn.Conv3d(32, 40, 3, 1, 1))
    with torch.no_grad():
        for prm in ref.parameters():
            prm.copy_(prm.bfloat16().float())
    mine.load_state_dict(ref.state_dict())
    mine.to(dev)
    F.invalidate_weight_cache()
    x = _rand((1, 32, 5, 9, 18), 7)
    xr = x.clone().requires_grad_()
    yr = ref(xr)
    gy = _rand(tuple(yr.shape), 8)
    yr.backward(gy)
    prev = lib.vfd_conv_set_halo_mode(2)
    try:
        F.set_compute_dtype(torch.bfloat16)
        xd = x.to(dev).requires_grad_()
        y = mine(F.to_cl(xd, torch.bfloat16)).to_torch()
        y.backward(gy.to(dev))
    finally:
        lib.vfd_conv_set_halo_mode(prev)
    assert relerr(y, yr) < 3e-2
    assert relerr(xd.grad, xr.grad) < 3e-2, relerr(xd.grad, xr.grad)
    for (k, pm), (_, pr) in zip(mine.named_parameters(), ref.named_parameters()):
        assert relerr(pm.grad, pr.grad) < 3e-2, (k, relerr(pm.grad, pr.grad))


WG_CASES = [
    # stride-1, 3 x 3 in-plane footprint, >= 33 channels on both sides: conv_wgrad_halo.hip
    ("c3d_k3_64to64", (2, 64, 6, 9, 20), 64, 3, 1, 1, 0, False, True),            # partial blocks in h and w, 3 depth taps
    ("c3d_k3_40to50", (1, 40, 5, 16, 16), 50, 3, 1, 1, 0, False, False),          # partial channel tiles on both sides
    ("c3d_k3_96to72", (1, 96, 3, 8, 32), 72, 3, 1, 1, 0, False, True),            # 2 x 2 channel tiles
    ("c3d_k133_86to96", (1, 86, 4, 11, 17), 96, (1, 3, 3), 1, (0, 1, 1), 0, False, True),   # one depth tap
    ("c2d_k3_64to48", (3, 64, 20, 24), 48, 3, 1, 1, 0, False, True),              # frames (D = 1)
    ("t3d_k3s1_128to64", (1, 128, 4, 8, 16), 64, 3, 1, 1, 0, True, True),         # transposed: (S, G) = (x, dy)
    # gathered operand with <= 32 channels: 64-byte halo rows, two depth taps per work item (anogan NetD 32 -> 64)
    ("c3d_k3_32to64", (2, 32, 6, 9, 20), 64, 3, 1, 1, 0, False, True),
    ("c3d_k3_24to40", (1, 24, 5, 16, 16), 40, 3, 1, 1, 0, False, False),
    # ... with ONE depth tap (mygan's (1,3,3) 32 -> 57 factor): the item's second plane is masked off
    ("c3d_k133_32to57", (2, 32, 4, 11, 17), 57, (1, 3, 3), 1, (0, 1, 1), 0, False, True),
    ("c2d_k3_24to64", (3, 24, 20, 24), 64, 3, 1, 1, 0, False, False),
]


@pytest.mark.parametrize("case", WG_CASES, ids=[c[0] for c in WG_CASES])
def test_halo_filter_gradient(case, dev):
    """conv_wgrad_halo against the torch CPU filter gradient on the same bf16-rounded operands and against conv_wgrad's
    per-tap gather on identical device inputs (float32 slabs of the same bf16 products in another order: 1e-5)."""
    from vfd_gan_amd import _lib, functional as F
    import ctypes
    name, xs, cout, k, s, p, op, tr, has_bias = case
    nd = len(xs) - 2
    h = _run(case, 1, dev, wg_mode=2)
    g = _run(case, 1, dev, wg_mode=1)
    # the layer really is eligible: the workspace geometry differs between the two modes (one slab set per CU vs per tile round)
    lib = _lib.load()
    kk = (k,) * nd if isinstance(k, int) else k
    xc = F.to_cl(h["x"].to(dev), torch.bfloat16)
    out_dhw = tuple(h["raw"].shape[1:4])
    desc = F._make_desc(xs[0], tuple(xc.t.shape[1:4]), xs[1], out_dhw, cout, F._triple(kk, nd, 1), F._triple(s, nd, 1), F._triple(p, nd, 0),
                        tr, torch.bfloat16)
    names = []
    for mode in (2, 1):
        prev = lib.vfd_wgrad_set_halo_mode(mode)
        try:
            buf = ctypes.create_string_buffer(64)
            _lib.check(lib.vfd_wgrad_kernel_name(ctypes.byref(desc), buf, 64))
            names.append(buf.value.decode())
        finally:
            lib.vfd_wgrad_set_halo_mode(prev)
    assert names[0] == "conv_wgrad_halo<bf16>" and names[1].startswith("conv_wgrad<bf16"), names
    xr, wr = h["x"].clone().requires_grad_(), h["w"].clone().requires_grad_()
    fn = {(2, False): TF.conv2d, (3, False): TF.conv3d, (2, True): TF.conv_transpose2d, (3, True): TF.conv_transpose3d}[(nd, tr)]
    yr = fn(xr, wr, None, s, p, op) if tr else fn(xr, wr, None, s, p)
    yr.backward(h["gy"])
    assert relerr(h["gw"], wr.grad) < TOL[torch.bfloat16], ("wgrad", relerr(h["gw"], wr.grad))
    assert relerr(h["gw"], g["gw"]) < 1e-5, ("wgrad vs conv_wgrad", relerr(h["gw"], g["gw"]))
