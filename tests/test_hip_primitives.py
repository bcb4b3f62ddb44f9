"""GPU parity of every HIP primitive against the torch CPU float32 op the reference calls (SURVEY.md 2.1 K1-K6,
E1-E6, L1, O1), at odd shapes (Cout=21, T=1, H=7 ...), through the C ABI (ctypes) only."""
import pytest
import torch
import torch.nn.functional as TF

from util import TOL, relerr, relrms

pytestmark = pytest.mark.gpu

DTYPES = [torch.float32, torch.bfloat16]


def _rand(shape, seed, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.rand(shape, generator=g) * 2 - 1) * scale


CONV_CASES = [
    # name, x shape (N,C,D,H,W) or (N,C,H,W), Cout, k, s, p, op, transposed, bias
    ("K1_3x3x3", (2, 5, 4, 9, 7), 21, 3, 1, 1, 0, False, True),
    ("K1_big", (1, 64, 3, 12, 12), 72, 3, 1, 1, 0, False, True),
    ("K2_1x3x3", (2, 3, 3, 10, 10), 21, (1, 3, 3), 1, (0, 1, 1), 0, False, True),
    ("K3_3x1x1", (2, 21, 5, 6, 6), 32, (3, 1, 1), 1, (1, 0, 0), 0, False, True),
    ("K4_1x1x1", (2, 14, 2, 5, 5), 3, 1, 1, 0, 0, False, True),
    ("K5_convT_s2", (2, 12, 2, 5, 6), 9, 3, 2, 1, 1, True, True),
    ("K5_convT_s1", (1, 16, 3, 6, 6), 8, 3, 1, 1, 0, True, True),
    ("K6_conv2d_k4s2", (3, 3, 16, 16), 16, 4, 2, 1, 0, False, False),
    ("K6_conv2d_k4s2_wide", (2, 40, 14, 14), 136, 4, 2, 1, 0, False, False),
    ("K6_conv2d_final_k7", (5, 24, 7, 7), 100, 7, 1, 0, 0, False, False),
    ("K6_convT2d_init_k7", (5, 100, 1, 1), 24, 7, 1, 0, 0, True, False),
    ("K6_convT2d_k4s2", (2, 24, 7, 7), 12, 4, 2, 1, 0, True, False),
    ("K6_convT2d_to3", (2, 16, 8, 8), 3, 4, 2, 1, 0, True, False),
    ("K6_conv2d_extra_k3", (2, 16, 8, 8), 16, 3, 1, 1, 0, False, False),
    # >= 256 channels on the strided side: the 8-wave 256 x 128 filter-gradient tile (partial second tile)
    ("K6_conv2d_k4s2_cout264", (2, 24, 10, 10), 264, 4, 2, 1, 0, False, True),
    ("K6_convT2d_k4s2_cin260", (2, 260, 5, 5), 20, 4, 2, 1, 0, True, False),
    # <= 64 channels on the strided side with >= 256 filter columns: the 64 x 256 filter-gradient tile (128-byte rows)
    ("K1_conv3d_k3_64to64", (1, 64, 3, 9, 9), 64, 3, 1, 1, 0, False, True),
    ("K1_conv3d_k3_40to50", (2, 40, 2, 6, 7), 50, 3, 1, 1, 0, False, False),
    # >= 256 (filter row, 64-gather-channel chunk) pairs: the LDS-turned slab fold (wgrad_reduce_turn_kernel) — whole and partial
    # (60 of 64) channel chunks, 16 / 9 / 49 taps, regular and transposed, with and without the bias rows
    ("K6_conv2d_k4s2_64to256_turn", (2, 64, 8, 8), 256, 4, 2, 1, 0, False, True),
    ("K6_conv2d_k3_60to256_turn", (2, 60, 6, 6), 256, 3, 1, 1, 0, False, False),
    ("K6_conv2d_k7_128to130_turn", (3, 128, 7, 7), 130, 7, 1, 0, 0, False, True),
    ("K6_convT2d_k4s2_256to64_turn", (2, 256, 4, 4), 64, 4, 2, 1, 0, True, True),
    # stride-1 layers with <= 8 output channels over >= 16 input channels: the filter gradient runs in the transposed form (x's
    # channels as tile rows, dy gathered) and is permuted back (functional._Conv.backward `swap`): mygan's conv_last shape, 2-D, k1
    ("K1_conv3d_last_32to1_k3", (2, 32, 4, 10, 12), 1, 3, 1, 1, 0, False, True),
    ("K6_conv2d_k3_24to3_swap", (2, 24, 9, 11), 3, 3, 1, 1, 0, False, False),
    ("K4_conv3d_k133_40to5_swap", (1, 40, 3, 8, 10), 5, (1, 3, 3), 1, (0, 1, 1), 0, False, True),
    # 517 tiles of 128c x 256p on 512 workgroup slots (more than one round of workgroups, XCD-ordered ids)
    ("K6_conv2d_k1_517tiles", (3, 8, 210, 210), 128, 1, 1, 0, 0, False, True),
    # thin-channel pyramid ends (bf16: conv_small.hip; f32: implicit GEMM) and their data gradients
    ("K6_conv2d_first_3to64", (3, 3, 20, 14), 64, 4, 2, 1, 0, False, True),
    ("K6_conv2d_first_3to40", (2, 3, 18, 22), 40, 4, 2, 1, 0, False, False),
    ("K1_conv3d_first_3to32", (2, 3, 6, 10, 10), 32, 4, 2, 1, 0, False, True),
    ("K6_convT2d_last_64to3", (3, 64, 9, 7), 3, 4, 2, 1, 0, True, True),
    ("K6_convT2d_last_32to1", (2, 32, 6, 6), 1, 4, 2, 1, 0, True, False),
    ("K6_convT2d_last_128to4", (1, 128, 5, 9), 4, 4, 2, 1, 0, True, True),
    ("K6_convT2d_last_64to3_tall", (1, 64, 37, 5), 3, 4, 2, 1, 0, True, False),
    # <= 8 input channels, any small filter (bf16: conv_cin8, round 2): 3-D k3 stems, the (1,3,3) / (3,1,1) / 1x1x1 factors of
    # the (2+1)D stems, few output channels, and stride-1 transposed convolutions (= data gradients of <= 8-channel outputs)
    ("K1_cin8_3d_k3_3to32", (2, 3, 6, 20, 20), 32, 3, 1, 1, 0, False, True),
    ("K2_cin8_k133_3to14", (2, 3, 4, 24, 24), 14, (1, 3, 3), 1, (0, 1, 1), 0, False, True),
    ("K3_cin8_k311_2to32", (1, 2, 8, 24, 24), 32, (3, 1, 1), 1, (1, 0, 0), 0, False, False),
    ("K4_cin8_k111_3to2", (2, 3, 4, 24, 24), 2, 1, 1, 0, 0, False, True),
    ("K5_cin8_T_k3s1_1to32", (1, 1, 6, 20, 20), 32, 3, 1, 1, 0, True, True),
    ("K5_cin8_T_k311s1_2to21", (1, 2, 6, 12, 12), 21, (3, 1, 1), 1, (1, 0, 0), 0, True, False),
    # pointwise (1x1x1) factors over 9..32 channels (bf16: conv_cin8 on the granule row, conv_small.hip), forward, transposed
    # (the data gradient form), with bias; 14 -> 32, 32 -> 14, 24 -> 27, 32 -> 54 as in mygan's discriminators
    ("K4_pw_k111_14to32", (2, 14, 3, 10, 12), 32, 1, 1, 0, 0, False, True),
    ("K4_pw_T_k111_32to14", (2, 32, 3, 10, 12), 14, 1, 1, 0, 0, True, False),
    ("K4_pw_k111_24to27", (1, 24, 2, 9, 7), 27, 1, 1, 0, 0, False, True),
    ("K4_pw_k111_32to54", (1, 32, 4, 12, 12), 54, 1, 1, 0, 0, False, False),
    ("K4_pw_k111_64to54", (2, 64, 2, 9, 12), 54, 1, 1, 0, 0, False, True),
    ("K4_pw_T_k111_54to64", (1, 54, 2, 9, 12), 64, 1, 1, 0, 0, True, False),      # CPAD(54) = 56: not a granule patch -> conv_igemm
    ("K4_pw_T_k111_48to40", (1, 48, 2, 9, 12), 40, 1, 1, 0, 0, True, True),
    # ... and the temporal (3,1,1) factors (depth taps x granules), regular and as stride-1 data gradients (depth taps reversed)
    ("K3_gr_k311_21to32", (2, 21, 5, 8, 12), 32, (3, 1, 1), 1, (1, 0, 0), 0, False, True),
    ("K3_gr_k311_27to64", (1, 27, 4, 9, 10), 64, (3, 1, 1), 1, (1, 0, 0), 0, False, False),
    ("K3_gr_T_k311_32to21", (2, 32, 5, 8, 12), 21, (3, 1, 1), 1, (1, 0, 0), 0, True, True),
    ("K3_gr_T_k311_64to27", (1, 64, 4, 9, 10), 27, (3, 1, 1), 1, (1, 0, 0), 0, True, False),
    ("K3_gr_k311s2_24to40", (1, 24, 6, 6, 8), 40, (3, 1, 1), (2, 1, 1), (1, 0, 0), 0, False, False),
]


@pytest.mark.parametrize("dt", DTYPES, ids=["f32", "bf16"])
@pytest.mark.parametrize("case", CONV_CASES, ids=[c[0] for c in CONV_CASES])
def test_conv_family(case, dt, dev):
    from vfd_gan_amd import functional as F
    name, xs, cout, k, s, p, op, tr, has_bias = case
    nd = len(xs) - 2
    cin = xs[1]
    kk = (k,) * nd if isinstance(k, int) else k
    wshape = ((cin, cout) if tr else (cout, cin)) + tuple(kk)
    x = _rand(xs, 1)
    w = _rand(wshape, 2, 0.2)
    b = _rand((cout,), 3, 0.5) if has_bias else None
    if dt == torch.bfloat16:  # compare against the same bf16-rounded operands (products are then exact in f32)
        x = x.bfloat16().float()
        w = w.bfloat16().float()
    xr, wr = x.clone().requires_grad_(), w.clone().requires_grad_()
    br = b.clone().requires_grad_() if has_bias else None
    fn = {(2, False): TF.conv2d, (3, False): TF.conv3d, (2, True): TF.conv_transpose2d, (3, True): TF.conv_transpose3d}[(nd, tr)]
    yr = fn(xr, wr, br, s, p, op) if tr else fn(xr, wr, br, s, p)
    gy = _rand(tuple(yr.shape), 4)
    if dt == torch.bfloat16:
        gy = gy.bfloat16().float()
    yr.backward(gy)

    xd = x.to(dev).requires_grad_()
    wd = torch.nn.Parameter(w.to(dev))
    bd = torch.nn.Parameter(b.to(dev)) if has_bias else None
    xc = F.to_cl(xd, dt)
    yc = F.conv(xc, wd, bd, s, p, op, tr)
    y = yc.to_torch()
    assert tuple(y.shape) == tuple(yr.shape)
    y.backward(gy.to(dev))
    torch.cuda.synchronize()
    tol = TOL[dt]
    assert relerr(y, yr) < tol, ("fwd", relerr(y, yr))
    assert relerr(xd.grad, xr.grad) < tol, ("dgrad", relerr(xd.grad, xr.grad))
    assert relerr(wd.grad, wr.grad) < tol, ("wgrad", relerr(wd.grad, wr.grad))
    if has_bias:
        assert relerr(bd.grad, br.grad) < tol, ("bgrad", relerr(bd.grad, br.grad))
    # pad channels of the raw block stay zero
    assert float(yc.t.detach()[..., yc.C:].abs().sum()) == 0.0


@pytest.mark.parametrize("sub", [0, 12, 13, 14, 15, 16])
@pytest.mark.parametrize("tr", [False, True], ids=["conv", "convT"])
def test_conv_igemm_flexible_pixel_tile(tr, sub, dev):
    """conv_igemm<bf16,256c x 256p> with a run-time pixel extent of 12..16 sub-tiles of 16 pixels (vfd_conv_set_tile_sub; 0 =
    chosen per layer): k4 s2 (transposed: four output classes), 200 output channels (a partial channel tile), ragged pixel count
    (several tiles, a partial last one), with bias + LeakyReLU + BatchNorm statistics in the epilogue: the same values
    whatever the tile extent, and equal to torch's on the bf16-rounded operands."""
    from vfd_gan_amd import _lib, functional as F
    lib = _lib.load()
    cin, cout = 40, 200
    xs = (3, cin, 26, 22) if not tr else (3, cin, 13, 11)
    x = _rand(xs, 11).bfloat16().float()
    w = _rand(((cin, cout) if tr else (cout, cin)) + (4, 4), 12, 0.2).bfloat16().float()
    b = _rand((cout,), 13, 0.5)
    yr = TF.conv_transpose2d(x, w, b, 2, 1) if tr else TF.conv2d(x, w, b, 2, 1)
    prev = lib.vfd_conv_set_tile_sub(sub)
    try:
        xc = F.to_cl(x.to(dev), torch.bfloat16)
        stats = F.new_stats_buffer(cout, dev)
        yc = F.conv(xc, torch.nn.Parameter(w.to(dev)), torch.nn.Parameter(b.to(dev)), 2, 1, 0, tr, _lib.ACT_LRELU, 0.2, stats)
        name = F._conv_kernel_name(F._make_desc(3, (1,) + xs[2:], cin, (1,) + tuple(yr.shape[2:]), cout, (1, 4, 4), (1, 2, 2), (0, 1, 1), tr, torch.bfloat16))
        torch.cuda.synchronize()
    finally:
        lib.vfd_conv_set_tile_sub(prev)
    assert name == "conv_igemm<bf16,256c_x_256p>", name
    want = TF.leaky_relu(yr, 0.2)
    assert relerr(yc.to_torch(), want) < TOL[torch.bfloat16]
    assert float(yc.t.detach()[..., cout:].abs().sum()) == 0.0
    st = stats.view(F.STATS_REPLICAS, 2, F.cpad(cout)).sum(0).cpu()
    yf = yr.permute(1, 0, 2, 3).reshape(cout, -1).double()
    assert relerr(st[0, :cout], yf.sum(1).float()) < 1e-3 and relerr(st[1, :cout], (yf * yf).sum(1).float()) < 1e-3


@pytest.mark.parametrize("dt", DTYPES, ids=["f32", "bf16"])
def test_conv_fused_act_and_stats(dt, dev):
    from vfd_gan_amd import _lib, functional as F
    x = _rand((2, 6, 3, 8, 8), 11)
    w = _rand((13, 6, 3, 3, 3), 12, 0.3)
    b = _rand((13,), 13)
    if dt == torch.bfloat16:
        x, w = x.bfloat16().float(), w.bfloat16().float()
    pre = TF.conv3d(x, w, b, 1, 1)
    for act, slope, ref in [(_lib.ACT_LRELU, 0.2, TF.leaky_relu(pre, 0.2)), (_lib.ACT_SIGMOID, 0.0, torch.sigmoid(pre)),
                            (_lib.ACT_TANH, 0.0, torch.tanh(pre)), (_lib.ACT_LRELU, 64.0, TF.leaky_relu(pre, 64.0))]:
        sums = F.new_stats_buffer(13, dev)
        yc = F.conv(F.to_cl(x.to(dev), dt), torch.nn.Parameter(w.to(dev)), torch.nn.Parameter(b.to(dev)), 1, 1, 0, False, act,
                    slope, stats=sums)
        assert relerr(yc.to_torch(), ref) < TOL[dt]
        s1 = pre.sum(dim=(0, 2, 3, 4))
        s2 = (pre * pre).sum(dim=(0, 2, 3, 4))
        folded = sums.view(F.STATS_REPLICAS, 2, 16).sum(0)
        assert relerr(folded[0, :13], s1) < 1e-4 and relerr(folded[1, :13], s2) < 1e-4


def test_conv_cin8_stats_and_kernel_names(dev):
    """conv_cin8 with BatchNorm sums in its epilogue (bf16): the 3-channel stems of anogan's NetD (Conv3d(3,32,3) + BN) and of
    the (2+1)D blocks run on it; sums against the float32 convolution of the same bf16 operands; the dispatcher names it."""
    from vfd_gan_amd import _lib, functional as F
    for xs, cout, k, p in (((2, 3, 6, 20, 20), 32, (3, 3, 3), (1, 1, 1)), ((2, 3, 4, 24, 24), 14, (1, 3, 3), (0, 1, 1)),
                           ((1, 2, 8, 24, 24), 32, (3, 1, 1), (1, 0, 0))):
        x = _rand(xs, 21).bfloat16().float()
        w = _rand((cout, xs[1]) + k, 22, 0.3).bfloat16().float()
        b = _rand((cout,), 23)
        pre = TF.conv3d(x, w, b, 1, p)
        sums = F.new_stats_buffer(cout, dev)
        xc = F.to_cl(x.to(dev), torch.bfloat16)
        yc = F.conv(xc, torch.nn.Parameter(w.to(dev)), torch.nn.Parameter(b.to(dev)), 1, p, 0, False, _lib.ACT_LRELU, 0.2, stats=sums)
        assert relerr(yc.to_torch(), TF.leaky_relu(pre, 0.2)) < TOL[torch.bfloat16]
        cp = F.cpad(cout)
        folded = sums.view(F.STATS_REPLICAS, 2, cp).sum(0)
        assert relerr(folded[0, :cout], pre.sum(dim=(0, 2, 3, 4))) < 1e-4 and relerr(folded[1, :cout], (pre * pre).sum(dim=(0, 2, 3, 4))) < 1e-4
        desc = F._make_desc(xs[0], tuple(xc.t.shape[1:4]), xs[1], tuple(yc.t.shape[1:4]), cout, k, (1, 1, 1), p, False, torch.bfloat16)
        assert F._conv_kernel_name(desc, sums) == "conv_cin8<bf16>"


@pytest.mark.parametrize("dt", DTYPES, ids=["f32", "bf16"])
def test_sequential_activation_gradient_handover(dt, dev):
    """Conv -> act -> Conv inside one Sequential: the consumer's data gradient carries the producer's activation gradient
    (vfd_conv_forward_mul) and the producer skips its act_backward pass; gradients must equal torch's, through the
    LDS-transposed epilogue (64 channels), the direct one (24 channels) and a transposed consumer."""
    import vfd_gan_amd.nn as vnn
    from vfd_gan_amd import functional as F
    torch.manual_seed(5)
    ref = torch.nn.Sequential(torch.nn.Conv2d(5, 64, 4, 2, 1, bias=False), torch.nn.LeakyReLU(0.2),
                              torch.nn.Conv2d(64, 24, 3, 1, 1), torch.nn.Tanh(),
                              torch.nn.ConvTranspose2d(24, 72, 4, 2, 1, bias=False), torch.nn.Sigmoid(),
                              torch.nn.Conv2d(72, 8, 1, 1, 0))
    mine = vnn.Sequential(vnn.Conv2d(5, 64, 4, 2, 1, bias=False), vnn.LeakyReLU(0.2), vnn.Conv2d(64, 24, 3, 1, 1), vnn.Tanh(),
                          vnn.ConvTranspose2d(24, 72, 4, 2, 1, bias=False), vnn.Sigmoid(), vnn.Conv2d(72, 8, 1, 1, 0))
    if dt == torch.bfloat16:
        with torch.no_grad():
            for prm in ref.parameters():
                prm.copy_(prm.bfloat16().float())
    mine.load_state_dict(ref.state_dict())
    mine.to(dev)
    x = _rand((3, 5, 12, 10), 21)
    if dt == torch.bfloat16:
        x = x.bfloat16().float()
    xr = x.clone().requires_grad_()
    yr = ref(xr)
    gy = _rand(tuple(yr.shape), 22)
    yr.backward(gy)
    xd = x.to(dev).requires_grad_()
    yc = mine(F.to_cl(xd, dt))
    y = yc.to_torch()
    y.backward(gy.to(dev))
    torch.cuda.synchronize()
    tol = TOL[dt] * (2 if dt == torch.bfloat16 else 5)     # three stacked layers
    assert relerr(y, yr) < tol
    assert relerr(xd.grad, xr.grad) < tol, relerr(xd.grad, xr.grad)
    for (n, pm), pr in zip(mine.named_parameters(), ref.parameters()):
        assert relerr(pm.grad, pr.grad) < tol, (n, relerr(pm.grad, pr.grad))


def _count_calls(lib, name, counter):
    orig = getattr(lib, name)

    def wrapped(*a):
        counter[name] = counter.get(name, 0) + 1
        return orig(*a)
    setattr(lib, name, wrapped)
    return orig


@pytest.mark.parametrize("nd", [2, 3], ids=["frames", "volumes"])
def test_sequential_batchnorm_gradient_handover(nd, dev):
    """Conv -> BatchNorm -> act -> Conv inside one Sequential (bf16): the consumer's data gradient stores g = dy*act'(z) and
    the per-channel sums of g and g*xhat (vfd_conv_forward_bn_backward), BatchNorm's backward is the apply pass alone
    (vfd_bn_backward_apply_sums) and its forward folds the epilogue statistics itself (vfd_bn_act_forward_sums).  Checked
    against torch and against the unfused path, through the 64-, 128- and 256-channel tiles (ragged channel counts), a
    stride-2 consumer (class-wise transposed data gradient), a transposed consumer and the halo kernels."""
    import vfd_gan_amd.nn as vnn
    from vfd_gan_amd import _lib, functional as F
    torch.manual_seed(11)
    dt = torch.bfloat16
    if nd == 2:
        T, V = torch.nn, vnn
        spec = lambda M: [M.Conv2d(5, 64, 4, 2, 1, bias=False), M.BatchNorm2d(64), M.LeakyReLU(0.2),      # noqa: E731
                          M.Conv2d(64, 136, 3, 1, 1, bias=True), M.BatchNorm2d(136), M.ReLU(),
                          M.Conv2d(136, 264, 4, 2, 1, bias=False), M.BatchNorm2d(264), M.LeakyReLU(0.2),
                          M.ConvTranspose2d(264, 40, 4, 2, 1, bias=False), M.BatchNorm2d(40),
                          M.Conv2d(40, 8, 3, 1, 1)]
        x = _rand((6, 5, 24, 20), 31)
    else:
        T, V = torch.nn, vnn
        spec = lambda M: [M.Conv3d(40, 64, 3, 1, 1, bias=False), M.BatchNorm3d(64), M.LeakyReLU(0.2),      # noqa: E731
                          M.Conv3d(64, 48, 3, 1, 1, bias=True), M.BatchNorm3d(48), M.ReLU(),
                          M.Conv3d(48, 8, (1, 3, 3), 1, (0, 1, 1))]
        x = _rand((2, 40, 4, 12, 12), 32)
    ref = torch.nn.Sequential(*spec(T))
    with torch.no_grad():
        for m in ref:
            if isinstance(m, (torch.nn.BatchNorm2d, torch.nn.BatchNorm3d)):
                m.weight.copy_(torch.rand_like(m.weight) + 0.5)
                m.bias.copy_(torch.randn_like(m.bias) * 0.3)
        for prm in ref.parameters():
            prm.copy_(prm.bfloat16().float())
    x = x.bfloat16().float()
    xr = x.clone().requires_grad_()
    state0 = {k: v.clone() for k, v in ref.state_dict().items()}      # before ref's forward updates the running statistics
    yr = ref(xr)
    gy = _rand(tuple(yr.shape), 33)
    yr.backward(gy)
    lib = _lib.load()
    prev = lib.vfd_conv_set_halo_mode(2 if nd == 3 else 0)      # volumes: the halo kernels, whatever the tile count
    results = {}
    try:
        for mode in ("fused", "unfused"):
            mine = vnn.Sequential(*spec(V))
            mine.load_state_dict(state0)
            mine.to(dev)
            calls = {}
            names = ("vfd_conv_forward_bn_backward", "vfd_bn_backward_apply_sums", "vfd_bn_act_forward_sums", "vfd_bn_act_backward",
                     "vfd_bn_act_backward_sums")
            origs = {nm: _count_calls(lib, nm, calls) for nm in names}
            old = vnn._NO_HANDOVER
            vnn._NO_HANDOVER = mode == "unfused"
            old_minc = lib.vfd_conv_set_bn_handover_min_channels(33)      # default: off (no gain once the reduce pass overlaps the side stream)
            try:
                xd = x.to(dev).requires_grad_()
                y = mine(F.to_cl(xd, dt)).to_torch()
                y.backward(gy.to(dev))
                torch.cuda.synchronize()
            finally:
                vnn._NO_HANDOVER = old
                lib.vfd_conv_set_bn_handover_min_channels(old_minc)
                for nm, o in origs.items():
                    setattr(lib, nm, o)
            nbn = sum(isinstance(m, (vnn.BatchNorm2d, vnn.BatchNorm3d)) for m in mine)
            assert calls.get("vfd_bn_act_forward_sums", 0) == nbn
            if mode == "fused":
                # every BatchNorm whose consumer's data gradient has more than 32 channels: 64, 136, 264, 40 (frames); 64, 48 (volumes)
                want = 4 if nd == 2 else 2
                assert calls.get("vfd_conv_forward_bn_backward", 0) == want and calls.get("vfd_bn_backward_apply_sums", 0) == want, calls
                assert calls.get("vfd_bn_act_backward_sums", 0) == nbn - want
            else:
                # no hand-over: reduce pass with atomics into the pooled sums + folding apply pass (two launches)
                assert calls.get("vfd_conv_forward_bn_backward", 0) == 0 and calls.get("vfd_bn_act_backward_sums", 0) == nbn
            assert calls.get("vfd_bn_act_backward", 0) == 0
            results[mode] = (y, xd.grad, {n: p.grad.clone() for n, p in mine.named_parameters()},
                             {n: b.clone() for n, b in mine.named_buffers()})
    finally:
        lib.vfd_conv_set_halo_mode(prev)
    tol = TOL[dt] * 2
    report = {}
    for mode, (y, gx, grads, bufs) in results.items():
        report[mode] = {"y": relerr(y, yr), "gx": (relerr(gx, xr.grad), relrms(gx, xr.grad))}
        for (n, pr) in ref.named_parameters():
            report[mode][n] = (relerr(grads[n], pr.grad), relrms(grads[n], pr.grad))
        for (n, br) in ref.named_buffers():
            report[mode][n] = (relerr(bufs[n].float(), br.float()), 0.0)
    print(report)
    # against torch's f32 graph: the forward tightly; the gradients of a stack of up to 4 BatchNorm+ReLU layers in bf16 are
    # dominated by derivative kinks that round to the other side (~6 % rms here, the SAME figure with
    # and without the hand-over), so the gate that pins the hand-over itself is fused-vs-unfused below
    for mode in report:
        assert report[mode]["y"] < tol, report
        for n, v in report[mode].items():
            if n.endswith(("running_mean", "running_var", "num_batches_tracked")):
                assert v[0] < 2e-3, (mode, n, v)
            elif n == "3.bias":
                # bias of a conv that feeds a BatchNorm: its gradient (column sums of the BatchNorm's dx, taken inside the
                # apply pass) is zero up to rounding, in torch as here: bound it against the layer's weight gradient
                gb, gw = results[mode][2]["3.bias"], results[mode][2]["3.weight"]
                assert float(gb.abs().max()) < 2e-2 * float(gw.abs().max()), (mode, float(gb.abs().max()), float(gw.abs().max()))
            elif n != "y":
                assert v[1] < 0.08, (mode, n, v)
                assert abs(v[1] - report["unfused"][n][1]) < 5e-3, (mode, n, v, report["unfused"][n])
    # fused vs unfused: same kernels up to the bf16 rounding of g before the apply pass
    # (same forward kernels; the epilogue statistics are float atomics, so two runs may differ in the last bf16 bit)
    assert relerr(results["fused"][0], results["unfused"][0]) < 1e-2
    assert relerr(results["fused"][1], results["unfused"][1]) < 1.5e-2
    for n in results["fused"][2]:
        if n != "3.bias":
            assert relerr(results["fused"][2][n], results["unfused"][2][n]) < 1.5e-2, n


@pytest.mark.parametrize("dt", DTYPES, ids=["f32", "bf16"])
@pytest.mark.parametrize("shape,act,slope", [((4, 21, 3, 5, 7), 1, 0.2), ((6, 8, 1, 9, 9), 1, 0.0), ((3, 130, 2, 4, 4), 1, 64.0),
                                             ((16, 40), 1, 0.0), ((2, 5, 2, 3, 3), 0, 0.0), ((2, 5, 2, 3, 3), 3, 0.0)])
def test_bn_act(shape, act, slope, dt, dev):
    from vfd_gan_amd import functional as F
    C = shape[1]
    x = _rand(shape, 21, 2.0) + 0.7
    if dt == torch.bfloat16:
        x = x.bfloat16().float()
    gamma, beta = _rand((C,), 22) + 1.5, _rand((C,), 23)
    rm, rv = _rand((C,), 24), _rand((C,), 25).abs() + 0.5
    xr, gr, br = x.clone().requires_grad_(), gamma.clone().requires_grad_(), beta.clone().requires_grad_()
    rmr, rvr = rm.clone(), rv.clone()
    z = TF.batch_norm(xr, rmr, rvr, gr, br, True, 0.1, 1e-5)
    yr = {0: lambda t: t, 1: lambda t: TF.leaky_relu(t, slope), 3: torch.tanh}[act](z)
    gy = _rand(shape, 26)
    yr.backward(gy)
    xd = x.to(dev).requires_grad_()
    gd, bd = torch.nn.Parameter(gamma.to(dev)), torch.nn.Parameter(beta.to(dev))
    rmd, rvd = rm.to(dev), rv.to(dev)
    y = F.bn_act(F.to_cl(xd, dt), gd, bd, rmd, rvd, 1e-5, 0.1, act, slope).to_torch()
    y.backward(gy.to(dev))
    tol = TOL[dt] * (4 if dt == torch.float32 else 1)
    assert relerr(y, yr) < tol
    assert relerr(rmd, rmr) < 1e-5 and relerr(rvd, rvr) < 1e-5
    btol = 5e-2 if dt == torch.bfloat16 else 2e-4
    assert relerr(xd.grad, xr.grad) < btol
    assert relerr(gd.grad, gr.grad) < btol and relerr(bd.grad, br.grad) < btol


@pytest.mark.parametrize("dt", DTYPES, ids=["f32", "bf16"])
def test_pool_upsample_concat_dropout(dt, dev):
    from vfd_gan_amd import functional as F
    tol = TOL[dt]
    x = _rand((2, 11, 4, 6, 8), 31)
    if dt == torch.bfloat16:
        x = x.bfloat16().float()
    for k in [(2, 2, 2), (1, 2, 2), (2, 1, 1), (4, 1, 1), (1, 6, 8)]:
        xr = x.clone().requires_grad_()
        yr = TF.avg_pool3d(xr, k)
        gy = _rand(tuple(yr.shape), 32)
        yr.backward(gy)
        xd = x.to(dev).requires_grad_()
        y = F.avg_pool(F.to_cl(xd, dt), k).to_torch()
        y.backward(gy.to(dev))
        assert relerr(y, yr) < tol and relerr(xd.grad, xr.grad) < tol
    for shp in [(2, 11, 4, 6, 8), (1, 3, 1, 2, 5)]:
        x2 = _rand(shp, 33)
        xr = x2.clone().requires_grad_()
        yr = TF.interpolate(xr, scale_factor=2, mode="trilinear", align_corners=True)
        gy = _rand(tuple(yr.shape), 34)
        yr.backward(gy)
        xd = x2.to(dev).requires_grad_()
        y = F.upsample_trilinear2x(F.to_cl(xd, dt)).to_torch()
        y.backward(gy.to(dev))
        assert relerr(y, yr) < tol * 2 and relerr(xd.grad, xr.grad) < tol * 4
    for ca, cb in [(8, 16), (5, 3), (11, 21)]:
        a, b = _rand((2, ca, 2, 3, 3), 35), _rand((2, cb, 2, 3, 3), 36)
        ar, br_ = a.clone().requires_grad_(), b.clone().requires_grad_()
        yr = torch.cat([ar, br_], 1)
        gy = _rand(tuple(yr.shape), 37)
        yr.backward(gy)
        ad, bd = a.to(dev).requires_grad_(), b.to(dev).requires_grad_()
        y = F.cat_channels(F.to_cl(ad, dt), F.to_cl(bd, dt)).to_torch()
        y.backward(gy.to(dev))
        assert relerr(y, yr) < tol and relerr(ad.grad, ar.grad) < tol and relerr(bd.grad, br_.grad) < tol
    g1 = _rand((2, 1, 3, 4, 4), 38)
    assert relerr(F.gray2rgb(F.to_cl(g1.to(dev), dt)).to_torch(), torch.cat([g1] * 3, 1)) < tol
    # dropout: imposed mask reproduces x*mask/(1-p); generated mask has the right rate and is reused in backward
    xm = _rand((2, 5, 2, 4, 4), 39)
    mask = (torch.rand(xm.shape, generator=torch.Generator().manual_seed(5)) > 0.25)
    F.set_dropout_mask_provider(lambda shape, p, i: mask)
    xd = xm.to(dev).requires_grad_()
    y = F.dropout(F.to_cl(xd, dt), 0.25).to_torch()
    y.backward(torch.ones_like(y))
    F.set_dropout_mask_provider(None)
    assert relerr(y, xm * mask / 0.75) < tol and relerr(xd.grad, mask.float() / 0.75) < tol
    big = torch.ones(4, 8, 4, 16, 16, device=dev)
    yb = F.dropout(F.to_cl(big, dt), 0.25).to_torch()
    rate = float((yb == 0).float().mean())
    assert abs(rate - 0.25) < 0.01


@pytest.mark.parametrize("dt", DTYPES, ids=["f32", "bf16"])
def test_losses(dt, dev):
    from vfd_gan_amd import functional as F
    from vfd_oracle.losses import l2_loss, weighted_bce
    # known-answer tests of SURVEY.md 8(c)
    a = torch.tensor([.2, .9, .5]).view(1, 3, 1, 1)
    b = torch.tensor([0., 1., 1.]).view(1, 3, 1, 1)
    ac, bc = F.to_cl(a.to(dev), torch.float32), F.to_cl(b.to(dev), torch.float32)
    assert abs(float(F.l2_loss(ac, bc)) - 0.1) < 1e-7
    assert abs(float(F.weighted_bce(ac, bc)) - 0.41493162) < 1e-6
    assert abs(float(F.bce_loss(ac, bc)) - 0.34055042) < 1e-6
    tol = 1e-5 if dt == torch.float32 else 1e-2
    shape = (3, 5, 2, 6, 6)
    p = torch.sigmoid(_rand(shape, 41, 3.0))
    t = (_rand(shape, 42) > 0.6).float()
    q = _rand(shape, 43)
    if dt == torch.bfloat16:
        p, q = p.bfloat16().float(), q.bfloat16().float()
    cases = [("l2", F.l2_loss, l2_loss, p, q), ("l1", F.l1_loss, torch.nn.L1Loss(), p, q),
             ("bce", F.bce_loss, torch.nn.BCELoss(), p, t), ("wbce", F.weighted_bce, weighted_bce, p, t)]
    for name, fh, fr, u, v in cases:
        ur, vr = u.clone().requires_grad_(), v.clone().requires_grad_(name in ("l2", "l1"))
        lr = fr(ur, vr) * 3.0
        lr.backward()
        ud = u.to(dev).requires_grad_()
        vd = v.to(dev).requires_grad_(name in ("l2", "l1"))
        lh = fh(F.to_cl(ud, dt), F.to_cl(vd, dt)) * 3.0
        lh.backward()
        assert abs(float(lh.detach()) - float(lr.detach())) < tol * max(1.0, abs(float(lr.detach()))), name
        assert relerr(ud.grad, ur.grad) < max(tol, TOL[dt]), name
        if name in ("l2", "l1"):
            assert relerr(vd.grad, vr.grad) < max(tol, TOL[dt]), name
    # constant-label BCE (ones / zeros labels of the GAN step)
    pr = torch.sigmoid(_rand((7, 1), 44, 2.0))
    for lab in (1.0, 0.0):
        prr = pr.clone().requires_grad_()
        lr = torch.nn.BCELoss()(prr.view(-1), torch.full((7,), lab))
        lr.backward()
        pd = pr.to(dev).requires_grad_()
        lh = F.bce_loss(F.to_cl(pd, torch.float32), lab)
        lh.backward()
        assert abs(float(lh) - float(lr)) < 1e-6 and relerr(pd.grad, prr.grad) < 1e-5


def test_zero_fill_and_loss_term_arithmetic(dev):
    """vfd_zero (any byte count, 16-byte-aligned base) and vfd_weighted_sum4 / vfd_scale4: the torch expression they replace,
    `a * w0 + b * w1 + c * w2` (float32, left to right, no fused multiply-add), forward and backward, bit for bit."""
    from vfd_gan_amd import functional as F
    for n in (1, 3, 4, 5, 1000, 4099):
        t = torch.full((n,), 7.0, device=dev)
        F.zero_(t)
        assert float(t.abs().sum()) == 0.0
        b = torch.full((n,), 3, dtype=torch.uint8, device=dev)
        F.zero_(b)
        assert int(b.sum()) == 0
    g = torch.Generator().manual_seed(8)
    for k in (1, 2, 3, 4):
        vals = [torch.rand((), generator=g).mul(3).to(dev).requires_grad_() for _ in range(k)]
        ws = [0.5, 50.0, 1.0, 0.37][:k]
        out = F.weighted_sum(*zip(vals, ws))
        ref_in = [v.detach().clone().requires_grad_() for v in vals]
        ref = ref_in[0] * ws[0]
        for v, w in zip(ref_in[1:], ws[1:]):
            ref = ref + v * w
        assert float(out.detach()) == float(ref.detach()), (k, float(out.detach()), float(ref.detach()))
        up = torch.tensor(1.75, device=dev)
        out.backward(up)
        ref.backward(up)
        for a, r in zip(vals, ref_in):
            assert float(a.grad) == float(r.grad)


def test_adam_matches_torch(dev):
    from vfd_gan_amd import optim as hoptim
    torch.manual_seed(0)
    ps = [torch.randn(33, 7), torch.randn(5), torch.randn(4, 3, 3, 3)]
    ref = [torch.nn.Parameter(p.clone()) for p in ps]
    mine = [torch.nn.Parameter(p.clone().to(dev)) for p in ps]
    o_ref = torch.optim.Adam(ref, lr=2e-4, betas=(0.5, 0.999))
    o_hip = hoptim.Adam(mine, lr=2e-4, betas=(0.5, 0.999))
    for it in range(4):
        gs = [torch.randn_like(p) * (0.1 + it) for p in ps]
        o_ref.zero_grad()
        o_hip.zero_grad()
        for p, g in zip(ref, gs):
            p.grad = g.clone()
        for p, g in zip(mine, gs):
            p.grad.copy_(g.to(dev))
        o_ref.step()
        o_hip.step()
    for a, b in zip(mine, ref):
        assert relerr(a, b) < 1e-6


def test_bn_running_update_repeats_the_forward_side_effect(dev):
    """vfd_bn_running_update(mean, rstd) == the running-statistics update of one more training-mode forward on the same batch
    (torch.nn.BatchNorm2d run twice on identical input): what ganomaly's backward_g applies instead of repeating netd(x)."""
    import vfd_gan_amd.nn as vnn
    from vfd_gan_amd import functional as F
    torch.manual_seed(3)
    x = _rand((6, 20, 9, 7), 41) * 2.0 + 0.7
    ref = torch.nn.BatchNorm2d(20)
    ref.train()
    ref(x)
    ref(x)
    mine = vnn.BatchNorm2d(20).to(dev)
    mine.train()
    mine._keep_batch_stats = True
    mine(F.to_cl(x.to(dev), torch.float32))
    mine.repeat_running_update()
    torch.cuda.synchronize()
    assert relerr(mine.running_mean, ref.running_mean) < 1e-6
    assert relerr(mine.running_var, ref.running_var) < 1e-6
    assert int(mine.num_batches_tracked.item()) == 2


def test_pack_filters_batched_equals_single(dev):
    """vfd_pack_filters (every filter copy of an optimiser in one launch, device job table) == vfd_pack_filter per filter:
    checked through the optimiser hook (functional.repack_owned) on filters of assorted shapes, both orientations."""
    import vfd_gan_amd.nn as vnn
    from vfd_gan_amd import functional as F, optim
    torch.manual_seed(9)
    net = vnn.Sequential(vnn.Conv2d(5, 72, 4, 2, 1, bias=False), vnn.LeakyReLU(0.2), vnn.Conv3d(72, 40, (1, 3, 3), 1, (0, 1, 1)),
                         vnn.LeakyReLU(0.2), vnn.ConvTranspose2d(40, 9, 4, 2, 1, bias=False)).to(dev)
    opt = optim.Adam(net.parameters(), lr=1e-2)
    x = F.to_cl(_rand((2, 5, 12, 12), 5).to(dev).requires_grad_(), torch.bfloat16)

    def step():
        opt.zero_grad()
        y = x
        for m in net:
            y = m(y) if not isinstance(m, vnn.Conv3d) else F.ClTensor(m(F.ClTensor(y.t, y.C, 3)).t, 40, 2)
        y.to_torch().float().pow(2).mean().backward()
        opt.step()
    step()          # first step: copies packed lazily, registered, table built
    step()          # second step: all copies re-packed by the batched launch after the update
    torch.cuda.synchronize()
    n = 0
    for prm in net.parameters():
        for key, ent in prm.__dict__.get("_vfd_packed", {}).items():
            if key[0] == "fp8":
                continue
            dt, tr = key
            batched = ent[1].clone()
            A, B = prm.shape[0], prm.shape[1]
            T = prm[0, 0].numel()
            single = torch.empty_like(batched)
            _lib_check(F.load().vfd_pack_filter(F.dtype_code(dt), prm.detach().contiguous().data_ptr(), single.data_ptr(), A, B, T, int(tr),
                                                F.stream()))
            torch.cuda.synchronize()
            assert torch.equal(batched, single), (tuple(prm.shape), key)
            n += 1
    assert n >= 5        # forward copies of 3 filters + data-gradient copies of the last 2


def _lib_check(rc):
    from vfd_gan_amd._lib import check
    check(rc, "pack_filter")


@pytest.mark.parametrize("pools", [((2, 2, 2), (2, 2, 2)), ((1, 2, 2), (2, 1, 1))], ids=["pool222", "pool122_211"])
@pytest.mark.parametrize("dt", DTYPES, ids=["f32", "bf16"])
def test_batchnorm_activation_avgpool_fused(dt, pools, dev):
    """Conv3d -> BatchNorm3d -> LeakyReLU -> AvgPool3d(2) -> Conv3d (anogan NetD's block, models/anogan.py:84-105): the
    normalise + activate + pool pass writes only the pooled tensor and BatchNorm's backward takes the pooled gradient
    (vfd_bn_act_pool_forward_sums / vfd_bn_act_pool_backward_sums), against torch and against the unfused path; f32 with the
    epilogue statistics switched on pins the arithmetic, bf16 is the production path.  Pools (2,2,2) (anogan) and (1,2,2) /
    (2,1,1) (mygan's SDisc / TDisc).  Conv biases included (their gradient
    comes out of the fused apply pass)."""
    import vfd_gan_amd.nn as vnn
    from vfd_gan_amd import _lib, functional as F
    torch.manual_seed(21)
    spec = lambda M: [M.Conv3d(6, 40, 3, 1, 1), M.BatchNorm3d(40), M.LeakyReLU(0.2), M.AvgPool3d(pools[0]),      # noqa: E731
                      M.Conv3d(40, 72, 3, 1, 1), M.BatchNorm3d(72), M.LeakyReLU(64.0), M.AvgPool3d(pools[1]), M.Conv3d(72, 8, 1, 1, 0)]
    ref = torch.nn.Sequential(*spec(torch.nn))
    with torch.no_grad():
        for m in ref:
            if isinstance(m, torch.nn.BatchNorm3d):
                m.weight.copy_(torch.rand_like(m.weight) + 0.5)
                m.bias.copy_(torch.randn_like(m.bias) * 0.3)
        if dt == torch.bfloat16:
            for prm in ref.parameters():
                prm.copy_(prm.bfloat16().float())
    x = _rand((2, 6, 4, 12, 8), 61)
    if dt == torch.bfloat16:
        x = x.bfloat16().float()
    state0 = {k: v.clone() for k, v in ref.state_dict().items()}
    xr = x.clone().requires_grad_()
    yr = ref(xr)
    gy = _rand(tuple(yr.shape), 62)
    yr.backward(gy)
    lib = _lib.load()
    vnn.set_epilogue_stats("on")
    results = {}
    try:
        for mode in ("fused", "unfused"):
            mine = vnn.Sequential(*spec(vnn))
            mine.load_state_dict(state0)
            mine.to(dev)
            calls = {}
            origs = {nm: _count_calls(lib, nm, calls) for nm in ("vfd_bn_act_pool_forward_sums", "vfd_bn_act_pool_backward_sums", "vfd_avgpool_forward")}
            old = vnn._NO_HANDOVER
            vnn._NO_HANDOVER = mode == "unfused"
            try:
                xd = x.to(dev).requires_grad_()
                y = mine(F.to_cl(xd, dt)).to_torch()
                y.backward(gy.to(dev))
                torch.cuda.synchronize()
            finally:
                vnn._NO_HANDOVER = old
                for nm, o in origs.items():
                    setattr(lib, nm, o)
            if mode == "fused":
                assert calls.get("vfd_bn_act_pool_forward_sums", 0) == 2 and calls.get("vfd_bn_act_pool_backward_sums", 0) == 2, calls
                assert calls.get("vfd_avgpool_forward", 0) == 0
            else:
                assert calls.get("vfd_bn_act_pool_forward_sums", 0) == 0 and calls.get("vfd_avgpool_forward", 0) == 2
            results[mode] = (y, xd.grad, {n: p.grad.clone() for n, p in mine.named_parameters()}, {n: b.clone() for n, b in mine.named_buffers()})
    finally:
        vnn.set_epilogue_stats("auto")
    tol = TOL[dt] * (10 if dt == torch.float32 else 2)
    for mode, (y, gx, grads, bufs) in results.items():
        assert relerr(y, yr) < tol, (mode, relerr(y, yr))
        assert relerr(gx, xr.grad) < (tol if dt == torch.float32 else 0.1), (mode, relerr(gx, xr.grad))
        for n, pr in ref.named_parameters():
            if n in ("0.bias", "4.bias"):       # bias in front of a BatchNorm: zero up to rounding, in torch as here
                assert float(grads[n].abs().max()) < 2e-2 * float(grads[n.replace("bias", "weight")].abs().max()), (mode, n)
                continue
            e = relrms(grads[n], pr.grad)
            assert e < (tol if dt == torch.float32 else 0.08), (mode, n, e)
        for n, br in ref.named_buffers():
            assert relerr(bufs[n].float(), br.float()) < 2e-3, (mode, n)
    assert relerr(results["fused"][0], results["unfused"][0]) < (1e-5 if dt == torch.float32 else 1e-2)
    # (bf16, max-norm: single elements whose pre-activation rounds to the other side of 0 in one of the two paths change by
    # the derivative jump, 63x for LeakyReLU(64): the rms is the meaningful figure)
    if dt == torch.float32:
        assert relerr(results["fused"][1], results["unfused"][1]) < 1e-4
    else:
        assert relrms(results["fused"][1], results["unfused"][1]) < 1e-1      # measured 6.8e-2 (the slope-64 layer)


@pytest.mark.parametrize("dt", DTYPES, ids=["f32", "bf16"])
def test_batchnorm_activation_avgpool_fused_with_skip(dt, dev):
    """The U-Net encoder form (models/mygannet.py:74-94): the activation feeds the AvgPool AND a full-resolution consumer.  One
    BatchNorm pass writes both tensors; its backward adds the two gradients that come back (no pooling-backward pass, no
    gradient-sum pass).  Against torch, f32 (epilogue statistics on) tightly, bf16 in rms."""
    import vfd_gan_amd.nn as vnn
    from vfd_gan_amd import functional as F
    torch.manual_seed(31)
    T = torch.nn
    r_c1, r_bn, r_ca, r_cb = T.Conv3d(5, 24, 3, 1, 1), T.BatchNorm3d(24), T.Conv3d(24, 16, 3, 1, 1), T.Conv3d(24, 8, (1, 3, 3), 1, (0, 1, 1))
    with torch.no_grad():
        r_bn.weight.copy_(torch.rand(24) + 0.5)
        r_bn.bias.copy_(torch.randn(24) * 0.2)
        if dt == torch.bfloat16:
            for m in (r_c1, r_bn, r_ca, r_cb):
                for prm in m.parameters():
                    prm.copy_(prm.bfloat16().float())
    x = _rand((2, 5, 4, 8, 12), 71)
    if dt == torch.bfloat16:
        x = x.bfloat16().float()
    state = [{k: v.clone() for k, v in m.state_dict().items()} for m in (r_c1, r_bn, r_ca, r_cb)]
    xr = x.clone().requires_grad_()
    full_r = torch.nn.functional.leaky_relu(r_bn(r_c1(xr)), 0.2)
    ya_r, yb_r = r_ca(torch.nn.functional.avg_pool3d(full_r, 2)), r_cb(full_r)
    ga, gb = _rand(tuple(ya_r.shape), 72), _rand(tuple(yb_r.shape), 73)
    ((ya_r * ga).sum() + (yb_r * gb).sum()).backward()
    V = vnn
    c1, bn, ca, cb = V.Conv3d(5, 24, 3, 1, 1), V.BatchNorm3d(24), V.Conv3d(24, 16, 3, 1, 1), V.Conv3d(24, 8, (1, 3, 3), 1, (0, 1, 1))
    for m, st in zip((c1, bn, ca, cb), state):
        m.load_state_dict(st)
        m.to(dev)
    xd = x.to(dev).requires_grad_()
    k = F.stats_buffer_numel(24)
    buf, fstats = torch.zeros(2 * k, dtype=torch.float32, device=dev), F.new_stats_buffer(24, dev)      # forward statistics: float64
    tok = {"taken": False, "rep": buf[k:]}
    h = c1(F.to_cl(xd, dt), stats=fstats, bias_token=tok)
    pooled, full = bn.forward_pooled(h, _lib_act_lrelu(), 0.2, fstats, buf[:k], c1.bias, tok, (2, 2, 2), True)
    ya, yb = ca(pooled).to_torch(), cb(full).to_torch()
    ((ya.float() * ga.to(dev)).sum() + (yb.float() * gb.to(dev)).sum()).backward()
    torch.cuda.synchronize()
    tol = TOL[dt] * (10 if dt == torch.float32 else 2)
    assert relerr(ya, ya_r) < tol and relerr(yb, yb_r) < tol
    assert relerr(full.to_torch(), full_r) < tol
    chk = relerr if dt == torch.float32 else relrms
    assert chk(xd.grad, xr.grad) < (tol if dt == torch.float32 else 0.05), chk(xd.grad, xr.grad)
    for mine, ref in ((c1, r_c1), (bn, r_bn), (ca, r_ca), (cb, r_cb)):
        for (n, pm), pr in zip(mine.named_parameters(), ref.parameters()):
            if mine is c1 and n == "bias":
                continue      # zero up to rounding (feeds a BatchNorm)
            assert chk(pm.grad, pr.grad) < (tol if dt == torch.float32 else 0.05), (type(mine).__name__, n, chk(pm.grad, pr.grad))
    assert relerr(bn.running_var, r_bn.running_var) < 2e-3


def _lib_act_lrelu():
    from vfd_gan_amd import _lib
    return _lib.ACT_LRELU


@pytest.mark.parametrize("dt", DTYPES, ids=["f32", "bf16"])
def test_upsample_cat_fused(dt, dev):
    """F.upsample_cat == cat([Upsample(scale 2, trilinear, align_corners=True)(x), skip], dim=1), forward and both gradients
    (the decoder joint of models/mygannet.py:78-94 written in one pass); ragged skip channel count."""
    from vfd_gan_amd import functional as F
    x = _rand((2, 16, 2, 3, 5), 81)
    skip = _rand((2, 13, 4, 6, 10), 82)
    if dt == torch.bfloat16:
        x, skip = x.bfloat16().float(), skip.bfloat16().float()
    xr, sr = x.clone().requires_grad_(), skip.clone().requires_grad_()
    yr = torch.cat([torch.nn.functional.interpolate(xr, scale_factor=2, mode="trilinear", align_corners=True), sr], dim=1)
    gy = _rand(tuple(yr.shape), 83)
    yr.backward(gy)
    xd, sd = x.to(dev).requires_grad_(), skip.to(dev).requires_grad_()
    yc = F.upsample_cat(F.to_cl(xd, dt), F.to_cl(sd, dt))
    assert yc.C == 29 and float(yc.t[..., yc.C:].float().abs().sum()) == 0.0
    y = yc.to_torch()
    y.backward(gy.to(dev))
    torch.cuda.synchronize()
    tol = TOL[dt]
    assert relerr(y, yr) < tol
    assert relerr(xd.grad, xr.grad) < tol * (1 if dt == torch.float32 else 2), relerr(xd.grad, xr.grad)
    assert relerr(sd.grad, sr.grad) < tol


def test_conv_to_conv_bias_gradient_from_data_gradient_statistics(dev):
    """conv(bias) -> conv with nothing in between (anogan NetG's ConvTranspose3d -> Conv3d pairs, models/anogan.py:51-52; NetD's
    Conv3d -> Conv3d, :85-86): the first layer's bias gradient is the column sum of the second layer's data gradient, taken
    as that launch's epilogue statistics and folded by the first layer's wgrad_reduce launch — no vfd_bias_grad pass.  bf16,
    against torch (these bias gradients are real, not rounding noise)."""
    import vfd_gan_amd.nn as vnn
    from vfd_gan_amd import _lib, functional as F
    torch.manual_seed(41)
    dt = torch.bfloat16
    spec = lambda M: [M.ConvTranspose3d(40, 24, 3, 1, 1), M.Conv3d(24, 48, 3, 1, 1), M.Conv3d(48, 40, (1, 3, 3), 1, (0, 1, 1)),      # noqa: E731
                      M.BatchNorm3d(40), M.LeakyReLU(0.2), M.Conv3d(40, 8, 1, 1, 0)]
    ref = torch.nn.Sequential(*spec(torch.nn))
    with torch.no_grad():
        for prm in ref.parameters():
            prm.copy_(prm.bfloat16().float())
    x = _rand((2, 40, 4, 10, 12), 91).bfloat16().float()
    state0 = {k: v.clone() for k, v in ref.state_dict().items()}
    xr = x.clone().requires_grad_()
    yr = ref(xr)
    gy = _rand(tuple(yr.shape), 92)
    yr.backward(gy)
    from vfd_gan_amd import optim
    mine = vnn.Sequential(*spec(vnn))
    mine.load_state_dict(state0)
    mine.to(dev)
    opt = optim.Adam(mine.parameters(), lr=1e-3)      # arena gradients: the fused path accumulates straight into them
    opt.zero_grad()
    lib = _lib.load()
    calls = {}
    origs = {nm: _count_calls(lib, nm, calls) for nm in ("vfd_bias_grad", "vfd_wgrad_reduce_bias")}
    try:
        xd = x.to(dev).requires_grad_()
        y = mine(F.to_cl(xd, dt)).to_torch()
        y.backward(gy.to(dev))
        F.join_side_stream()
        torch.cuda.synchronize()
    finally:
        for nm, o in origs.items():
            setattr(lib, nm, o)
    # biases of layers 0 and 1 come from their consumers' data-gradient statistics, layer 2's from the BatchNorm apply pass;
    # only the last conv's bias needs the column-sum pass
    assert calls.get("vfd_wgrad_reduce_bias", 0) == 3 and calls.get("vfd_bias_grad", 0) == 1, calls
    tol = TOL[dt] * 2
    assert relerr(y, yr) < tol
    for (n, pm), pr in zip(mine.named_parameters(), ref.parameters()):
        if n == "2.bias":
            continue      # feeds a BatchNorm: zero up to rounding
        e = relrms(pm.grad, pr.grad)
        assert e < 0.06, (n, e)
    assert relerr(mine[0].bias.grad, ref[0].bias.grad) < tol and relerr(mine[1].bias.grad, ref[1].bias.grad) < tol


@pytest.mark.parametrize("dt,ratio", [(torch.float32, 100.0), (torch.float32, 1000.0), (torch.bfloat16, 30.0), (torch.bfloat16, 300.0)],
                         ids=["f32_100", "f32_1000", "bf16_30", "bf16_300"])
def test_batchnorm_epilogue_statistics_large_mean_over_sigma(dt, ratio, dev):
    """|mean| / sigma >> 1 on the conv-epilogue-sums path: conv(bias = ratio * sigma) ->
    BatchNorm -> LeakyReLU with the statistics summed in the conv epilogue.  The variance is E[x^2] - mean^2: float32 sums
    lose ratio^2 digits there (mygan's SDisc on the sparse 0/1 mask sits at ~30 and was off by 5e-3 in its loss; with float32
    atomics all the way, rounds 1-2, the error was ~1e-6 ratio^2).  The epilogues therefore sum SHIFTED values (t - c, c = the
    channel's value in the tile's first row) in float32 and form the raw sums per workgroup in double (conv_epilogue.hpp): the
    variance no longer depends on the ratio — 1e3 (VERDICT r02's request) passes the same 2e-4 gate as 100.  Checked against float64 BatchNorm of
    the stored conv output: running variance, forward, and the backward's re-associated apply pass (dx = g A + x B + D).
    (bf16 storage itself resolves a tensor only to 2^-9 |mean|: a ratio of 30 leaves sigma / 17 of rounding noise in x.)"""
    import vfd_gan_amd.nn as vnn
    from vfd_gan_amd import functional as F
    torch.manual_seed(3)
    N, Ci, Co, sp = 4, 8, 16, (4, 12, 12)
    x = _rand((N, Ci) + sp, 41)
    w = _rand((Co, Ci, 1, 1, 1), 42, 0.3)
    sigma = float(TF.conv3d(x, w).std())
    b = torch.full((Co,), ratio * sigma) * (1 + 0.1 * _rand((Co,), 43))
    if dt == torch.bfloat16:
        x, w = x.bfloat16().float(), w.bfloat16().float()
    conv, bn, act = vnn.Conv3d(Ci, Co, 1, 1, 0), vnn.BatchNorm3d(Co), vnn.LeakyReLU(0.2)
    with torch.no_grad():
        conv.weight.copy_(w)
        conv.bias.copy_(b)
        bn.weight.copy_(1 + 0.2 * _rand((Co,), 44))
        bn.bias.copy_(_rand((Co,), 45))
    seq = vnn.Sequential(conv, bn, act).to(dev).train()
    vnn.set_epilogue_stats("on")
    try:
        xd = x.to(dev).requires_grad_()
        yc = seq(F.to_cl(xd, dt))
        y = yc.to_torch()
        gy = _rand(tuple(y.shape), 46)
        y.backward(gy.to(dev))
        torch.cuda.synchronize()
    finally:
        vnn.set_epilogue_stats("auto")
    # float64 reference on the conv output AS STORED (float32 / bf16 rounding of the conv output is not what is tested)
    t = TF.conv3d(x.double(), w.double(), b.double())
    ts = (t.float() if dt == torch.float32 else t.float().bfloat16().float()).double().requires_grad_()
    mean, var = t.mean((0, 2, 3, 4)), t.var((0, 2, 3, 4), unbiased=False)       # statistics of the unrounded accumulators
    sh = (1, -1, 1, 1, 1)
    xh = (ts - mean.view(sh)) / torch.sqrt(var.view(sh) + bn.eps)
    yr = TF.leaky_relu(xh * bn.weight.detach().cpu().double().view(sh) + bn.bias.detach().cpu().double().view(sh), 0.2)
    n = t.numel() // Co
    assert relerr(bn.running_var.cpu(), (0.9 + 0.1 * var * n / (n - 1)).float()) < 2e-4      # 1 - momentum, momentum * unbiased variance
    assert relerr(bn.running_mean.cpu(), (0.1 * mean).float()) < 1e-5
    tol = 2e-4 if dt == torch.float32 else 4e-2
    assert relerr(y, yr.float()) < tol, relerr(y, yr.float())
    # backward through the BatchNorm (statistics as functions of the data: the standard formula on the stored tensor)
    g = gy.double() * torch.where(yr > 0, 1.0, 0.2)
    gam = bn.weight.detach().cpu().double().view(sh)
    rstd = 1.0 / torch.sqrt(var.view(sh) + bn.eps)
    dxh = g * gam
    dt_ref = rstd * (dxh - dxh.mean((0, 2, 3, 4), keepdim=True) - xh.detach() * (dxh * xh.detach()).mean((0, 2, 3, 4), keepdim=True))
    dx_ref = TF.conv_transpose3d(dt_ref, w.double())          # back through the 1x1x1 conv
    assert relrms(xd.grad, dx_ref.float()) < (1e-3 if dt == torch.float32 else 5e-2), relrms(xd.grad, dx_ref.float())
    assert relerr(bn.weight.grad.cpu(), (g * xh.detach()).sum((0, 2, 3, 4)).float()) < (1e-3 if dt == torch.float32 else 5e-2)
    assert relerr(bn.bias.grad.cpu(), g.sum((0, 2, 3, 4)).float()) < (1e-4 if dt == torch.float32 else 2e-2)
