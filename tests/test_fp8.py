"""fp8 (OCP e4m3fn) operand path (BASELINE configs[4]; include/vfdgan_hip.h "fp8 operands"): the quantiser against torch's
own float8_e4m3fn cast (an independent implementation of the same format) bit for bit, and the v_mfma_f32_16x16x128_f8f6f4
convolution against a float64 convolution of the DEQUANTISED operands (products of two e4m3 numbers are exact in f32, so the
only differences are the accumulation order and the bf16 rounding of the output)."""
import pytest
import torch
import torch.nn.functional as TF

from util import relerr

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    return torch.device("cuda", 0)


def _rand(shape, seed):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(shape, generator=g)


def _to_cl_bf16(x, dev, F):
    """(N,C,*sp) float -> channels-last bf16 block [N,D,H,W,CPAD(C)] on the device."""
    N, C = x.shape[:2]
    sp = tuple(x.shape[2:])
    sp3 = (1,) * (3 - len(sp)) + sp
    cl = torch.zeros((N,) + sp3 + (F.cpad(C),), dtype=torch.bfloat16)
    cl[..., :C] = x.reshape((N, C) + sp3).permute(0, 2, 3, 4, 1).bfloat16()
    return cl.to(dev)


@pytest.mark.parametrize("shape,C", [((3, 5, 7, 40), 37), ((4, 9, 9, 64), 64), ((2, 3, 3, 24), 20)])
def test_quantize_fp8_matches_torch_e4m3(shape, C, dev):
    from vfd_gan_amd import functional as F
    x = (_rand(shape, 3) * 3.0).bfloat16()
    x[..., C:] = 0
    q, scale = F.quantize_fp8(x.to(dev), C)
    torch.cuda.synchronize()
    sc = float(scale.item())
    amax = float(x.float().abs().max())
    assert abs(sc - 448.0 / amax) <= 1e-6 * sc
    want = (x.float() * scale.cpu()).to(torch.float8_e4m3fn).view(torch.uint8)
    got = q.cpu()
    assert got.shape[-1] == F.cpad16(C)
    assert torch.equal(got[..., :x.shape[-1]], want)
    assert int(got[..., x.shape[-1]:].abs().sum()) == 0


CASES = [
    # name, N, Cin, Cout, sp, k, s, p, transposed, act, stats
    ("conv2d_k3_c64_160", 3, 64, 160, (20, 24), 3, 1, 1, False, 0, False),
    ("conv2d_k4s2_c128_96_stats_lrelu", 4, 128, 96, (28, 28), 4, 2, 1, False, 1, True),
    ("convT2d_k4s2_c80_64", 2, 80, 64, (14, 10), 4, 2, 1, True, 0, False),
    ("conv3d_k3_c48_264", 1, 48, 264, (4, 12, 12), 3, 1, 1, False, 0, True),
    ("conv2d_k4s2_c512_1024", 2, 512, 1024, (14, 14), 4, 2, 1, False, 0, False),
]


@pytest.mark.parametrize("case", CASES, ids=[c[0] for c in CASES])
def test_conv_fp8_against_dequantised_reference(case, dev):
    from vfd_gan_amd import _lib, functional as F
    name, N, Cin, Cout, sp, k, s, p, tr, act, want_stats = case
    nd = len(sp)
    x = _rand((N, Cin) + sp, 11)
    w = _rand((Cin, Cout) + (k,) * nd if tr else (Cout, Cin) + (k,) * nd, 12) * 0.05
    bias = _rand((Cout,), 13) * 0.1
    xc = _to_cl_bf16(x, dev, F)
    xq, xs = F.quantize_fp8(xc, Cin)
    T = k ** nd
    A, B = (Cin, Cout) if tr else (Cout, Cin)
    wq, ws = F.pack_filter_fp8(w.to(dev), transpose_ab=tr, A=A, B=B, T=T)
    sp3 = (1,) * (3 - nd) + sp
    k3, s3, p3 = (1,) * (3 - nd) + (k,) * nd, (1,) * (3 - nd) + (s,) * nd, (0,) * (3 - nd) + (p,) * nd
    if tr:
        out3 = tuple((sp3[i] - 1) * s3[i] - 2 * p3[i] + k3[i] for i in range(3))
    else:
        out3 = tuple((sp3[i] + 2 * p3[i] - k3[i]) // s3[i] + 1 for i in range(3))
    stats = F.new_stats_buffer(Cout, dev) if want_stats else None
    y = F.conv_fp8(xq, xs, wq, ws, bias.to(dev), N, sp3, Cin, out3, Cout, k3, s3, p3, tr, act=act, slope=0.2, stats=stats)
    torch.cuda.synchronize()
    # reference on the dequantised operands (what the kernel multiplies), float64
    xs_, ws_ = float(xs.item()), float(ws.item())
    xdq = xq.cpu().view(torch.float8_e4m3fn).double()[..., :Cin] / xs_            # [N,D,H,W,Cin]
    xdq = xdq.permute(0, 4, 1, 2, 3).reshape((N, Cin) + sp)
    wdq = wq.cpu().view(torch.float8_e4m3fn).double() / ws_                       # packed [R][T][Cc16]
    R, Cc = (B, A) if tr else (A, B)
    wdq = wdq[..., :Cc].reshape((R,) + (k,) * nd + (Cc,))
    # packed[r][t][c] = w[c][r][t] (transposed: r = Cout, c = Cin)  |  w[r][c][t] (regular: r = Cout, c = Cin)
    wref = wdq.permute((nd + 1, 0) + tuple(range(1, nd + 1))) if tr else wdq.permute((0, nd + 1) + tuple(range(1, nd + 1)))
    conv = {2: (TF.conv2d, TF.conv_transpose2d), 3: (TF.conv3d, TF.conv_transpose3d)}[nd][1 if tr else 0]
    pre = conv(xdq, wref.contiguous(), bias.double(), stride=s, padding=p)
    ref = TF.leaky_relu(pre, 0.2) if act == 1 else pre
    got = y[..., :Cout].float().cpu().permute(0, 4, 1, 2, 3).reshape(ref.shape)
    assert relerr(got, ref) < 6e-3, relerr(got, ref)            # bf16 output rounding: 2^-8 = 3.9e-3 of max |y|
    assert float(y[..., Cout:].float().abs().sum()) == 0.0
    if want_stats:
        folded = stats.view(F.STATS_REPLICAS, 2, F.cpad(Cout)).sum(0).cpu().double()
        dims = (0,) + tuple(range(2, 2 + nd))
        assert relerr(folded[0, :Cout], pre.sum(dim=dims)) < 2e-3 and relerr(folded[1, :Cout], (pre * pre).sum(dim=dims)) < 2e-3
    desc = F._make_desc(N, sp3, Cin, out3, Cout, k3, s3, p3, tr, torch.bfloat16)
    desc.dtype = _lib.FP8
    assert F._conv_kernel_name(desc) == ("conv_igemm<fp8,256c_x_256p>" if Cout > 128 else "conv_igemm<fp8,128c_x_128p>")
