"""bf16-FAITHFUL mode of the oracle (TEST INFRASTRUCTURE, like the rest of oracle/).

The float32 oracle pins the arithmetic of the reference (models/{ganomaly,anogan,mygannet,spatiotempconv}.py); the
benchmarked HIP path stores activations, packed filters and activation gradients in bfloat16 (float32 accumulation,
float32 BatchNorm statistics, float32 master weights / Adam).  Against the float32 oracle that path can only be gated
at ~5e-2: a bf16 forward deviates by ~1e-2, about 1 % of the activations land on the other side of a ReLU / LeakyReLU /
L1 kink, and each such flip is an O(1) relative error of one gradient element in ANY bf16 implementation.

This module re-runs the SAME oracle modules (same parameters, same step functions) with a round-to-bfloat16 placed at
exactly the tensors the HIP path stores — and nowhere else:

* every filter is used as ``w + (bf16(w) - w).detach()`` (the packed bf16 copy; the gradient reaches the float32 master
  unrounded, as vfd_wgrad_reduce writes it),
* a conv / linear output is rounded once, AFTER the epilogue that is fused into the kernel (bias, activation),
* BatchNorm -> activation (-> AvgPool3d where vfd_gan_amd.nn.run_fused / models.mygannet._conv_bn_act fuse it) is rounded
  once, after the last fused stage (the U-Net encoder's ``keep_full`` form: pooled and full-resolution outputs rounded
  separately, the pool taken from the unrounded activation),
* the batch statistics of a conv -> BatchNorm pair are those of the conv's float32 accumulators (the HIP conv epilogue sums
  them BEFORE the output is rounded; `bn_from`), applied to the rounded tensor.  Measured: with the statistics of the
  rounded tensor instead, ~1.5 % of a layer's outputs land on the other side of a bf16 rounding boundary (the means differ by
  the average rounding error, ~5e-6 sigma), the next layers amplify that, and four stages on a third of all elements differ,
* gradients are rounded at the same tensors on the way back (``_Round.backward``), plus the activation gradient
  ``g = dy * act'(y)`` of a conv+activation kernel, which the HIP path stores before the data / filter gradient use it.

Kink decisions then happen on (almost always) identical values on both sides, and the whole-step bf16 gates of the GPU
suite drop from 5e-2 to a few 1e-3 (what remains: float32 summation order can move a value across a bf16 rounding
boundary — a 2^-8 relative change of ~0.03 % of the elements — and the few places listed in `KNOWN_MISMATCHES`).
The fusion plan below restates vfd_gan_amd/nn.py:run_fused and the model files; a fusion added there must be added here,
or the tight gates fail (that is the point).
"""
import torch
import torch.nn as nn
import torch.nn.functional as TF

from . import anogan as OA
from . import ganomaly as OG
from . import mygannet as OM
from .spatiotempconv import SpatioTemporalConv

KNOWN_MISMATCHES = (
    "float32 summation order (MFMA K order, atomics) differs from torch's: a value within ~1e-6 relative of a bf16 rounding "
    "boundary may round the other way (~0.001 % of a conv's outputs)",
)


def rbf(t):
    """Round a float32 tensor to the nearest bfloat16 (ties to even, like v_cvt_pk_bf16_f32), result in float32."""
    return t.to(torch.bfloat16).to(torch.float32)


class _Round(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, fwd, bwd):
        ctx.bwd = bwd
        return rbf(x) if fwd else x.clone()

    @staticmethod
    def backward(ctx, g):
        return (rbf(g) if ctx.bwd else g), None, None


def R(x):
    """A tensor the HIP path stores in bf16 in both directions (value forward, its gradient backward)."""
    return _Round.apply(x, True, True)


def RF(x):
    """Stored forward only (its gradient is consumed inside the kernel that produces it)."""
    return _Round.apply(x, True, False)


def RB(x):
    """Stored backward only (the gradient w.r.t. a value that never leaves the registers forward)."""
    return _Round.apply(x, False, True)


class _ActFromOutput(torch.autograd.Function):
    """y = bf16(act(t)) as a conv + activation kernel stores it, with the derivative taken from the STORED output, as
    vfd_act_backward / the `mul` epilogue do (csrc/common.hpp act_grad_from_out): LeakyReLU by the sign of y, sigmoid y(1-y),
    tanh 1-y^2 — near saturation 1-y^2 of a bf16-rounded y differs from the exact derivative by tens of percent, which is the
    HIP path's arithmetic, not an error of it.  round_g: the incoming gradient is a stored tensor (no consumer claimed it)."""

    @staticmethod
    def forward(ctx, t, kind, slope, round_g):
        if kind == "lrelu":
            y = torch.where(t > 0, t, t * slope)
        elif kind == "sigmoid":
            y = torch.sigmoid(t)
        else:
            y = torch.tanh(t)
        y = rbf(y)
        ctx.save_for_backward(y)
        ctx.meta = (kind, slope, round_g)
        return y

    @staticmethod
    def backward(ctx, g):
        (y,) = ctx.saved_tensors
        kind, slope, round_g = ctx.meta
        if round_g:
            g = rbf(g)
        if kind == "lrelu":
            d = torch.where(y > 0, torch.ones_like(y), torch.full_like(y, slope))
        elif kind == "sigmoid":
            d = y * (1 - y)
        else:
            d = 1 - y * y
        return g * d, None, None, None


def act_from_output(act, t, round_g):
    """The activation module `act` applied in a conv epilogue (see _ActFromOutput)."""
    if isinstance(act, nn.LeakyReLU):
        return _ActFromOutput.apply(t, "lrelu", float(act.negative_slope), round_g)
    if isinstance(act, nn.ReLU):
        return _ActFromOutput.apply(t, "lrelu", 0.0, round_g)
    if isinstance(act, nn.Sigmoid):
        return _ActFromOutput.apply(t, "sigmoid", 0.0, round_g)
    if isinstance(act, nn.Tanh):
        return _ActFromOutput.apply(t, "tanh", 0.0, round_g)
    raise NotImplementedError(type(act).__name__)


def wq(w):
    """The packed bf16 filter copy, straight-through to the float32 master parameter."""
    return w + (rbf(w.detach()) - w.detach())


_CONVS = (nn.Conv2d, nn.Conv3d, nn.ConvTranspose2d, nn.ConvTranspose3d, nn.Linear)
_BNS = (nn.BatchNorm1d, nn.BatchNorm2d, nn.BatchNorm3d)
_ACTS = (nn.ReLU, nn.LeakyReLU, nn.Sigmoid, nn.Tanh)


def conv_q(m, x):
    """The layer `m` with its bf16 filter copy; bias stays float32 (added to the float32 accumulator)."""
    w = wq(m.weight)
    if isinstance(m, nn.Linear):
        return TF.linear(x, w, m.bias)
    if isinstance(m, nn.ConvTranspose2d):
        return TF.conv_transpose2d(x, w, m.bias, m.stride, m.padding, m.output_padding, m.groups, m.dilation)
    if isinstance(m, nn.ConvTranspose3d):
        return TF.conv_transpose3d(x, w, m.bias, m.stride, m.padding, m.output_padding, m.groups, m.dilation)
    if isinstance(m, nn.Conv2d):
        return TF.conv2d(x, w, m.bias, m.stride, m.padding, m.dilation, m.groups)
    return TF.conv3d(x, w, m.bias, m.stride, m.padding, m.dilation, m.groups)


def _triple(v):
    return (v,) * 3 if isinstance(v, int) else tuple(v)


def pool_fusable(pool, dhw):
    """vfd_gan_amd.functional.pool_fusable: kernel == stride, every extent 1 or 2 (not all 1), no padding, input a multiple."""
    k = _triple(pool.kernel_size)
    s = _triple(pool.stride if pool.stride is not None else pool.kernel_size)
    p = _triple(pool.padding)
    return k == s and p == (0, 0, 0) and all(v in (1, 2) for v in k) and k != (1, 1, 1) and all(d % v == 0 for d, v in zip(dhw, k))


class _BnFrom(torch.autograd.Function):
    """y = x * E + F, E = gamma * rstd, F = beta - mean * E with GIVEN batch statistics (those of the producing conv's float32
    accumulators), and the backward of csrc/bn.hip: dx = E * (g - mean(g) - xh * mean(g * xh)), xh = (x - mean) * rstd taken from
    the stored (rounded) x and the saved statistics."""

    @staticmethod
    def forward(ctx, x, mean, rstd, gamma, beta):
        shape = [1, -1] + [1] * (x.dim() - 2)
        E = (gamma * rstd) if gamma is not None else rstd
        Fc = (beta if beta is not None else 0.0) - mean * E
        ctx.save_for_backward(x, mean, rstd, gamma)
        ctx.has = (gamma is not None, beta is not None)
        return x * E.view(shape) + Fc.view(shape)

    @staticmethod
    def backward(ctx, g):
        x, mean, rstd, gamma = ctx.saved_tensors
        shape = [1, -1] + [1] * (x.dim() - 2)
        dims = [0] + list(range(2, x.dim()))
        xh = (x - mean.view(shape)) * rstd.view(shape)
        n = x.numel() // x.shape[1]
        sg = g.sum(dims)
        sgx = (g * xh).sum(dims)
        E = (gamma * rstd) if gamma is not None else rstd
        dx = E.view(shape) * (g - (sg / n).view(shape) - xh * (sgx / n).view(shape))
        return dx, None, None, (sgx if ctx.has[0] else None), (sg if ctx.has[1] else None)


def bn_from(bn, x, t):
    """Training-mode BatchNorm of the stored tensor `x` with the batch statistics of `t` (the producing conv's output BEFORE it
    was rounded: the HIP conv epilogue adds up its float32 accumulators); running statistics updated like torch's."""
    if t is None or not bn.training:
        return bn(x)
    dims = [0] + list(range(2, t.dim()))
    n = t.numel() // t.shape[1]
    td = t.detach().double()
    mean = td.mean(dims)
    var = (td * td).mean(dims) - mean * mean          # the fold of csrc/bn.hip (sum, sum of squares; in double)
    rstd = (1.0 / torch.sqrt(var + bn.eps)).float()
    if bn.track_running_stats:
        with torch.no_grad():
            m = bn.momentum
            bn.running_mean.mul_(1 - m).add_(m * mean.float())
            bn.running_var.mul_(1 - m).add_(m * (var * n / max(n - 1, 1)).float())
            bn.num_batches_tracked += 1
    return _BnFrom.apply(x, mean.float(), rstd, bn.weight, bn.bias)


def bn_act_pool(bn, act, x, pool=None, keep_full=False, t=None):
    """BatchNorm(+activation)(+AvgPool3d) as ONE HIP pass: rounded after the last fused stage.  `t`: the unrounded output of
    the conv that produced x, when its epilogue supplied the statistics."""
    y = bn_from(bn, x, t)
    if act is not None:
        y = act(y)
    if pool is None:
        return R(y)
    if keep_full:
        return R(pool(y)), R(y)
    return R(pool(y))


def run_seq(mods, x, return_last_t=False):
    """vfd_gan_amd.nn.run_fused on a list of stock torch.nn layers, with the bf16 rounding points of the HIP kernels.
    return_last_t: also return the last layer's (a conv's) output before rounding (`last_stats` of run_fused: the caller
    applies the BatchNorm that follows)."""
    mods = list(mods)
    n, i = len(mods), 0
    last_t = None

    def is_conv(j):
        return j < n and isinstance(mods[j], _CONVS) and not isinstance(mods[j], nn.Linear)

    def run_bn(j, x, t):
        bn = mods[j]
        act = mods[j + 1] if (j + 1 < n and isinstance(mods[j + 1], _ACTS)) else None
        k = j + (2 if act is not None else 1)
        if (t is not None and x.dim() == 5 and k < n and isinstance(mods[k], nn.AvgPool3d) and bn.training
                and pool_fusable(mods[k], tuple(x.shape[2:]))):
            return bn_act_pool(bn, act, x, mods[k], t=t), k + 1
        return bn_act_pool(bn, act, x, t=t), k

    while i < n:
        m = mods[i]
        nxt = mods[i + 1] if i + 1 < n else None
        if isinstance(m, _CONVS):
            if isinstance(nxt, _ACTS):
                # conv + activation in one kernel.  Backward: g = dy * act'(y) is stored (RB); dy itself is stored by the
                # consumer's data-gradient kernel unless that kernel applies act' in its own epilogue (the next conv of the
                # same list claims it: nn.run_fused `claim_act_grad`)
                claimed = not isinstance(m, nn.Linear) and is_conv(i + 2)
                x = act_from_output(nxt, RB(conv_q(m, x)), round_g=not claimed)
                i += 2
                continue
            t = conv_q(m, x)
            x = R(t)
            if isinstance(nxt, _BNS) and not isinstance(m, nn.Linear) and nxt.training:
                x, i = run_bn(i + 1, x, t)
                continue
            if i == n - 1:
                last_t = t
            i += 1
            continue
        if isinstance(m, _BNS):
            x, i = run_bn(i, x, None)
            continue
        if isinstance(m, nn.Dropout):
            x = R(m(x)) if (m.training and m.p > 0) else x
        elif isinstance(m, _ACTS):
            x = act_from_output(m, RB(x), round_g=True)      # a pass of its own (vfd_act_forward / vfd_act_backward: derivative from y)
        elif isinstance(m, (nn.AvgPool3d, nn.Upsample)):
            x = R(m(x))
        else:
            raise NotImplementedError("bf16-faithful plan for %s" % type(m).__name__)
        i += 1
    return (x, last_t) if return_last_t else x


# ---- the nets (the fusion structure of vfd_gan_amd/models/*.py) --------------------------------------------------------------
def stconv(m, x, return_last_t=False):
    """SpatioTemporalConv (vfd_gan_amd/models/spatiotempconv.py: run_fused over its four layers)."""
    return run_seq([m.spatial_conv, m.bn, m.relu, m.temporal_conv], x, return_last_t)


def conv_bn_act(block, x, pool=None, keep_full=False):
    """models/mygannet.py:_conv_bn_act (NetgConv / NetdConv): the temporal conv's epilogue supplies the BatchNorm statistics."""
    y, t = stconv(block.conv, x, True)
    if pool is not None and pool_fusable(pool, tuple(y.shape[2:])) and block.bn.training:
        return bn_act_pool(block.bn, block.lrelu, y, pool, keep_full, t=t)
    y = bn_act_pool(block.bn, block.lrelu, y, t=t)
    if pool is None:
        return y
    return (R(pool(y)), y) if keep_full else R(pool(y))


def upsample_cat(up, x, skip):
    """functional.upsample_cat: cat([Upsample(x), skip]) written by one pass (channel count of x a multiple of 8), else the
    two-pass form (up-sampled tensor stored, then concatenated)."""
    return R(torch.cat([up(x), skip], dim=1)) if x.shape[1] % 8 == 0 else torch.cat([R(up(x)), skip], dim=1)


def _drop(m, x):
    return R(m(x)) if (m.training and m.p > 0) else x


def mygan_netg(g, x):
    p1, d1 = conv_bn_act(g.dconv1, x, g.avgpool, True)
    p2, d2 = conv_bn_act(g.dconv2, p1, g.avgpool, True)
    p3, d3 = conv_bn_act(g.dconv3, p2, g.avgpool, True)
    p4, d4 = conv_bn_act(g.dconv4, p3, g.avgpool, True)
    latent = conv_bn_act(g.dconv5, p4)
    x = _drop(g.dropout, conv_bn_act(g.uconv5, latent))
    x = _drop(g.dropout, conv_bn_act(g.uconv4, upsample_cat(g.upsamp, x, d4)))
    x = _drop(g.dropout, conv_bn_act(g.uconv3, upsample_cat(g.upsamp, x, d3)))
    x = _drop(g.dropout, conv_bn_act(g.uconv2, upsample_cat(g.upsamp, x, d2)))
    x = conv_bn_act(g.uconv1, upsample_cat(g.upsamp, x, d1))
    return run_seq([g.conv_last, g.sigmoid], x)


def _disc(d, x, convs):
    for c in convs:
        x = conv_bn_act(c, x, d.avgpool)
    features = x
    x = R(d.gpool(features))
    cls = run_seq([d.linear, d.sigmoid], x.view(x.shape[0], -1))
    return cls.squeeze(1), features


def mygan_netd(d, x, y):
    s = d.spatdisc
    t = d.tempdisc
    s_cls, s_feat = _disc(s, x, (s.dconv1, s.dconv2, s.dconv3, s.dconv4, s.dconv5, s.dconv6))
    t_cls, t_feat = _disc(t, y, (t.dconv1, t.dconv2, t.dconv3))
    return s_cls, s_feat, t_cls, t_feat


def anogan_netg(g, z):
    x = run_seq(g.layer1, z)
    x = x.view(x.size()[0], *g.seed_shape)
    return run_seq(g.layer3, run_seq(g.layer2, x))


def anogan_netd(d, x):
    x = run_seq(d.layer2, run_seq(d.layer1, x))
    x = x.view(x.size()[0], -1)
    return run_seq(d.fc, x), x


def ganomaly_netd(d, x):
    features = run_seq(d.features, x)
    return run_seq(d.classifier, features).view(-1, 1).squeeze(1), features


def ganomaly_netg(g, x):
    latent_i = run_seq(g.encoder1.main, x)
    gen = run_seq(g.decoder.main, latent_i)
    return gen, latent_i, run_seq(g.encoder2.main, gen)


_FORWARD = {OG.NetG: ganomaly_netg, OG.NetD: ganomaly_netd, OG.Encoder: lambda m, x: run_seq(m.main, x),
            OG.Decoder: lambda m, x: run_seq(m.main, x), OA.NetG: anogan_netg, OA.NetD: anogan_netd,
            OM.NetG: mygan_netg, OM.NetD: mygan_netd, SpatioTemporalConv: stconv, nn.Sequential: run_seq}


class Faithful(nn.Module):
    """``Faithful(net)(x)`` = the bf16-faithful forward of oracle net `net` (same parameters: the oracle's step functions,
    optimisers and state_dict comparisons keep working on `net` itself).  Inputs are rounded at the entry (F.to_cl)."""

    def __init__(self, net):
        super().__init__()
        self.net = net
        self._fwd = _FORWARD[type(net)]

    def forward(self, *xs):
        return self._fwd(self.net, *[RF(x) for x in xs])
