"""HIP-backed layers with the constructor signatures, attribute names and ``state_dict`` keys of the torch.nn
layers the reference builds its nets from.

Each class SUBCLASSES its torch.nn counterpart — parameters, buffers, initialisation, ``isinstance`` checks
(the reference's ``weights_init`` dispatches on ``isinstance(m, nn.Conv3d)``, lib/utils.py:51-56) and
checkpoint compatibility come from the parent — and overrides only ``forward``, which runs hand-written HIP
kernels on a :class:`~vfd_gan_amd.functional.ClTensor` (channels-last, bf16/f32).  There is no torch fallback:
calling these layers with CPU tensors raises.
"""
import os

import torch
import torch.nn as tnn

from . import _lib
from . import functional as F
from .functional import ClTensor


def _need_cl(x, who):
    if not isinstance(x, ClTensor):
        raise TypeError("%s expects a ClTensor (use vfd_gan_amd.functional.to_cl at the network entry), got %s"
                        % (who, type(x).__name__))
    return x


def _act_of(m):
    """(act code, slope) of an activation module, or None."""
    if isinstance(m, tnn.LeakyReLU):
        return _lib.ACT_LRELU, float(m.negative_slope)
    if isinstance(m, tnn.ReLU):
        return _lib.ACT_LRELU, 0.0
    if isinstance(m, tnn.Sigmoid):
        return _lib.ACT_SIGMOID, 0.0
    if isinstance(m, tnn.Tanh):
        return _lib.ACT_TANH, 0.0
    return None


# ---- convolution family ------------------------------------------------------------------------------------
class _ConvMixin:
    _transposed = False

    def forward(self, x, act=_lib.ACT_NONE, slope=0.0, stats=None, claim_act_grad=False, bias_token=None, in_bias_token=None):
        _need_cl(x, type(self).__name__)
        if self.groups != 1 or any(d != 1 for d in self.dilation):
            raise NotImplementedError("groups/dilation are not used by the reference nets")
        if self.padding_mode != "zeros":
            raise NotImplementedError("padding_mode %r" % (self.padding_mode,))
        op = self.output_padding if self._transposed else 0
        return F.conv(x, self.weight, self.bias, self.stride, self.padding, op, self._transposed, act, slope, stats,
                      claim_act_grad, bias_token, in_bias_token)


class Conv3d(_ConvMixin, tnn.Conv3d):
    pass


class Conv2d(_ConvMixin, tnn.Conv2d):
    pass


class ConvTranspose3d(_ConvMixin, tnn.ConvTranspose3d):
    _transposed = True


class ConvTranspose2d(_ConvMixin, tnn.ConvTranspose2d):
    _transposed = True


class Linear(tnn.Linear):
    """nn.Linear on an (N, F) block.  When the input is a flattened (N,C,D,H,W) block (``x.view(N, -1)`` in the
    reference, models/anogan.py:115, models/mygannet.py:158,191) pass the un-flattened ClTensor: the layer then
    runs as a convolution whose kernel covers the whole block, with the weight viewed as [out, C, D, H, W] —
    identical arithmetic, no layout shuffle."""

    def forward(self, x, act=_lib.ACT_NONE, slope=0.0):
        _need_cl(x, "Linear")
        n, d, h, w, _ = x.t.shape
        feat = x.C * d * h * w
        if feat != self.in_features:
            raise RuntimeError("Linear: input has %d features, expected %d (shape %s)" % (feat, self.in_features, x.shape))
        if d * h * w == 1:
            y = F.conv(ClTensor(x.t, x.C, 0), self.weight, self.bias, 1, 0, 0, False, act, slope)
        else:
            wv = self.weight.view(self.out_features, x.C, d, h, w)
            y = F.conv(ClTensor(x.t, x.C, 3), wv, self.bias, 1, 0, 0, False, act, slope)
        return ClTensor(y.t, self.out_features, 0)


# ---- normalisation ---------------------------------------------------------------------------------------------
class _BatchNormMixin:
    def forward(self, x, act=_lib.ACT_NONE, slope=0.0, sums=None, bwd_sums=None, conv_bias=None, bias_token=None):
        _need_cl(x, type(self).__name__)
        if x.C != self.num_features:
            raise RuntimeError("BatchNorm: %d channels, expected %d" % (x.C, self.num_features))
        if not self.training and self.track_running_stats:
            # .eval(): normalise with the running statistics (reference models/anogan.py:146-147 puts both nets in eval mode
            # for its test sweep); statistics handed over by a conv epilogue (`sums`) are ignored
            return F.bn_act_eval(x, self.weight, self.bias, self.running_mean, self.running_var, self.eps, act, slope)
        if self.momentum is None:
            raise NotImplementedError("cumulative-average BatchNorm (momentum=None)")
        rm = self.running_mean if self.track_running_stats else None
        rv = self.running_var if self.track_running_stats else None
        nbt = self.num_batches_tracked if self.track_running_stats else None      # += 1 inside the statistics kernel
        y = F.bn_act(x, self.weight, self.bias, rm, rv, self.eps, self.momentum, act, slope, sums, nbt, bwd_sums,
                     conv_bias, bias_token)
        if getattr(self, "_keep_batch_stats", False):
            self._batch_stats = F._LAST_BN_STATS[0]      # (mean, rstd, rows): see repeat_running_update
        return y

    def forward_pooled(self, x, act, slope, sums, bwd_sums, conv_bias=None, bias_token=None, pool=(2, 2, 2), keep_full=False):
        """Training-mode forward fused with the activation and the AvgPool3d (kernel = stride = `pool`) that follow."""
        nbt = self.num_batches_tracked if self.track_running_stats else None
        rm = self.running_mean if self.track_running_stats else None
        rv = self.running_var if self.track_running_stats else None
        y = F.bn_act_pool(x, self.weight, self.bias, rm, rv, self.eps, self.momentum, act, slope, sums, nbt, bwd_sums,
                          conv_bias, bias_token, pool, keep_full)
        if getattr(self, "_keep_batch_stats", False):
            self._batch_stats = F._LAST_BN_STATS[0]
        return y

    def repeat_running_update(self):
        """Apply the running-statistics update of the last training-mode forward once more (a repeated forward on the same
        input and weights, without the forward)."""
        mean, rstd, rows = self._batch_stats
        if self.track_running_stats:
            F.bn_running_update(mean, rstd, rows, self.num_features, self.running_mean, self.running_var, self.eps, self.momentum,
                                self.num_batches_tracked)


class BatchNorm3d(_BatchNormMixin, tnn.BatchNorm3d):
    pass


class BatchNorm2d(_BatchNormMixin, tnn.BatchNorm2d):
    pass


class BatchNorm1d(_BatchNormMixin, tnn.BatchNorm1d):
    pass


# ---- activations / pooling / resampling / dropout -----------------------------------------------------------------
class LeakyReLU(tnn.LeakyReLU):
    def forward(self, x):
        return F.activation(_need_cl(x, "LeakyReLU"), _lib.ACT_LRELU, self.negative_slope)


class ReLU(tnn.ReLU):
    def forward(self, x):
        return F.activation(_need_cl(x, "ReLU"), _lib.ACT_LRELU, 0.0)


class Sigmoid(tnn.Sigmoid):
    def forward(self, x):
        return F.activation(_need_cl(x, "Sigmoid"), _lib.ACT_SIGMOID)


class Tanh(tnn.Tanh):
    def forward(self, x):
        return F.activation(_need_cl(x, "Tanh"), _lib.ACT_TANH)


class AvgPool3d(tnn.AvgPool3d):
    def forward(self, x):
        _need_cl(x, "AvgPool3d")
        k = F._triple(self.kernel_size, 3, 1)
        s = F._triple(self.stride if self.stride is not None else self.kernel_size, 3, 1)
        in_dhw = tuple(x.t.shape[1:4])
        # kernel == stride (ordinary pooling) or a pool that spans the whole extent (the "global" pools of
        # SDisc / TDisc, stride 1, output extent 1)
        for i in range(3):
            if not (k[i] == s[i] or k[i] == in_dhw[i]):
                raise NotImplementedError("AvgPool3d kernel %s stride %s on %s" % (k, s, in_dhw))
        if F._triple(self.padding, 3, 0) != (0, 0, 0):
            raise NotImplementedError("padded AvgPool3d")
        return F.avg_pool(x, k)


class MaxPool3d(tnn.MaxPool3d):
    """nn.MaxPool3d (models/xception.py:57); floor mode, no dilation, no indices."""

    def forward(self, x):
        _need_cl(x, "MaxPool3d")
        if self.ceil_mode or self.return_indices or F._triple(self.dilation, 3, 1) != (1, 1, 1):
            raise NotImplementedError("MaxPool3d: ceil_mode / return_indices / dilation are not used by the reference nets")
        return F.max_pool(x, self.kernel_size, self.stride, self.padding)


class Upsample(tnn.Upsample):
    def factors(self):
        sf = self.scale_factor
        f = tuple(float(v) for v in sf) if isinstance(sf, (tuple, list)) else (float(sf),) * 3
        if not (self.mode == "trilinear" and self.align_corners and len(f) == 3 and all(v in (1.0, 2.0) for v in f)):
            raise NotImplementedError("only Upsample(scale_factor in {1,2}^3, mode='trilinear', align_corners=True)")
        return tuple(int(v) for v in f)

    def check(self):
        if self.factors() != (2, 2, 2):
            raise NotImplementedError("only Upsample(scale_factor=2, mode='trilinear', align_corners=True)")

    def forward(self, x):
        _need_cl(x, "Upsample")
        return F.upsample_trilinear(x, self.factors())


class Dropout(tnn.Dropout):
    def forward(self, x):
        return F.dropout(_need_cl(x, "Dropout"), self.p, self.training)


# ---- Sequential with peephole fusion ---------------------------------------------------------------------------------
_CONVS = (Conv3d, Conv2d, ConvTranspose3d, ConvTranspose2d, Linear)
_BNS = (BatchNorm3d, BatchNorm2d, BatchNorm1d)


_NO_HANDOVER = bool(os.environ.get("VFD_NO_ACT_HANDOVER"))    # tuning / bisecting switch


def run_fused(mods, x, last_stats=None, last_bias_token=None):
    """Run a list of HIP-backed layers, fusing conv->act into the conv epilogue, BatchNorm->act into one
    normalise+activate pass, (bf16) conv->BatchNorm statistics into the conv epilogue, and the activation / BatchNorm
    backward of a producer into the data-gradient epilogue of the conv that consumes it.  `last_stats` / `last_bias_token`:
    statistics buffer / bias-gradient token (see F.bn_act) for the list's LAST layer, a conv whose BatchNorm the caller
    applies itself."""
    i, n = 0, len(mods)
    fresh = False      # x is the (single-consumer) output of a conv+activation or BatchNorm(+activation) of this list
    epi = use_epilogue_stats(x)
    handover = epi and not _NO_HANDOVER

    def is_conv(j):
        return j < n and isinstance(mods[j], _CONVS) and not isinstance(mods[j], Linear)

    def bn_span(j):
        """BatchNorm at j (training) -> index of the layer after BatchNorm(+activation)."""
        return j + 2 if (j + 1 < n and _act_of(mods[j + 1]) is not None) else j + 1

    # one zero-filled allocation for every sum buffer of the list (one fill launch): forward statistics of each
    # conv -> BatchNorm pair (float64: the conv epilogues sum in double), backward sums of each BatchNorm (float32)
    pool, pool64, pool_off, pool64_off = None, None, 0, 0
    if epi:
        need = need64 = 0
        for j, m in enumerate(mods):
            if isinstance(m, _BNS) and m.training:
                if j > 0 and is_conv(j - 1):
                    need64 += F.stats_buffer_numel(m.num_features)      # forward statistics from the conv epilogue
                    if mods[j - 1].bias is not None:
                        need += F.stats_buffer_numel(m.num_features)    # the conv's bias gradient, left by this BatchNorm's apply pass
                need += F.stats_buffer_numel(m.num_features)      # backward sums (every training BatchNorm)
            if handover and is_conv(j) and m.bias is not None and is_conv(j + 1):
                need64 += F.stats_buffer_numel(m.out_channels)      # conv(bias) -> conv: the first one's bias gradient (epilogue sums of the second's data gradient)
        if need or need64:
            pool = F.zeros(need + 2 * need64, torch.float32, x.t.device)
            F.register_pool(pool)
            pool64 = pool[:2 * need64].view(torch.float64)      # the float64 part first (8-byte aligned: start of the allocation)
            pool_off = 2 * need64

    def take(C):
        nonlocal pool_off
        k = F.stats_buffer_numel(C)
        buf = pool.narrow(0, pool_off, k)
        pool_off += k
        return buf

    def take64(C):
        nonlocal pool64_off
        k = F.stats_buffer_numel(C)
        buf = pool64.narrow(0, pool64_off, k)
        pool64_off += k
        return buf

    def pool2_after(j, in_dhw):
        """(index, kernel) of an AvgPool3d right after BatchNorm j and its activation that the BatchNorm pass can absorb."""
        k = bn_span(j)
        if k < n and isinstance(mods[k], AvgPool3d) and not _NO_HANDOVER:
            pm = mods[k]
            ks = F.pool_fusable(pm.kernel_size, pm.stride, pm.padding, in_dhw)
            if ks is not None:
                return k, ks
        return -1, None

    def run_bn(j, x, sums, conv=None, tok=None):
        """BatchNorm at j with the activation that follows it; returns (x, next index, x has a hand-over token).  `conv`
        (with bias token `tok`): the conv that feeds it — its bias gradient is the column sum of this BatchNorm's dx."""
        bn = mods[j]
        nxt_i = bn_span(j)
        a = _act_of(mods[j + 1]) if nxt_i == j + 2 else None
        pk, pks = pool2_after(j, tuple(x.t.shape[1:4])) if x.nsp == 3 else (-1, None)
        if pk >= 0 and sums is not None and bn.training and pool is not None and bn.momentum is not None:
            # BatchNorm -> activation -> AvgPool3d, nothing else reads the activation: one pass writes the pooled tensor
            act, slope = (a[0], a[1]) if a is not None else (_lib.ACT_NONE, 0.0)
            y = bn.forward_pooled(x, act, slope, sums, take(bn.num_features), conv.bias if tok is not None else None, tok, pks)
            return y, pk + 1, False
        kw = {"act": a[0], "slope": a[1]} if a is not None else {}
        if bn.training and pool is not None:
            kw["bwd_sums"] = take(bn.num_features)      # backward: reduce with atomics + folding apply, or the conv hand-over
            if tok is not None:
                kw["conv_bias"], kw["bias_token"] = conv.bias, tok
        give = handover and "bwd_sums" in kw and is_conv(nxt_i)
        return bn(x, sums=sums, **kw) if sums is not None else bn(x, **kw), nxt_i, give

    chain_tok = None      # bias token handed from a conv (with bias) to the conv that directly consumes its output
    while i < n:
        m = mods[i]
        nxt = mods[i + 1] if i + 1 < n else None
        if isinstance(m, _CONVS):
            # x straight out of the previous conv+activation / BatchNorm of THIS list has no other consumer: this conv's data
            # gradient takes that producer's backward into its epilogue (functional._Conv)
            kw = {"claim_act_grad": True} if (fresh and not isinstance(m, Linear) and not _NO_HANDOVER) else {}
            fresh = False
            if chain_tok is not None:
                kw["in_bias_token"] = chain_tok
                chain_tok = None
            if handover and pool is not None and is_conv(i) and m.bias is not None and is_conv(i + 1):
                # conv(bias) -> conv with nothing in between: the consumer's data-gradient launch takes the column sums of
                # its output (this layer's bias gradient) as epilogue statistics
                chain_tok = {"taken": False, "rep": take64(m.out_channels), "stride": 2 * F.cpad(m.out_channels)}
                kw["bias_token"] = chain_tok
            a = _act_of(nxt) if nxt is not None else None
            if a is not None:
                x = m(x, act=a[0], slope=a[1], **kw)
                fresh = not isinstance(m, Linear)
                i += 2
                continue
            if isinstance(nxt, _BNS) and not isinstance(m, Linear) and nxt.training and epi:
                sums = take64(m.out_channels)
                tok = {"taken": False, "rep": take(m.out_channels)} if m.bias is not None else None
                if tok is not None:
                    kw["bias_token"] = tok
                x = m(x, stats=sums, **kw)
                x, i, fresh = run_bn(i + 1, x, sums, m, tok)
                continue
            if last_stats is not None and i == n - 1:
                kw["stats"] = last_stats
                if last_bias_token is not None:
                    kw["bias_token"] = last_bias_token
            x = m(x, **kw)
            i += 1
            continue
        fresh = False
        if isinstance(m, _BNS):
            x, i, fresh = run_bn(i, x, None)
            continue
        x = m(x)
        i += 1
    return x


_EPILOGUE_STATS = {"mode": "auto"}


def set_epilogue_stats(mode):
    """'auto' (bf16: BatchNorm statistics come from the conv epilogue; f32: separate exact pass), 'on', 'off'."""
    assert mode in ("auto", "on", "off")
    _EPILOGUE_STATS["mode"] = mode


def use_epilogue_stats(x):
    mode = _EPILOGUE_STATS["mode"]
    if mode == "auto":
        return x.t.dtype == torch.bfloat16
    return mode == "on"


class Sequential(tnn.Sequential):
    """nn.Sequential (same child naming, hence same state_dict keys) executed through :func:`run_fused`."""

    def forward(self, x):
        return run_fused(list(self), x)
