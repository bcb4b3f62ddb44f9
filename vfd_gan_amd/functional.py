"""Autograd glue over the C ABI (include/vfdgan_hip.h).

Every op here is a `torch.autograd.Function` whose forward and backward launch hand-written HIP kernels from
libvfdgan_hip.so on torch's current stream.  Tensors that flow between ops are channels-last blocks
``t[N, D, H, W, Cp]`` (Cp = C rounded up to 8, pad channels zero) in the compute dtype (bfloat16 or float32),
wrapped in :class:`ClTensor`, which remembers the logical channel count.  PyTorch supplies device memory,
streams and the autograd tape.  The arithmetic of the hot path runs in the library's kernels; what torch's own element-wise
kernels still do per step is bookkeeping on a few hundred floats (rocprofv3, profiles/r03_*_kernel_stats.csv: about 0.4 % of a
step's kernel time): autograd's sum where a small tensor has two consumers, the 864-float permutation of a role-swapped filter
gradient, the dropout / Adam step counters.  Zero fills, the sum of a fan-out's gradients and the loss-term arithmetic go
through vfd_zero / vfd_add / vfd_weighted_sum4.
"""
import contextlib
import ctypes
import weakref
import os

import torch

from . import _lib
from ._lib import ConvDesc, check, dtype_code, load, ptr, stream

_COMPUTE_DTYPE = torch.bfloat16


_FP8 = [False]


def set_compute_dtype(dt):
    """Storage type of activations / packed filters: torch.bfloat16 (default, MFMA bf16) or torch.float32; "fp8" = bf16
    storage with e4m3 OPERANDS for the forward pass and the data gradient of every convolution wide enough for the fp8
    tiles (BASELINE configs[4]; see set_fp8)."""
    global _COMPUTE_DTYPE
    fp8 = isinstance(dt, str) and dt in ("fp8", "e4m3")
    if fp8:
        dt = torch.bfloat16
    if isinstance(dt, str):
        dt = {"bf16": torch.bfloat16, "bfloat16": torch.bfloat16, "f32": torch.float32, "fp32": torch.float32,
              "float32": torch.float32}[dt]
    dtype_code(dt)
    _COMPUTE_DTYPE = dt
    _FP8[0] = fp8


def set_fp8(on):
    """fp8 operand mode (bf16 activations only): a convolution with >= 64 input and > 128 output channels runs its forward
    pass on v_mfma_f32_16x16x128_f8f6f4 with e4m3 copies of its input and filter (per-tensor current scaling: vfd_amax +
    vfd_quantize_fp8 / vfd_pack_filter_fp8), and its data gradient likewise when the roles' channel counts allow; outputs,
    BatchNorm, the filter gradient and the master weights stay bf16 / f32.  Returns the previous setting."""
    prev = _FP8[0]
    _FP8[0] = bool(on)
    return prev


def fp8_enabled():
    return _FP8[0]


_FP8_MIN_OUT = [129]     # tests lower it to 65 to drive the 128 x 128 fp8 tile through a whole step


def _fp8_eligible(K_channels, out_channels, dt):
    # more than 128 output channels: the 16-wave 256 x 256 fp8 tile (1.1-1.6 PFLOP/s on the ganomaly pyramid against
    # 0.8-1.0 for bf16); the 4-wave 128 x 128 fp8 tile (<= 128 output channels) is no faster than the bf16 tiles of that
    # range, so those layers stay bf16
    return _FP8[0] and dt == torch.bfloat16 and K_channels >= 64 and out_channels >= _FP8_MIN_OUT[0]


def get_compute_dtype():
    return _COMPUTE_DTYPE


def cpad(c):
    return (c + 7) & ~7


STATS_REPLICAS = 8  # == VFD_STATS_REPLICAS (include/vfdgan_hip.h); 64 measured no faster (atomic contention is not a cost)

# Debug switch (env VFD_POISON_WS=1 or set_workspace_poison): every kernel workspace (split-K slabs of the filter
# gradient and of few-pixel convolutions, BatchNorm / bias / loss partials) is filled with NaN bit patterns before
# the kernel that writes it runs.  Workspaces come from torch.empty, i.e. hold stale FINITE data of earlier
# launches: a slab that is never written, or read before it is written, would otherwise show up as a small
# deviation (the round-1 observation); poisoned, it turns the result into NaN in one run.
# The GPU test suite runs with the switch on (tests/conftest.py).
_POISON_WS = [bool(int(os.environ.get("VFD_POISON_WS", "0") or 0))]


def set_workspace_poison(on):
    _POISON_WS[0] = bool(on)


def _workspace(nbytes, device):
    """Scratch bytes for one kernel call (torch's caching allocator; stream-ordered reuse)."""
    ws = torch.empty(int(nbytes), dtype=torch.uint8, device=device)
    if _POISON_WS[0] and ws.numel():
        ws.fill_(0xFF)          # 0xFFFFFFFF = NaN as float32; 0xFFFF = NaN as bfloat16
    return ws


# ---- filter gradients on a side stream ----------------------------------------------------------------------------------
# A layer's filter gradient (conv_wgrad + wgrad_reduce: MFMA-bound, needed only by the optimiser) is independent of
# everything that follows it in the backward pass, which continues with the HBM-bound BatchNorm passes of the layer
# below: issued on a second stream the two can share the CUs.  The launch stream waits for the side stream before
# anything reads the gradients (join_side_stream: optimiser step, gradient reduction, end of a captured phase).
# Measured (graph replay): mygan 56.4 -> 54.3 ms, ganomaly 11.33 -> 11.24, anogan unchanged.  VFD_SIDE_WGRAD=0 turns it off.
_SIDE = {"on": bool(int(os.environ.get("VFD_SIDE_WGRAD", "1") or 0)), "stream": None, "dirty": False}


def set_side_wgrad(on):
    prev = _SIDE["on"]
    _SIDE["on"] = bool(on)
    return prev


def _side_stream(device):
    if _SIDE["stream"] is None:
        _SIDE["stream"] = torch.cuda.Stream(device)
    return _SIDE["stream"]


def join_side_stream():
    if _SIDE["dirty"]:
        torch.cuda.current_stream().wait_stream(_SIDE["stream"])
        _SIDE["dirty"] = False


def zero_(t):
    """t[:] = 0 through the library's own fill (vfd_zero): no torch element-wise operator on the hot path."""
    if t.numel():
        if not t.is_contiguous():
            raise RuntimeError("zero_: contiguous tensors only")
        check(load().vfd_zero(t.data_ptr(), t.numel() * t.element_size(), stream()), "zero")
    return t


def zeros(shape, dtype, device):
    return zero_(torch.empty(shape, dtype=dtype, device=device))


class _WeightedSum(torch.autograd.Function):
    """sum_i w_i * t_i of up to four float32 device scalars in ONE launch (backward: one launch for all terms): the
    `err_g = err_g_adv * w_adv + err_g_con * w_con + ...` lines of the reference steps (models/ganomaly.py:487-490,
    models/mygannet.py:416-433), which as torch expressions cost a multiply / add launch per operator in both directions."""

    @staticmethod
    def forward(ctx, weights, *terms):
        n = len(terms)
        ts = [t.detach().float().contiguous() for t in terms]
        out = torch.empty((), dtype=torch.float32, device=ts[0].device)
        w = list(weights) + [0.0] * (4 - n)
        p = [t.data_ptr() for t in ts] + [None] * (4 - n)
        check(load().vfd_weighted_sum4(p[0], p[1], p[2], p[3], w[0], w[1], w[2], w[3], n, out.data_ptr(), stream()), "weighted_sum4")
        ctx.weights = w
        ctx.n = n
        return out

    @staticmethod
    def backward(ctx, g):
        n, w = ctx.n, ctx.weights
        g = g.contiguous().float()
        out = torch.empty(4, dtype=torch.float32, device=g.device)
        check(load().vfd_scale4(g.data_ptr(), w[0], w[1], w[2], w[3], n, out.data_ptr(), stream()), "scale4")
        return (None,) + tuple(out[i] if ctx.needs_input_grad[1 + i] else None for i in range(n))


def weighted_sum(*pairs):
    """weighted_sum((t0, w0), (t1, w1), ...) -> float32 scalar tensor sum_i w_i * t_i (1..4 terms, summed left to right)."""
    if not 1 <= len(pairs) <= 4:
        raise ValueError("weighted_sum takes 1..4 (tensor, weight) pairs")
    terms = [t for t, _ in pairs]
    for t in terms:
        if t.numel() != 1 or t.dtype != torch.float32 or not t.is_cuda:
            raise RuntimeError("weighted_sum: float32 device scalars only")
    return _WeightedSum.apply(tuple(float(w) for _, w in pairs), *terms)


def stats_buffer_numel(C):
    """Elements of a [STATS_REPLICAS][2][CPAD(C)] sums buffer: float64 for a conv epilogue's forward statistics
    (new_stats_buffer), float32 for the BatchNorm backward sums / bias-gradient rows."""
    return STATS_REPLICAS * 2 * cpad(C)


def new_stats_buffer(channels, device):
    """Zeroed [STATS_REPLICAS][2][CPAD(C)] FLOAT64 buffer for a conv epilogue's BatchNorm partial sums (the kernels add up
    their float32 accumulators in double: include/vfdgan_hip.h, vfd_conv_forward)."""
    return zeros(STATS_REPLICAS * 2 * cpad(channels), torch.float64, device)


def _stats_arg(stats):
    """(pointer, bytes) of a conv-epilogue statistics buffer; float64 is part of the contract (a float32 buffer of the right
    byte size would be half as many elements as the kernel writes)."""
    if stats is None:
        return 0, 0
    if stats.dtype != torch.float64:
        raise TypeError("conv epilogue statistics are float64 (functional.new_stats_buffer), got %s" % stats.dtype)
    return stats.data_ptr(), stats.numel() * 8


class ClTensor:
    """A channels-last activation block plus its logical channel count.

    ``t``   torch tensor [N, D, H, W, Cp] (part of the autograd graph)
    ``C``   logical channels (pad channels t[..., C:] are zero)
    ``nsp`` spatial rank seen by the user: 3 -> (N,C,D,H,W), 2 -> (N,C,H,W), 0 -> (N,C)
    """
    __slots__ = ("t", "C", "nsp", "fused_act", "fused_bn")

    def __init__(self, t, C, nsp, fused_act=None):
        self.t, self.C, self.nsp = t, C, nsp
        # set by conv(): this block is the output of a conv with a fused activation; a sole consumer may claim the
        # activation's gradient into its own data-gradient epilogue (see _Conv)
        self.fused_act = fused_act
        self.fused_bn = None      # hand-over token of the BatchNorm(+activation) that produced t (see _BnAct)

    @property
    def shape(self):
        n, d, h, w, _ = self.t.shape
        return {3: (n, self.C, d, h, w), 2: (n, self.C, h, w), 0: (n, self.C)}[self.nsp]

    @property
    def rows(self):
        n, d, h, w, _ = self.t.shape
        return n * d * h * w

    @property
    def dtype(self):
        return self.t.dtype

    def detach(self):
        return ClTensor(self.t.detach(), self.C, self.nsp)

    def size(self, i=None):
        return self.shape if i is None else self.shape[i]

    def to_torch(self):
        """float32 tensor in the reference's layout (N,C,[D,]H,W)."""
        return from_cl(self)

    def __repr__(self):
        return "ClTensor(shape=%s, dtype=%s)" % (self.shape, self.t.dtype)


# ---------------------------------------------------------------------------------------------------------
# boundary layout conversion
# ---------------------------------------------------------------------------------------------------------
class _ToCl(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, dt):
        _lib.require_device(x)
        x = x.contiguous().float()
        N, C = x.shape[0], x.shape[1]
        sp = tuple(x.shape[2:])
        S = 1
        for s in sp:
            S *= s
        dhw = {3: sp, 2: (1,) + sp, 0: (1, 1, 1)}[len(sp)]
        out = torch.empty((N,) + dhw + (cpad(C),), dtype=dt, device=x.device)
        check(load().vfd_ncs_to_nsc(dtype_code(dt), x.data_ptr(), out.data_ptr(), N, C, S, stream()), "ncs_to_nsc")
        ctx.meta = (N, C, S, tuple(x.shape))
        return out

    @staticmethod
    def backward(ctx, g):
        N, C, S, shape = ctx.meta
        g = g.contiguous()
        out = torch.empty(shape, dtype=torch.float32, device=g.device)
        check(load().vfd_nsc_to_ncs(dtype_code(g.dtype), g.data_ptr(), out.data_ptr(), N, C, S, stream()), "nsc_to_ncs")
        return out, None


class _FromCl(torch.autograd.Function):
    @staticmethod
    def forward(ctx, t, C, nsp):
        t = t.contiguous()
        N, D, H, W, Cp = t.shape
        S = D * H * W
        shape = {3: (N, C, D, H, W), 2: (N, C, H, W), 0: (N, C)}[nsp]
        out = torch.empty(shape, dtype=torch.float32, device=t.device)
        check(load().vfd_nsc_to_ncs(dtype_code(t.dtype), t.data_ptr(), out.data_ptr(), N, C, S, stream()), "nsc_to_ncs")
        ctx.meta = (N, C, S, tuple(t.shape), t.dtype)
        return out

    @staticmethod
    def backward(ctx, g):
        N, C, S, shape, dt = ctx.meta
        g = g.contiguous().float()
        out = torch.empty(shape, dtype=dt, device=g.device)
        check(load().vfd_ncs_to_nsc(dtype_code(dt), g.data_ptr(), out.data_ptr(), N, C, S, stream()), "ncs_to_nsc")
        return out, None, None


def to_cl(x, dtype=None):
    """(N,C,[D,]H,W) or (N,C) float tensor -> ClTensor in the compute dtype."""
    if isinstance(x, ClTensor):
        return x
    nsp = x.dim() - 2
    if nsp not in (0, 2, 3):
        raise ValueError("expected (N,C), (N,C,H,W) or (N,C,D,H,W), got %s" % (tuple(x.shape),))
    return ClTensor(_ToCl.apply(x, dtype or _COMPUTE_DTYPE), x.shape[1], nsp)


def from_cl(x):
    if not isinstance(x, ClTensor):
        return x
    return _FromCl.apply(x.t, x.C, x.nsp)


class _Reshape(torch.autograd.Function):
    """unflatten: (N, C*S) feature vector -> (N, C, D, H, W) block;  flatten: the inverse (reference order)."""

    @staticmethod
    def forward(ctx, t, C, dhw, unflatten):
        t = t.contiguous()
        N = t.shape[0]
        S = dhw[0] * dhw[1] * dhw[2]
        lib = load()
        if unflatten:
            out = torch.empty((N,) + tuple(dhw) + (cpad(C),), dtype=t.dtype, device=t.device)
            check(lib.vfd_unflatten(dtype_code(t.dtype), t.data_ptr(), out.data_ptr(), N, C, S, stream()), "unflatten")
        else:
            out = torch.empty((N, 1, 1, 1, C * S), dtype=t.dtype, device=t.device)
            check(lib.vfd_flatten(dtype_code(t.dtype), t.data_ptr(), out.data_ptr(), N, C, S, stream()), "flatten")
        ctx.meta = (N, C, S, tuple(t.shape), unflatten)
        return out

    @staticmethod
    def backward(ctx, g):
        N, C, S, shape, unflatten = ctx.meta
        g = g.contiguous()
        out = torch.empty(shape, dtype=g.dtype, device=g.device)
        lib = load()
        if unflatten:
            check(lib.vfd_flatten(dtype_code(g.dtype), g.data_ptr(), out.data_ptr(), N, C, S, stream()), "flatten")
        else:
            check(lib.vfd_unflatten(dtype_code(g.dtype), g.data_ptr(), out.data_ptr(), N, C, S, stream()), "unflatten")
        return out, None, None, None


def unflatten(x, shape):
    """``x.view(N, C, D, H, W)`` of an (N, C*D*H*W) ClTensor (reference models/anogan.py:76)."""
    C, dhw = shape[0], tuple(shape[1:])
    if x.nsp != 0 or x.C != C * dhw[0] * dhw[1] * dhw[2] or x.C % 8:
        raise RuntimeError("unflatten: cannot view %s as %s" % (x.shape, shape))
    return ClTensor(_Reshape.apply(x.t, C, dhw, True), C, 3)


def flatten(x):
    """``x.view(N, -1)`` of an (N, C, D, H, W) ClTensor, in the reference's (C, D, H, W) feature order."""
    n, d, h, w, _ = x.t.shape
    if (x.C * d * h * w) % 8:
        raise RuntimeError("flatten: %s has a feature count that is not a multiple of 8" % (x.shape,))
    return ClTensor(_Reshape.apply(x.t, x.C, (d, h, w), False), x.C * d * h * w, 0)


def _direct_grad(param):
    """Parameters owned by vfd_gan_amd.optim.Adam carry a gradient that is a view into the optimiser's flat arena
    (zeroed by zero_grad).  Backward kernels then ACCUMULATE straight into it (no separate gradient tensor, no
    autograd `+=` pass) and return None to autograd.  The leaf's AccumulateGrad node still runs once per backward,
    after its last contribution, so post-accumulate hooks (the data-parallel reducer's) keep firing at the right time."""
    if param is not None and getattr(param, "_vfd_direct_grad", False) and param.grad is not None:
        return param.grad
    return None


# ---------------------------------------------------------------------------------------------------------
# convolution family
# ---------------------------------------------------------------------------------------------------------
_WEIGHT_EPOCH = [0]


def invalidate_weight_cache():
    """Packed (bf16, K-major) filter copies are cached per parameter; bump this after mutating parameters
    behind autograd's back (``param.data.normal_()``, a fused optimiser step)."""
    _WEIGHT_EPOCH[0] += 1


def _triple(v, nsp, fill=1):
    """Per-dimension (D,H,W) triple from an int or an nsp-tuple; missing leading dims get `fill`."""
    if isinstance(v, int):
        v = (v,) * max(nsp, 1) if nsp > 0 else ()
    v = tuple(v)
    return (fill,) * (3 - len(v)) + v if len(v) < 3 else v


_PACK_REGISTRY = {}      # id(optimiser epoch cell) -> {"cell", "entries": [...], "table": device jobs or None, "blocks"}


def _packed_filter(weight, dt, transpose_ab, A, B, T):
    """Cached K-major copy of a filter parameter (see vfd_pack_filter).  The copy's buffer is allocated once and re-packed
    IN PLACE (a captured graph keeps pointing at it)."""
    cache = weight.__dict__.setdefault("_vfd_packed", {})
    key = (dt, transpose_ab)
    # a parameter owned by vfd_gan_amd.optim.Adam also carries its optimiser's own epoch cell: the fused Adam kernel of
    # ONE net then invalidates that net's packed copies only, and the optimiser re-packs all of them in one launch right
    # after its step (repack_owned) so that this function hits
    own = getattr(weight, "_vfd_epoch", None)
    tag = (weight._version, _WEIGHT_EPOCH[0], own[0] if own is not None else 0, weight.data_ptr())
    hit = cache.get(key)
    if hit is not None and hit[0] == tag:
        return hit[1]
    R, Cc = (B, A) if transpose_ab else (A, B)
    shape = (R, T, cpad(Cc))
    w = weight.detach()
    direct = w.is_contiguous() and w.dtype == torch.float32
    if not direct:
        w = w.contiguous().float()
    if hit is not None and tuple(hit[1].shape) == shape and hit[1].dtype == dt and hit[1].device == weight.device:
        out = hit[1]
    else:
        out = torch.empty(shape, dtype=dt, device=weight.device)
        hit = None
    check(load().vfd_pack_filter(dtype_code(dt), w.data_ptr(), out.data_ptr(), A, B, T, int(transpose_ab), stream()),
          "pack_filter")
    entry = [tag, out]
    cache[key] = entry
    if own is not None and direct and hit is None:
        reg = _PACK_REGISTRY.setdefault(id(own), {"cell": own, "entries": [], "table": None, "blocks": 0})
        reg["entries"] = [e for e in reg["entries"] if not (e[0]() is weight and e[1] == key)]
        reg["entries"].append((weakref.ref(weight), key, A, B, T, int(transpose_ab), dtype_code(dt)))
        reg["table"] = None           # rebuilt (one host-to-device copy) at the optimiser's next step
    return out


def repack_owned(cell):
    """Re-pack, in ONE launch, every cached filter copy of the parameters that carry the optimiser epoch cell `cell` (called
    by vfd_gan_amd.optim.Adam.step right after the update, with the cell already bumped) and mark them fresh."""
    reg = _PACK_REGISTRY.get(id(cell))
    if reg is None or not reg["entries"]:
        return
    lib = load()
    live = []
    for (wref, key, A, B, T, tr, dtc) in reg["entries"]:
        weight = wref()
        ent = weight.__dict__.get("_vfd_packed", {}).get(key) if weight is not None else None
        if ent is not None and getattr(weight, "_vfd_epoch", None) is cell:
            live.append((weight, key, A, B, T, tr, dtc, ent))
    if len(live) != len(reg["entries"]):
        reg["entries"] = [(weakref.ref(e[0]),) + e[1:7] for e in live]
        reg["table"] = None
    if not live:
        return
    sig = tuple((e[0].data_ptr(), e[7][1].data_ptr()) for e in live)
    if reg["table"] is None or reg.get("sig") != sig:
        if torch.cuda.is_current_stream_capturing():
            raise RuntimeError("the packed-filter table changed during graph capture: run one full eager step first (GraphedStep warmup)")
        rows, first = [], 0
        for (weight, key, A, B, T, tr, dtc, ent) in live:
            rows.append([weight.data_ptr(), ent[1].data_ptr(), A, B, T, tr, dtc, first])
            first += int(lib.vfd_pack_filter_blocks(A, B, T, tr))
        reg["table"] = torch.tensor(rows, dtype=torch.int64).to(live[0][0].device)
        reg["blocks"] = first
        reg["sig"] = sig
    check(lib.vfd_pack_filters(reg["table"].data_ptr(), len(live), reg["blocks"], stream()), "pack_filters")
    for (weight, key, A, B, T, tr, dtc, ent) in live:
        ent[0] = (weight._version, _WEIGHT_EPOCH[0], cell[0], weight.data_ptr())


def _make_desc(N, in_dhw, Cin, out_dhw, Cout, k, s, p, transposed, dt, act=0, slope=0.0):
    d = ConvDesc()
    d.N = N
    d.Di, d.Hi, d.Wi = in_dhw
    d.Cin = Cin
    d.Do, d.Ho, d.Wo = out_dhw
    d.Cout = Cout
    d.kd, d.kh, d.kw = k
    d.sd, d.sh, d.sw = s
    d.pd, d.ph, d.pw = p
    d.transposed = int(transposed)
    d.dtype = dtype_code(dt)
    d.act = act
    d.slope = slope
    return d


class KernelTimer:
    """Optional per-launch timing of the MFMA kernels with HIP events on the launch stream (bench.py's live
    roofline).  ``records`` holds (kernel name, algorithmic FLOPs, start event, end event)."""

    def __init__(self):
        self.records = []

    def by_geometry(self):
        """(kernel, geometry) -> launches / total ms / TFLOP/s; for tuning (bench.py --layers)."""
        torch.cuda.synchronize()
        out = {}
        for rec in self.records:
            name, flops, e0, e1 = rec[:4]
            d = out.setdefault((name, rec[4] if len(rec) > 4 else ""), {"launches": 0, "flops": 0.0, "ms": 0.0})
            d["launches"] += 1
            d["flops"] += flops
            d["ms"] += e0.elapsed_time(e1)
        return out

    def summary(self):
        torch.cuda.synchronize()
        out = {}
        for name, flops, e0, e1, *_ in self.records:
            d = out.setdefault(name, {"launches": 0, "flops": 0.0, "ms": 0.0})
            d["launches"] += 1
            d["flops"] += flops
            d["ms"] += e0.elapsed_time(e1)
        return out


_TIMER = [None]


def set_kernel_timer(t):
    _TIMER[0] = t


def _conv_kernel_name(desc, stats=None):
    """Name of the kernel vfd_conv_forward dispatches this layer to (asked of the library, not restated here)."""
    buf = ctypes.create_string_buffer(96)
    check(load().vfd_conv_kernel_name(ctypes.byref(desc), int(stats is not None), buf, 96), "conv_kernel_name")
    return buf.value.decode()


def _geom_str(desc, stats=None):
    return "%s N%d %dx%dx%d c%d -> %dx%dx%d c%d k%d%d%d s%d%d%d p%d%d%d%s%s" % (
        "T" if desc.transposed else "C", desc.N, desc.Di, desc.Hi, desc.Wi, desc.Cin, desc.Do, desc.Ho, desc.Wo, desc.Cout,
        desc.kd, desc.kh, desc.kw, desc.sd, desc.sh, desc.sw, desc.pd, desc.ph, desc.pw,
        " act%d" % desc.act if desc.act else "", " stats" if stats is not None else "")


def _conv_flops(desc):
    taps = desc.kd * desc.kh * desc.kw
    if desc.transposed:
        px = desc.N * desc.Di * desc.Hi * desc.Wi
    else:
        px = desc.N * desc.Do * desc.Ho * desc.Wo
    return 2.0 * px * taps * desc.Cin * desc.Cout


def _conv_launch(desc, x, packed, bias, out, stats=None, mul=None, bn=None):
    """`mul` = (tensor of out's shape, act, slope): out *= act'(tensor) in the epilogue (vfd_conv_forward_mul).
    `bn` = hand-over token of a BatchNorm+activation producer (_BnAct): out = g and the sums of g, g*xhat go to the token's
    buffer (vfd_conv_forward_bn_backward)."""
    timer = _TIMER[0]
    if timer is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    lib = load()
    if bn is not None:
        assert stats is None and mul is None and bias is None and tuple(bn["x"].shape) == tuple(out.shape) and bn["x"].dtype == out.dtype
        check(lib.vfd_conv_forward_bn_backward(ctypes.byref(desc), x.data_ptr(), packed.data_ptr(), out.data_ptr(), bn["x"].data_ptr(),
                                               bn["mean"].data_ptr(), bn["rstd"].data_ptr(), ptr(bn["gamma"]), ptr(bn["beta"]),
                                               int(bn["act"]), float(bn["slope"]), bn["sums"].data_ptr(), bn["sums"].numel() * 4,
                                               stream()), "conv_forward_bn_backward")
    elif mul is not None:
        assert stats is None and tuple(mul[0].shape) == tuple(out.shape) and mul[0].dtype == out.dtype
        check(lib.vfd_conv_forward_mul(ctypes.byref(desc), x.data_ptr(), packed.data_ptr(), ptr(bias), out.data_ptr(),
                                       mul[0].data_ptr(), int(mul[1]), float(mul[2]), stream()), "conv_forward_mul")
    else:
        need = ctypes.c_size_t()
        check(lib.vfd_conv_workspace(ctypes.byref(desc), int(stats is not None), ctypes.byref(need)), "conv_workspace")
        ws = _workspace(need.value, x.device) if need.value else None
        check(lib.vfd_conv_forward(ctypes.byref(desc), x.data_ptr(), packed.data_ptr(), ptr(bias), out.data_ptr(),
                                   *_stats_arg(stats), ptr(ws), need.value, stream()),
              "conv_forward")
    if timer is not None:
        e1.record()
        # (the BatchNorm hand-over runs an epilogue variant of its own, which also does BatchNorm's backward reduce: its own row)
        timer.records.append((_conv_kernel_name(desc, stats) + ("+bn_bwd" if bn is not None else ""), _conv_flops(desc), e0, e1,
                              _geom_str(desc, stats)))


class _Conv(torch.autograd.Function):
    """y = act(conv(x, w) + b) for regular and transposed convolutions (fwd: implicit GEMM; bwd: gather-form data
    gradient through the same kernel with swapped roles + split-K filter gradient)."""

    @staticmethod
    def forward(ctx, x, weight, bias, geom, stats, in_act, out_act, bias_token=None, in_bias_token=None):
        (Cin, Cout, k, s, p, out_dhw, transposed, act, slope) = geom
        x = x.contiguous()
        N = x.shape[0]
        in_dhw = tuple(x.shape[1:4])
        dt = x.dtype
        T = k[0] * k[1] * k[2]
        A, B = (Cin, Cout) if transposed else (Cout, Cin)  # torch filter layout [A][B][T]
        b = bias.detach() if bias is not None else None
        if _fp8_eligible(Cin, Cout, dt):
            xq, xs = _quantized(x, Cin)
            wq, ws = _packed_filter_fp8(weight, bool(transposed), A, B, T)
            out = conv_fp8(xq, xs, wq, ws, b, N, in_dhw, Cin, out_dhw, Cout, k, s, p, transposed, act, slope, stats)
        else:
            packed = _packed_filter(weight, dt, transpose_ab=bool(transposed), A=A, B=B, T=T)
            out = torch.empty((N,) + tuple(out_dhw) + (cpad(Cout),), dtype=dt, device=x.device)
            desc = _make_desc(N, in_dhw, Cin, out_dhw, Cout, k, s, p, transposed, dt, act, slope)
            _conv_launch(desc, x, packed, b, out, stats)
        ctx.geom = geom
        ctx.in_dhw = in_dhw
        ctx.has_bias = bias is not None
        ctx.bias_param = bias
        ctx.weight_param = weight
        # activation-gradient hand-over (nn.run_fused only, where the producer's output has exactly one consumer):
        # `in_act` is the token of the conv+activation that produced x; claiming it means THIS layer's data gradient is
        # delivered already multiplied by act'(x) (vfd_conv_forward_mul) and the producer skips its act_backward pass.
        ctx.in_act = None
        ctx.in_bn = None
        if in_act is not None and ctx.needs_input_grad[0]:
            if "sums" in in_act:
                # BatchNorm(+activation) producer: this layer's data gradient also carries BatchNorm's backward reduce
                # (vfd_conv_forward_bn_backward), where the data-gradient kernel has the row-store epilogue
                ddesc = _make_desc(N, out_dhw, Cout, in_dhw, Cin, k, s, p, not transposed, dt)
                if load().vfd_conv_bn_backward_supported(ctypes.byref(ddesc)):
                    in_act["claimed"] = True
                    ctx.in_bn = in_act
            else:
                in_act["claimed"] = True
                ctx.in_act = (in_act["act"], in_act["slope"])
        ctx.out_act = out_act
        ctx.bias_token = bias_token
        # `in_bias_token`: the conv (with a bias, nothing in between) that produced x asks for the column sums of THIS layer's
        # data gradient — its bias gradient — which the data-gradient launch can take as its epilogue "statistics"
        ctx.in_bias_token = in_bias_token
        ctx.save_for_backward(x, weight, out if act != _lib.ACT_NONE else None)
        return out

    @staticmethod
    def backward(ctx, gy):
        (Cin, Cout, k, s, p, out_dhw, transposed, act, slope) = ctx.geom
        x, weight, y = ctx.saved_tensors
        lib = load()
        gy = gy.contiguous()
        dt = x.dtype
        N = x.shape[0]
        in_dhw = ctx.in_dhw
        T = k[0] * k[1] * k[2]
        rows_out = N * out_dhw[0] * out_dhw[1] * out_dhw[2]
        if act != _lib.ACT_NONE and not (ctx.out_act is not None and ctx.out_act["claimed"]):
            # g = dy * act'(y), from the saved output (unless the consumer already applied it, see forward)
            g = torch.empty_like(gy)
            check(lib.vfd_act_backward(dtype_code(dt), y.data_ptr(), gy.data_ptr(), g.data_ptr(), rows_out, Cout, act,
                                       slope, stream()), "act_backward")
            gy = g
        A, B = (Cin, Cout) if transposed else (Cout, Cin)
        gx = gw = gb = None
        bias_done = False
        frozen = _frozen(ctx.weight_param)
        if ctx.needs_input_grad[0] and x.data_ptr() not in _SKIP_INPUT_GRAD:
            # data gradient = the opposite kind of convolution with the A/B-swapped filter packing
            if ctx.in_act is None and ctx.in_bn is None and _fp8_eligible(Cout, Cin, dt):
                gq, gs = _quantized(gy, Cout)
                wq, ws = _packed_filter_fp8(weight, not transposed, A, B, T)
                gx = conv_fp8(gq, gs, wq, ws, None, N, out_dhw, Cout, in_dhw, Cin, k, s, p, not transposed)
            else:
                packed = _packed_filter(weight, dt, transpose_ab=not transposed, A=A, B=B, T=T)
                gx = torch.empty_like(x)
                desc = _make_desc(N, out_dhw, Cout, in_dhw, Cin, k, s, p, not transposed, dt)
                itok = ctx.in_bias_token
                if itok is not None and ctx.in_act is None and ctx.in_bn is None:
                    # the sum row of the epilogue statistics of gx = the producer conv's bias gradient (folded by its own
                    # vfd_wgrad_reduce_bias): no separate column-sum pass over gx
                    _conv_launch(desc, gy, packed, None, gx, stats=itok["rep"])
                    itok["taken"] = True
                else:
                    _conv_launch(desc, gy, packed, None, gx, mul=(x,) + ctx.in_act if ctx.in_act is not None else None, bn=ctx.in_bn)
        if ctx.needs_input_grad[1] and not frozen:
            # A stride-1 convolution IS a transposed convolution with the taps reversed and padding k-1-p, whose filter is
            # w'[ci][co][t] = w[co][ci][T-1-t].  The filter-gradient kernel keeps ONE operand's channels as tile rows (64 at least) and
            # gathers the other tap by tap: for a layer with <= 8 output channels (mygan's conv_last, 32 -> 1: models/mygannet.py:70)
            # the regular form pads dy's 1 channel to a 64-row tile (1.83 ms for 11 GFLOP); in the transposed form the rows are x's
            # channels and dy is the gathered side, a quarter of the padded work.  The slabs then hold dw', permuted back below.
            # Generally: padded tile area (64-row x 128-column granularity) of the two orientations; the transposed form is taken
            # when it is under 0.7 of the regular one (also mygan's 86 -> 32 (3,1,1) factor: 64 x 384 against 128 x 128) and the layer
            # is not one of conv_wgrad_halo's.
            swap = False
            if (not transposed and s == (1, 1, 1) and dt == torch.bfloat16 and all(k[i] - 1 - p[i] >= 0 for i in range(3))
                    and not (k[1] == 3 and k[2] == 3 and Cin >= 33 and Cout >= 33)):
                def area(a, b):
                    return 64 * ((a + 63) // 64) * 128 * ((T * cpad(b) + 127) // 128)
                swap = area(Cin, Cout) < 0.7 * area(Cout, Cin)
            if swap:
                desc = _make_desc(N, in_dhw, Cin, out_dhw, Cout, k, s, tuple(k[i] - 1 - p[i] for i in range(3)), True, dt)
            else:
                desc = _make_desc(N, in_dhw, Cin, out_dhw, Cout, k, s, p, transposed, dt)
            nsplit = ctypes.c_int32()
            nbytes = ctypes.c_size_t()
            check(lib.vfd_wgrad_workspace(ctypes.byref(desc), ctypes.byref(nsplit), ctypes.byref(nbytes)), "wgrad_workspace")
            timer = _TIMER[0]
            direct = _direct_grad(weight)
            tok = ctx.bias_token
            bdirect = _direct_grad(ctx.bias_param) if (tok is not None and tok["taken"] and ctx.has_bias and ctx.needs_input_grad[2]) else None
            side = _side_stream(x.device) if (_SIDE["on"] and direct is not None and timer is None) else None
            if side is not None:
                side.wait_stream(torch.cuda.current_stream())      # x, gy (and the bias sums) are products of the launch stream
                for t in (x, gy) + ((tok["rep"],) if bdirect is not None else ()):
                    t.record_stream(side)
                _SIDE["dirty"] = True
            with torch.cuda.stream(side) if side is not None else contextlib.nullcontext():
                ws = _workspace(nbytes.value, x.device)
                if timer is not None:
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                check(lib.vfd_conv_wgrad(ctypes.byref(desc), x.data_ptr(), gy.data_ptr(), ws.data_ptr(), nbytes.value, stream()),
                      "conv_wgrad")
                if timer is not None:
                    e1.record()
                    buf = ctypes.create_string_buffer(64)
                    check(lib.vfd_wgrad_kernel_name(ctypes.byref(desc), buf, 64), "wgrad_kernel_name")
                    timer.records.append((buf.value.decode(), _conv_flops(desc), e0, e1, _geom_str(desc)))
                if swap:
                    tw = torch.empty((Cin, Cout, T), dtype=torch.float32, device=x.device)
                    check(lib.vfd_wgrad_reduce(ctypes.byref(desc), ws.data_ptr(), tw.data_ptr(), 0.0, stream()), "wgrad_reduce")
                    gsw = tw.flip(2).permute(1, 0, 2).reshape(weight.shape)      # dw[co][ci][t] = dw'[ci][co][T-1-t]  (864 floats)
                    if direct is not None:
                        direct.add_(gsw)
                    else:
                        gw = gsw.contiguous()
                elif direct is not None and bdirect is not None:
                    # the BatchNorm that consumes this conv's output left the column sums of its dx (= this layer's bias
                    # gradient) in replica rows: folded by the same launch
                    check(lib.vfd_wgrad_reduce_bias(ctypes.byref(desc), ws.data_ptr(), direct.data_ptr(), 1.0, tok["rep"].data_ptr(),
                                                    int(tok.get("stride", cpad(Cout))), int(tok["rep"].dtype == torch.float64),
                                                    bdirect.data_ptr(), stream()), "wgrad_reduce_bias")
                    bias_done = True
                elif direct is not None:
                    check(lib.vfd_wgrad_reduce(ctypes.byref(desc), ws.data_ptr(), direct.data_ptr(), 1.0, stream()), "wgrad_reduce")
                else:
                    gw = torch.empty(weight.shape, dtype=torch.float32, device=x.device)
                    check(lib.vfd_wgrad_reduce(ctypes.byref(desc), ws.data_ptr(), gw.data_ptr(), 0.0, stream()), "wgrad_reduce")
        if ctx.has_bias and ctx.needs_input_grad[2] and not bias_done and not frozen:
            bias = ctx.bias_param
            direct = _direct_grad(bias)
            bws = _workspace(lib.vfd_bias_grad_workspace(Cout), x.device)
            if direct is not None:
                check(lib.vfd_bias_grad(dtype_code(dt), gy.data_ptr(), direct.data_ptr(), rows_out, Cout, 1.0, bws.data_ptr(),
                                        stream()), "bias_grad")
            else:
                gb = torch.empty(Cout, dtype=torch.float32, device=x.device)
                check(lib.vfd_bias_grad(dtype_code(dt), gy.data_ptr(), gb.data_ptr(), rows_out, Cout, 0.0, bws.data_ptr(),
                                        stream()), "bias_grad")
        return gx, gw, gb, None, None, None, None, None, None


def conv(x, weight, bias, stride, padding, output_padding=0, transposed=False, act=_lib.ACT_NONE, slope=0.0,
         stats=None, claim_act_grad=False, bias_token=None, in_bias_token=None):
    """Convolution on a ClTensor.  `weight` is the torch-layout float32 parameter ([Cout,Cin,k..] or, transposed,
    [Cin,Cout,k..]); Linear layers pass a [out,in] matrix with x.nsp == 0."""
    nsp = x.nsp
    if weight.dim() == 2:
        k = (1, 1, 1)
        Cout, Cin = weight.shape
    else:
        k = _triple(tuple(weight.shape[2:]), nsp, 1)
        if transposed:
            Cin, Cout = weight.shape[0], weight.shape[1]
        else:
            Cout, Cin = weight.shape[0], weight.shape[1]
    if Cin != x.C:
        raise RuntimeError("conv: input has %d channels, filter expects %d" % (x.C, Cin))
    s = _triple(stride, nsp, 1)
    p = _triple(padding, nsp, 0)
    op = _triple(output_padding, nsp, 0)
    in_dhw = tuple(x.t.shape[1:4])
    if transposed:
        out_dhw = tuple((in_dhw[i] - 1) * s[i] - 2 * p[i] + k[i] + op[i] for i in range(3))
    else:
        out_dhw = tuple((in_dhw[i] + 2 * p[i] - k[i]) // s[i] + 1 for i in range(3))
    if min(out_dhw) <= 0:
        raise RuntimeError("conv: kernel %s does not fit input %s (padding %s)" % (k, in_dhw, p))
    geom = (Cin, Cout, k, s, p, out_dhw, bool(transposed), act, float(slope))
    in_act = (x.fused_act or x.fused_bn) if claim_act_grad else None
    out_act = {"claimed": False, "act": act, "slope": float(slope)} if act != _lib.ACT_NONE else None
    out = _Conv.apply(x.t, weight, bias, geom, stats, in_act, out_act, bias_token, in_bias_token)
    return ClTensor(out, Cout, nsp, out_act)


# ---------------------------------------------------------------------------------------------------------
# fp8 operands (OCP e4m3fn, per-tensor current scaling; include/vfdgan_hip.h "fp8 operands")
# ---------------------------------------------------------------------------------------------------------
def cpad16(c):
    return (c + 15) & ~15


def quantize_fp8(t, C):
    """Channels-last block `t` [..., CPAD(C)] (bf16 / f32) -> (uint8 tensor [..., CPAD16(C)] of e4m3 bytes, device scale):
    q = e4m3(t * scale), scale = 448 / max|t| taken from the tensor itself."""
    t = t.contiguous()
    rows = t.numel() // t.shape[-1]
    lib = load()
    amax = torch.empty(1, dtype=torch.float32, device=t.device)
    scale = torch.empty(1, dtype=torch.float32, device=t.device)
    q = torch.empty(tuple(t.shape[:-1]) + (cpad16(C),), dtype=torch.uint8, device=t.device)
    check(lib.vfd_amax(dtype_code(t.dtype), t.data_ptr(), rows, C, amax.data_ptr(), stream()), "amax")
    check(lib.vfd_quantize_fp8(dtype_code(t.dtype), t.data_ptr(), q.data_ptr(), rows, C, amax.data_ptr(), scale.data_ptr(), stream()),
          "quantize_fp8")
    return q, scale


def pack_filter_fp8(weight, transpose_ab, A, B, T):
    """K-major e4m3 copy of a float32 filter [A][B][T] (vfd_pack_filter's index map, channels padded to 16) and its scale."""
    w = weight.detach().contiguous().float()
    R, Cc = (B, A) if transpose_ab else (A, B)
    out = torch.empty((R, T, cpad16(Cc)), dtype=torch.uint8, device=w.device)
    amax = torch.empty(1, dtype=torch.float32, device=w.device)
    scale = torch.empty(1, dtype=torch.float32, device=w.device)
    check(load().vfd_pack_filter_fp8(w.data_ptr(), out.data_ptr(), A, B, T, int(transpose_ab), amax.data_ptr(), scale.data_ptr(), stream()),
          "pack_filter_fp8")
    return out, scale


def _quantized(t, C):
    """e4m3 copy of a channels-last block, made once per tensor (a tensor with several consumers — netD's two forwards of the
    same features, forward input re-used by nothing else — is quantised once)."""
    hit = getattr(t, "_vfd_q", None)
    if hit is not None and hit[2] == t._version:
        return hit[0], hit[1]
    q, scale = quantize_fp8(t, C)
    try:
        t._vfd_q = (q, scale, t._version)
    except AttributeError:
        pass
    return q, scale


def _packed_filter_fp8(weight, transpose_ab, A, B, T):
    """Cached e4m3 K-major copy of a filter parameter + its scale (same freshness tag as _packed_filter; re-packed lazily:
    amax + pack launches per filter and step)."""
    cache = weight.__dict__.setdefault("_vfd_packed", {})
    key = ("fp8", transpose_ab)
    own = getattr(weight, "_vfd_epoch", None)
    tag = (weight._version, _WEIGHT_EPOCH[0], own[0] if own is not None else 0, weight.data_ptr())
    hit = cache.get(key)
    if hit is not None and hit[0] == tag:
        return hit[1], hit[2]
    w = weight.detach()
    if not w.is_contiguous() or w.dtype != torch.float32:
        w = w.contiguous().float()
    R, Cc = (B, A) if transpose_ab else (A, B)
    shape = (R, T, cpad16(Cc))
    if hit is not None and tuple(hit[1].shape) == shape:
        out, amax, scale = hit[1], hit[3], hit[2]
    else:
        out = torch.empty(shape, dtype=torch.uint8, device=w.device)
        amax = torch.empty(1, dtype=torch.float32, device=w.device)
        scale = torch.empty(1, dtype=torch.float32, device=w.device)
    check(load().vfd_pack_filter_fp8(w.data_ptr(), out.data_ptr(), A, B, T, int(transpose_ab), amax.data_ptr(), scale.data_ptr(), stream()),
          "pack_filter_fp8")
    cache[key] = [tag, out, scale, amax]
    return out, scale


def conv_fp8(xq, xscale, wq, wscale, bias, N, in_dhw, Cin, out_dhw, Cout, k, s, p, transposed, act=_lib.ACT_NONE, slope=0.0, stats=None):
    """y (bf16, [N, *out_dhw, CPAD(Cout)]) = act(conv(xq, wq) / (xscale * wscale) + bias) on the fp8 MFMA path."""
    desc = _make_desc(N, in_dhw, Cin, out_dhw, Cout, k, s, p, transposed, torch.bfloat16, act, slope)
    desc.dtype = _lib.FP8
    y = torch.empty((N,) + tuple(out_dhw) + (cpad(Cout),), dtype=torch.bfloat16, device=xq.device)
    timer = _TIMER[0]
    if timer is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    check(load().vfd_conv_forward_fp8(ctypes.byref(desc), xq.data_ptr(), xscale.data_ptr(), wq.data_ptr(), wscale.data_ptr(), ptr(bias),
                                      y.data_ptr(), *_stats_arg(stats), stream()), "conv_forward_fp8")
    if timer is not None:
        e1.record()
        timer.records.append((_conv_kernel_name(desc, stats), _conv_flops(desc), e0, e1, _geom_str(desc, stats)))
    return y


# ---------------------------------------------------------------------------------------------------------
# BatchNorm (train mode) + activation
# ---------------------------------------------------------------------------------------------------------
# ---- one forward, two backward passes (ganomaly: netd(fake) serves backward_g and backward_d) -------------------------------
# A parameter marked `_vfd_frozen` receives no gradient work in a backward pass although it required grad when the graph
# was built (the pass that runs with the net "frozen"); inputs whose data_ptr is in _SKIP_INPUT_GRAD get no data gradient;
# sum pools created inside collect_pools() are listed so that they can be zeroed between the two passes.
_SKIP_INPUT_GRAD = set()
_POOL_SINK = [None]


def _frozen(prm):
    return prm is not None and getattr(prm, "_vfd_frozen", False)


class collect_pools:
    def __enter__(self):
        self.prev, self.pools = _POOL_SINK[0], []
        _POOL_SINK[0] = self.pools
        return self.pools

    def __exit__(self, *exc):
        _POOL_SINK[0] = self.prev
        return False


def register_pool(t):
    if _POOL_SINK[0] is not None:
        _POOL_SINK[0].append(t)


_LAST_BN_STATS = [None]      # (mean, rstd, rows) of the most recent training-mode bn_act (nn._BatchNormMixin keeps it on request)


def bn_running_update(mean, rstd, rows, C, running_mean, running_var, eps, momentum, num_batches_tracked):
    """The running-statistics side effect of one more training-mode forward with the same batch statistics."""
    check(load().vfd_bn_running_update(mean.data_ptr(), rstd.data_ptr(), rows, C, float(eps), float(momentum), ptr(running_mean),
                                       ptr(running_var), ptr(num_batches_tracked), stream()), "bn_running_update")


class _BnAct(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, gamma, beta, C, running_mean, running_var, eps, momentum, act, slope, sums, nbt, token, conv_bias=None,
                bias_token=None):
        lib = load()
        x = x.contiguous()
        rows = x.numel() // x.shape[-1]
        dev = x.device
        mean = torch.empty(C, dtype=torch.float32, device=dev)
        rstd = torch.empty(C, dtype=torch.float32, device=dev)
        dtc = dtype_code(x.dtype)
        y = torch.empty_like(x)
        g_, b_ = (gamma.detach() if gamma is not None else None), (beta.detach() if beta is not None else None)
        if sums is not None:
            # statistics handed over by the producing conv's epilogue: fold + normalise + activate in one launch
            if sums.numel() < STATS_REPLICAS * 2 * cpad(C) or sums.dtype != torch.float64:
                raise RuntimeError("bn_act: statistics buffer too small or not float64 (use functional.new_stats_buffer)")
            check(lib.vfd_bn_act_forward_sums(dtc, x.data_ptr(), y.data_ptr(), rows, C, sums.data_ptr(), eps, momentum,
                                              mean.data_ptr(), rstd.data_ptr(), ptr(running_mean), ptr(running_var), ptr(nbt),
                                              ptr(g_), ptr(b_), act, slope, stream()), "bn_act_forward_sums")
        else:
            ws = _workspace(lib.vfd_bn_workspace(rows, C), dev)
            check(lib.vfd_bn_stats(dtc, x.data_ptr(), rows, C, eps, momentum, mean.data_ptr(), rstd.data_ptr(),
                                   ptr(running_mean), ptr(running_var), ptr(nbt), ws.data_ptr(), stream()), "bn_stats")
            check(lib.vfd_bn_act_forward(dtc, x.data_ptr(), y.data_ptr(), rows, C, mean.data_ptr(), rstd.data_ptr(), ptr(g_),
                                         ptr(b_), act, slope, stream()), "bn_act_forward")
        ctx.meta = (rows, C, act, slope)
        _LAST_BN_STATS[0] = (mean, rstd, rows)
        ctx.token = token
        ctx.conv_bias = None
        if (token is not None and bias_token is not None and conv_bias is not None and conv_bias.requires_grad
                and _direct_grad(conv_bias) is not None):
            # x is conv(..) + conv_bias: the bias gradient is the column sum of THIS backward's dx, which the apply pass has
            # in registers; the conv then skips its own column-sum pass over dx
            bias_token["taken"] = True
            ctx.conv_bias = bias_token["rep"]      # zeroed [STATS_REPLICAS][CPAD(C)] rows; folded by the conv's vfd_wgrad_reduce_bias
        if token is not None:
            # what the consumer conv's data-gradient epilogue needs to take over the reduce pass of this backward
            token.update(x=x, mean=mean, rstd=rstd, gamma=g_, beta=b_, act=act, slope=slope, claimed=False)
        ctx.save_for_backward(x, gamma, beta, mean, rstd)
        return y

    @staticmethod
    def backward(ctx, gy):
        lib = load()
        rows, C, act, slope = ctx.meta
        x, gamma, beta, mean, rstd = ctx.saved_tensors
        gy = gy.contiguous()
        dev = x.device
        dx = torch.empty_like(x)
        dgamma = torch.empty(C, dtype=torch.float32, device=dev)
        dbeta = torch.empty(C, dtype=torch.float32, device=dev)
        ws = _workspace(lib.vfd_bn_workspace(rows, C), dev)
        g_, b_ = (gamma.detach() if gamma is not None else None), (beta.detach() if beta is not None else None)
        frozen = _frozen(gamma) or _frozen(beta)
        dg_acc = _direct_grad(gamma) if (ctx.needs_input_grad[1] and not frozen) else None     # a frozen net's gradients stay untouched
        db_acc = _direct_grad(beta) if (ctx.needs_input_grad[2] and not frozen) else None
        nret = (None,) * 12
        cs_acc = ctx.conv_bias
        if ctx.token is not None and ctx.token["claimed"]:
            # gy is already g = dy * act'(z) and the sums of g, g*xhat sit in the token's buffer (written by the consumer's
            # data gradient): only the apply pass is left
            check(lib.vfd_bn_backward_apply_sums(dtype_code(x.dtype), x.data_ptr(), gy.data_ptr(), dx.data_ptr(), rows, C,
                                                 mean.data_ptr(), rstd.data_ptr(), ptr(g_), ctx.token["sums"].data_ptr(),
                                                 dgamma.data_ptr(), dbeta.data_ptr(), ptr(dg_acc), ptr(db_acc), ptr(cs_acc),
                                                 stream()),
                  "bn_backward_apply_sums")
            return (dx, dgamma if (gamma is not None and dg_acc is None and not frozen) else None,
                    dbeta if (beta is not None and db_acc is None and not frozen) else None) + nret
        if ctx.token is not None:
            # unclaimed, but a zeroed sums buffer is at hand: reduce into it with atomics, fold in the apply pass (two launches)
            check(lib.vfd_bn_act_backward_sums(dtype_code(x.dtype), x.data_ptr(), gy.data_ptr(), dx.data_ptr(), rows, C,
                                               mean.data_ptr(), rstd.data_ptr(), ptr(g_), ptr(b_), act, slope,
                                               ctx.token["sums"].data_ptr(), dgamma.data_ptr(), dbeta.data_ptr(), ptr(dg_acc),
                                               ptr(db_acc), ptr(cs_acc), stream()), "bn_act_backward_sums")
            return (dx, dgamma if (gamma is not None and dg_acc is None and not frozen) else None,
                    dbeta if (beta is not None and db_acc is None and not frozen) else None) + nret
        check(lib.vfd_bn_act_backward(dtype_code(x.dtype), x.data_ptr(), gy.data_ptr(), dx.data_ptr(), rows, C,
                                      mean.data_ptr(), rstd.data_ptr(), ptr(g_), ptr(b_), act, slope, dgamma.data_ptr(),
                                      dbeta.data_ptr(), ptr(dg_acc), ptr(db_acc), ws.data_ptr(), stream()), "bn_act_backward")
        return (dx, dgamma if (gamma is not None and dg_acc is None and not frozen) else None,
                dbeta if (beta is not None and db_acc is None and not frozen) else None) + nret


class _BnActPool(torch.autograd.Function):
    """BatchNorm (statistics from the producing conv's epilogue sums) -> activation -> AvgPool3d(2), the activation having no
    other consumer: vfd_bn_act_pool_forward_sums / vfd_bn_act_pool_backward_sums."""

    @staticmethod
    def forward(ctx, x, gamma, beta, C, running_mean, running_var, eps, momentum, act, slope, sums, nbt, bwd_sums, conv_bias,
                bias_token, pool, keep_full):
        lib = load()
        x = x.contiguous()
        N, D, H, W, Cp = x.shape
        pd, ph, pw = pool
        dev = x.device
        mean = torch.empty(C, dtype=torch.float32, device=dev)
        rstd = torch.empty(C, dtype=torch.float32, device=dev)
        y = torch.empty((N, D // pd, H // ph, W // pw, Cp), dtype=x.dtype, device=dev)
        yfull = torch.empty_like(x) if keep_full else None      # the activation's second, full-resolution consumer (U-Net skip)
        g_, b_ = (gamma.detach() if gamma is not None else None), (beta.detach() if beta is not None else None)
        check(lib.vfd_bn_act_pool_forward_sums(dtype_code(x.dtype), x.data_ptr(), y.data_ptr(), N, D, H, W, pd, ph, pw, C, sums.data_ptr(), eps,
                                               momentum, mean.data_ptr(), rstd.data_ptr(), ptr(running_mean), ptr(running_var),
                                               ptr(nbt), ptr(g_), ptr(b_), act, slope, ptr(yfull), stream()), "bn_act_pool_forward_sums")
        _LAST_BN_STATS[0] = (mean, rstd, N * D * H * W)
        ctx.meta = (N, D, H, W, C, act, slope, pool)
        ctx.keep_full = keep_full
        ctx.bwd_sums = bwd_sums
        ctx.cs_rep = None
        if (bias_token is not None and conv_bias is not None and conv_bias.requires_grad and _direct_grad(conv_bias) is not None):
            bias_token["taken"] = True
            ctx.cs_rep = bias_token["rep"]
        ctx.save_for_backward(x, gamma, beta, mean, rstd)
        return (y, yfull) if keep_full else y

    @staticmethod
    def backward(ctx, gp, gfull=None):
        lib = load()
        N, D, H, W, C, act, slope, (pd, ph, pw) = ctx.meta
        x, gamma, beta, mean, rstd = ctx.saved_tensors
        if gp is None:          # only the full-resolution output was used
            gp = zeros((N, D // pd, H // ph, W // pw, x.shape[-1]), x.dtype, x.device)
        gp = gp.contiguous()
        gfull = gfull.contiguous() if gfull is not None else None
        dev = x.device
        dx = torch.empty_like(x)
        dgamma = torch.empty(C, dtype=torch.float32, device=dev)
        dbeta = torch.empty(C, dtype=torch.float32, device=dev)
        g_, b_ = (gamma.detach() if gamma is not None else None), (beta.detach() if beta is not None else None)
        frozen = _frozen(gamma) or _frozen(beta)
        dg_acc = _direct_grad(gamma) if (ctx.needs_input_grad[1] and not frozen) else None
        db_acc = _direct_grad(beta) if (ctx.needs_input_grad[2] and not frozen) else None
        check(lib.vfd_bn_act_pool_backward_sums(dtype_code(x.dtype), x.data_ptr(), gp.data_ptr(), dx.data_ptr(), N, D, H, W, pd, ph, pw, C,
                                                mean.data_ptr(), rstd.data_ptr(), ptr(g_), ptr(b_), act, slope,
                                                ctx.bwd_sums.data_ptr(), dgamma.data_ptr(), dbeta.data_ptr(), ptr(dg_acc), ptr(db_acc),
                                                ptr(ctx.cs_rep), ptr(gfull), stream()), "bn_act_pool_backward_sums")
        return (dx, dgamma if (gamma is not None and dg_acc is None and not frozen) else None,
                dbeta if (beta is not None and db_acc is None and not frozen) else None) + (None,) * 14


def pool_fusable(kernel, stride, padding, in_dhw):
    """AvgPool3d that _BnActPool can absorb: kernel == stride, every extent 1 or 2 (not all 1), no padding, input a multiple."""
    k, s, p = _triple(kernel, 3, 1), _triple(stride if stride is not None else kernel, 3, 1), _triple(padding, 3, 0)
    ok = k == s and p == (0, 0, 0) and all(v in (1, 2) for v in k) and k != (1, 1, 1) and all(d % v == 0 for d, v in zip(in_dhw, k))
    return k if ok else None


def bn_act_pool(x, gamma, beta, running_mean, running_var, eps, momentum, act, slope, sums, num_batches_tracked, bwd_sums,
                conv_bias=None, bias_token=None, pool=(2, 2, 2), keep_full=False):
    """bn_act + AvgPool3d fused (see _BnActPool): x a 3-D channels-last block, `sums` the producing conv's epilogue statistics,
    `bwd_sums` a zeroed sums buffer for the backward, `pool` = kernel = stride (pool_fusable).  keep_full: the activation has a
    full-resolution consumer too; returns (pooled, full) and the backward sums the two incoming gradients itself."""
    out = _BnActPool.apply(x.t, gamma, beta, x.C, running_mean, running_var, float(eps), float(momentum), int(act), float(slope),
                           sums, num_batches_tracked, bwd_sums, conv_bias, bias_token, tuple(pool), bool(keep_full))
    if keep_full:
        return ClTensor(out[0], x.C, x.nsp), ClTensor(out[1], x.C, x.nsp)
    return ClTensor(out, x.C, x.nsp)


def bn_act(x, gamma, beta, running_mean, running_var, eps=1e-5, momentum=0.1, act=_lib.ACT_NONE, slope=0.0,
           sums=None, num_batches_tracked=None, bwd_sums=None, conv_bias=None, bias_token=None):
    """Training-mode batch normalisation over all rows of `x` followed by `act`; updates the running statistics
    in place (momentum rule, unbiased variance) exactly like torch.nn.BatchNormNd.train(); `num_batches_tracked`
    (int64 device scalar) is incremented by the statistics kernel."""
    if num_batches_tracked is not None and (num_batches_tracked.dtype != torch.int64 or not num_batches_tracked.is_cuda):
        raise TypeError("num_batches_tracked must be an int64 device tensor")
    token = {"sums": bwd_sums} if bwd_sums is not None else None
    y = _BnAct.apply(x.t, gamma, beta, x.C, running_mean, running_var, float(eps), float(momentum), int(act),
                     float(slope), sums, num_batches_tracked, token, conv_bias, bias_token)
    out = ClTensor(y, x.C, x.nsp)
    out.fused_bn = token      # `bwd_sums`: zeroed [STATS_REPLICAS][2][CPAD(C)] buffer; y must have exactly ONE consumer conv
    return out


def bn_act_eval(x, gamma, beta, running_mean, running_var, eps=1e-5, act=_lib.ACT_NONE, slope=0.0):
    """Evaluation-mode batch normalisation (torch.nn.BatchNormNd.eval(): the running statistics, no update) followed by `act`,
    through the same normalise+activate kernel as training (vfd_bn_act_forward) with mean = running_mean and
    rstd = 1/sqrt(running_var + eps).  Inference only — the reference evaluates under torch.no_grad()
    (models/anogan.py:146-158); a gradient through frozen statistics is not provided."""
    if torch.is_grad_enabled() and x.t.requires_grad:
        raise NotImplementedError("eval-mode BatchNorm is forward-only (run it under torch.no_grad(), as the reference's test() does)")
    if running_mean is None or running_var is None:
        raise RuntimeError("eval-mode BatchNorm needs running statistics (track_running_stats=True)")
    xt = x.t.contiguous()
    rows = xt.numel() // xt.shape[-1]
    rstd = torch.rsqrt(running_var.float() + eps)              # C floats: parameter preparation, not tensor arithmetic
    y = torch.empty_like(xt)
    g_, b_ = (gamma.detach() if gamma is not None else None), (beta.detach() if beta is not None else None)
    check(load().vfd_bn_act_forward(dtype_code(xt.dtype), xt.data_ptr(), y.data_ptr(), rows, x.C, running_mean.data_ptr(),
                                    rstd.data_ptr(), ptr(g_), ptr(b_), int(act), float(slope), stream()), "bn_act_forward(eval)")
    return ClTensor(y, x.C, x.nsp)


# ---------------------------------------------------------------------------------------------------------
# element-wise activation
# ---------------------------------------------------------------------------------------------------------
class _Act(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, C, act, slope):
        x = x.contiguous()
        rows = x.numel() // x.shape[-1]
        y = torch.empty_like(x)
        check(load().vfd_act_forward(dtype_code(x.dtype), x.data_ptr(), y.data_ptr(), rows, C, act, slope, stream()),
              "act_forward")
        ctx.meta = (rows, C, act, slope)
        ctx.save_for_backward(y)
        return y

    @staticmethod
    def backward(ctx, gy):
        rows, C, act, slope = ctx.meta
        (y,) = ctx.saved_tensors
        gy = gy.contiguous()
        dx = torch.empty_like(gy)
        check(load().vfd_act_backward(dtype_code(y.dtype), y.data_ptr(), gy.data_ptr(), dx.data_ptr(), rows, C, act, slope,
                                      stream()), "act_backward")
        return dx, None, None, None


def activation(x, act, slope=0.0):
    return ClTensor(_Act.apply(x.t, x.C, int(act), float(slope)), x.C, x.nsp)


# ---------------------------------------------------------------------------------------------------------
# pooling / resampling / channel plumbing / dropout
# ---------------------------------------------------------------------------------------------------------
class _AvgPool(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, C, k):
        x = x.contiguous()
        N, D, H, W, Cp = x.shape
        y = torch.empty((N, D // k[0], H // k[1], W // k[2], Cp), dtype=x.dtype, device=x.device)
        check(load().vfd_avgpool_forward(dtype_code(x.dtype), x.data_ptr(), y.data_ptr(), N, D, H, W, C, k[0], k[1], k[2],
                                         stream()), "avgpool_forward")
        ctx.meta = (N, D, H, W, C, k, tuple(x.shape))
        return y

    @staticmethod
    def backward(ctx, gy):
        N, D, H, W, C, k, shape = ctx.meta
        gy = gy.contiguous()
        dx = torch.empty(shape, dtype=gy.dtype, device=gy.device)
        check(load().vfd_avgpool_backward(dtype_code(gy.dtype), gy.data_ptr(), dx.data_ptr(), N, D, H, W, C, k[0], k[1],
                                          k[2], stream()), "avgpool_backward")
        return dx, None, None


def avg_pool(x, kernel):
    k = _triple(kernel, x.nsp, 1)
    return ClTensor(_AvgPool.apply(x.t, x.C, k), x.C, x.nsp)


class _Upsample2x(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, C):
        x = x.contiguous()
        N, D, H, W, Cp = x.shape
        y = torch.empty((N, 2 * D, 2 * H, 2 * W, Cp), dtype=x.dtype, device=x.device)
        check(load().vfd_upsample2x_forward(dtype_code(x.dtype), x.data_ptr(), y.data_ptr(), N, D, H, W, C, stream()),
              "upsample2x_forward")
        ctx.meta = (N, D, H, W, C, tuple(x.shape))
        return y

    @staticmethod
    def backward(ctx, gy):
        N, D, H, W, C, shape = ctx.meta
        gy = gy.contiguous()
        dx = torch.empty(shape, dtype=gy.dtype, device=gy.device)
        check(load().vfd_upsample2x_backward(dtype_code(gy.dtype), gy.data_ptr(), dx.data_ptr(), N, D, H, W, C, stream()),
              "upsample2x_backward")
        return dx, None


def upsample_trilinear2x(x):
    if x.nsp != 3:
        raise RuntimeError("upsample_trilinear2x expects an (N,C,D,H,W) block")
    return ClTensor(_Upsample2x.apply(x.t, x.C), x.C, x.nsp)


class _UpsampleN(torch.autograd.Function):
    """nn.Upsample(scale_factor=(fd,fh,fw), trilinear, align_corners=True) with factors 1 or 2 (vfd_upsample_*)."""

    @staticmethod
    def forward(ctx, x, C, f):
        x = x.contiguous()
        N, D, H, W, Cp = x.shape
        y = torch.empty((N, f[0] * D, f[1] * H, f[2] * W, Cp), dtype=x.dtype, device=x.device)
        check(load().vfd_upsample_forward(dtype_code(x.dtype), x.data_ptr(), y.data_ptr(), N, D, H, W, C, f[0], f[1], f[2], stream()),
              "upsample_forward")
        ctx.meta = (N, D, H, W, C, f, tuple(x.shape))
        return y

    @staticmethod
    def backward(ctx, gy):
        N, D, H, W, C, f, shape = ctx.meta
        gy = gy.contiguous()
        dx = torch.empty(shape, dtype=gy.dtype, device=gy.device)
        check(load().vfd_upsample_backward(dtype_code(gy.dtype), gy.data_ptr(), dx.data_ptr(), N, D, H, W, C, f[0], f[1], f[2], stream()),
              "upsample_backward")
        return dx, None, None


def upsample_trilinear(x, factors):
    """Upsample(scale_factor=factors, mode='trilinear', align_corners=True), every factor 1 or 2 (models/xception.py:84)."""
    f = tuple(int(v) for v in _triple(factors, 3, 1))
    if x.nsp != 3 or any(v not in (1, 2) for v in f):
        raise NotImplementedError("upsample_trilinear: (N,C,D,H,W) blocks and scale factors 1 or 2 only, got %s" % (f,))
    if f == (2, 2, 2):
        return upsample_trilinear2x(x)
    return ClTensor(_UpsampleN.apply(x.t, x.C, f), x.C, x.nsp)


class _MaxPool(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, C, k, s, p):
        x = x.contiguous()
        N, D, H, W, Cp = x.shape
        out = tuple((v + 2 * pp - kk) // ss + 1 for v, kk, ss, pp in zip((D, H, W), k, s, p))
        y = torch.empty((N,) + out + (Cp,), dtype=x.dtype, device=x.device)
        idx = torch.empty((N,) + out + (Cp,), dtype=torch.uint8, device=x.device)
        check(load().vfd_maxpool_forward(dtype_code(x.dtype), x.data_ptr(), y.data_ptr(), idx.data_ptr(), N, D, H, W, C, *k, *s, *p, stream()),
              "maxpool_forward")
        ctx.meta = (N, D, H, W, C, k, s, p, tuple(x.shape))
        ctx.save_for_backward(idx)
        return y

    @staticmethod
    def backward(ctx, gy):
        N, D, H, W, C, k, s, p, shape = ctx.meta
        (idx,) = ctx.saved_tensors
        gy = gy.contiguous()
        dx = torch.empty(shape, dtype=gy.dtype, device=gy.device)
        check(load().vfd_maxpool_backward(dtype_code(gy.dtype), gy.data_ptr(), idx.data_ptr(), dx.data_ptr(), N, D, H, W, C, *k, *s, *p,
                                          stream()), "maxpool_backward")
        return dx, None, None, None, None


def max_pool(x, kernel, stride=None, padding=0):
    """nn.MaxPool3d(kernel, stride, padding) on an (N,C,D,H,W) block (models/xception.py:57)."""
    k = _triple(kernel, 3, 1)
    s = _triple(stride if stride is not None else kernel, 3, 1)
    p = _triple(padding, 3, 0)
    return ClTensor(_MaxPool.apply(x.t, x.C, tuple(k), tuple(s), tuple(p)), x.C, x.nsp)


class _Add(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, b):
        a, b = a.contiguous(), b.contiguous()
        y = torch.empty_like(a)
        check(load().vfd_add(dtype_code(a.dtype), a.data_ptr(), b.data_ptr(), 0, y.data_ptr(), a.numel(), stream()), "add")
        return y

    @staticmethod
    def backward(ctx, g):
        return g, g


def add(a, b):
    """a + b of two equally shaped blocks (the residual join `x += skip` of models/xception.py:68)."""
    if tuple(a.t.shape) != tuple(b.t.shape) or a.C != b.C:
        raise RuntimeError("add: shapes differ: %s vs %s" % (a.shape, b.shape))
    return ClTensor(_Add.apply(a.t, b.t), a.C, a.nsp)


class _Fanout(torch.autograd.Function):
    """n aliases of one tensor whose gradients are summed by ONE launch (float32 sum, one rounding) instead of autograd's
    pairwise adds in the storage dtype: a tensor with several consumers (ganomaly's fake: L1 loss, encoder2, netD)."""

    @staticmethod
    def forward(ctx, x, n):
        ctx.n = n
        return tuple(x.view_as(x) for _ in range(n))

    @staticmethod
    def backward(ctx, *gs):
        gs = [g.contiguous() for g in gs if g is not None]
        if not gs:
            return None, None
        if len(gs) == 1:
            return gs[0], None
        acc = gs[0]
        lib = load()
        i = 1
        while i < len(gs):
            c = gs[i + 1] if i + 1 < len(gs) else None
            y = torch.empty_like(acc)
            check(lib.vfd_add(dtype_code(acc.dtype), acc.data_ptr(), gs[i].data_ptr(), ptr(c), y.data_ptr(), acc.numel(), stream()), "add")
            acc = y
            i += 2
        return acc, None


def fanout(x, n):
    """`n` handles of the block x for n consumers; see _Fanout."""
    return tuple(ClTensor(t, x.C, x.nsp) for t in _Fanout.apply(x.t, n))


class _UpsampleCat(torch.autograd.Function):
    """torch.cat([Upsample(scale 2, trilinear, align_corners)(x), skip], dim=1) without the up-sampled intermediate."""

    @staticmethod
    def forward(ctx, x, skip, Ca, Cb):
        x, skip = x.contiguous(), skip.contiguous()
        N, D, H, W, _ = x.shape
        y = torch.empty((N, 2 * D, 2 * H, 2 * W, Ca + cpad(Cb)), dtype=x.dtype, device=x.device)
        check(load().vfd_upsample2x_cat_forward(dtype_code(x.dtype), x.data_ptr(), skip.data_ptr(), y.data_ptr(), N, D, H, W, Ca, Cb,
                                                stream()), "upsample2x_cat_forward")
        ctx.meta = (N, D, H, W, Ca, Cb, tuple(x.shape), tuple(skip.shape))
        return y

    @staticmethod
    def backward(ctx, g):
        N, D, H, W, Ca, Cb, sx, ss = ctx.meta
        g = g.contiguous()
        dx = torch.empty(sx, dtype=g.dtype, device=g.device)
        ds = torch.empty(ss, dtype=g.dtype, device=g.device)
        check(load().vfd_upsample2x_cat_backward(dtype_code(g.dtype), g.data_ptr(), dx.data_ptr(), ds.data_ptr(), N, D, H, W, Ca, Cb,
                                                 stream()), "upsample2x_cat_backward")
        return dx, ds, None, None


def upsample_cat(x, skip):
    """cat_channels(upsample_trilinear2x(x), skip) in one pass (x's channel count a multiple of 8, else the two-pass form)."""
    if x.nsp != 3 or skip.nsp != 3:
        raise RuntimeError("upsample_cat expects (N,C,D,H,W) blocks")
    if x.C % 8 != 0 or x.t.shape[-1] != x.C:
        return cat_channels(upsample_trilinear2x(x), skip)
    if tuple(skip.t.shape[:4]) != (x.t.shape[0], 2 * x.t.shape[1], 2 * x.t.shape[2], 2 * x.t.shape[3]):
        raise RuntimeError("upsample_cat: skip %s is not twice x %s" % (skip.shape, x.shape))
    return ClTensor(_UpsampleCat.apply(x.t, skip.t, x.C, skip.C), x.C + skip.C, 3)


class _Concat(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, b, Ca, Cb):
        a, b = a.contiguous(), b.contiguous()
        rows = a.numel() // a.shape[-1]
        out = torch.empty(tuple(a.shape[:-1]) + (cpad(Ca + Cb),), dtype=a.dtype, device=a.device)
        check(load().vfd_concat_channels(dtype_code(a.dtype), a.data_ptr(), b.data_ptr(), out.data_ptr(), rows, Ca, Cb,
                                         stream()), "concat")
        ctx.meta = (rows, Ca, Cb, tuple(a.shape), tuple(b.shape))
        return out

    @staticmethod
    def backward(ctx, g):
        rows, Ca, Cb, sa, sb = ctx.meta
        g = g.contiguous()
        ga = torch.empty(sa, dtype=g.dtype, device=g.device)
        gb = torch.empty(sb, dtype=g.dtype, device=g.device)
        check(load().vfd_split_channels(dtype_code(g.dtype), g.data_ptr(), ga.data_ptr(), gb.data_ptr(), rows, Ca, Cb,
                                        stream()), "split")
        return ga, gb, None, None


def cat_channels(a, b):
    """torch.cat([a, b], dim=1) on ClTensors."""
    if tuple(a.t.shape[:-1]) != tuple(b.t.shape[:-1]):
        raise RuntimeError("cat: spatial shapes differ: %s vs %s" % (a.shape, b.shape))
    return ClTensor(_Concat.apply(a.t, b.t, a.C, b.C), a.C + b.C, a.nsp)


def gray2rgb(x):
    """lib/utils.py:91-92 of the reference: cat([v, v, v], dim=1) of a 1-channel block (no gradient needed:
    the reference only applies it to detached tensors, models/mygannet.py:279-280)."""
    if x.C != 1:
        raise RuntimeError("gray2rgb expects 1 channel")
    src = x.t.detach().contiguous()
    rows = src.numel() // src.shape[-1]
    out = torch.empty_like(src)  # CPAD(3) == CPAD(1) == 8
    check(load().vfd_broadcast_channel(dtype_code(src.dtype), src.data_ptr(), out.data_ptr(), rows, 3, stream()), "broadcast")
    return ClTensor(out, 3, x.nsp)


class _Dropout(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, p, seed, mask_in, step_dev):
        x = x.contiguous()
        n = x.numel()
        y = torch.empty_like(x)
        mask = torch.empty(n, dtype=torch.uint8, device=x.device)
        check(load().vfd_dropout_forward(dtype_code(x.dtype), x.data_ptr(), y.data_ptr(), mask.data_ptr(), ptr(mask_in), n, p,
                                         seed, ptr(step_dev), stream()), "dropout_forward")
        ctx.p = p
        ctx.save_for_backward(mask)
        return y

    @staticmethod
    def backward(ctx, gy):
        (mask,) = ctx.saved_tensors
        gy = gy.contiguous()
        dx = torch.empty_like(gy)
        check(load().vfd_dropout_backward(dtype_code(gy.dtype), gy.data_ptr(), dx.data_ptr(), mask.data_ptr(), gy.numel(),
                                          ctx.p, stream()), "dropout_backward")
        return dx, None, None, None, None


_DROPOUT_STATE = {"seed": 0x5EED, "calls": 0, "mask_provider": None, "step_dev": None}


def dropout_manual_seed(seed):
    _DROPOUT_STATE["seed"] = int(seed)
    _DROPOUT_STATE["calls"] = 0


def dropout_begin_step(device):
    """Call once at the start of every training step: resets the per-step call index (so a step issues the same
    host-side seeds every time — required for hipGraph replay) and advances the DEVICE step counter that the kernels
    mix into the seed (so the masks still change from step to step, also under replay)."""
    st = _DROPOUT_STATE
    if st["step_dev"] is None or st["step_dev"].device != device:
        st["step_dev"] = torch.zeros(1, dtype=torch.int64, device=device)
    st["step_dev"].add_(1)
    st["calls"] = 0


def dropout_state():
    """(seed, device step counter) for checkpoints: with it a resumed run draws the masks the original would have."""
    st = _DROPOUT_STATE
    return {"seed": int(st["seed"]), "step": int(st["step_dev"].item()) if st["step_dev"] is not None else 0}


def set_dropout_state(state, device):
    st = _DROPOUT_STATE
    st["seed"] = int(state["seed"])
    st["step_dev"] = torch.full((1,), int(state["step"]), dtype=torch.int64, device=device)
    st["calls"] = 0


def set_dropout_mask_provider(fn):
    """Parity hook: fn(logical_shape, p, call_index) -> bool/uint8 keep-mask (torch tensor, reference layout
    (N,C,D,H,W)) or None.  Lets tests impose the masks the CPU oracle used (torch's CPU and device RNG streams
    differ)."""
    _DROPOUT_STATE["mask_provider"] = fn
    _DROPOUT_STATE["calls"] = 0


def dropout(x, p, training=True):
    if not training or p == 0.0:
        return x
    st = _DROPOUT_STATE
    idx = st["calls"]
    st["calls"] += 1
    mask_in = None
    if st["mask_provider"] is not None:
        m = st["mask_provider"](x.shape, p, idx)
        if m is not None:
            # reference layout -> channels-last uint8 with padded channels (pad lanes irrelevant: inputs there are 0)
            m = m.to(device=x.t.device, dtype=torch.float32)
            mcl = to_cl(m, dtype=torch.float32).t
            mask_in = (mcl != 0).to(torch.uint8).contiguous().view(-1)
    seed = (st["seed"] * 0x9E3779B1 + idx * 0x85EBCA6B) & 0xFFFFFFFFFFFFFFFF
    return ClTensor(_Dropout.apply(x.t, float(p), seed, mask_in, st["step_dev"]), x.C, x.nsp)


# ---------------------------------------------------------------------------------------------------------
# losses
# ---------------------------------------------------------------------------------------------------------
class _Loss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, b, kind, C, bconst, pos_weight):
        lib = load()
        a = a.contiguous()
        if b is not None:
            b = b.contiguous()
        rows = a.numel() // a.shape[-1]
        loss = torch.empty((), dtype=torch.float32, device=a.device)
        ws = _workspace(lib.vfd_loss_workspace(rows, C), a.device)
        check(lib.vfd_loss_forward(kind, dtype_code(a.dtype), a.data_ptr(), ptr(b), bconst, loss.data_ptr(), rows, C,
                                   pos_weight, ws.data_ptr(), stream()), "loss_forward")
        ctx.meta = (kind, C, rows, bconst, pos_weight)
        ctx.save_for_backward(a, b)
        return loss

    @staticmethod
    def backward(ctx, gout):
        kind, C, rows, bconst, pos_weight = ctx.meta
        a, b = ctx.saved_tensors
        gout = gout.contiguous().float()
        need_a = ctx.needs_input_grad[0]
        need_b = b is not None and ctx.needs_input_grad[1]
        ga = torch.empty_like(a) if need_a else None
        gb = torch.empty_like(b) if need_b else None
        if need_a or need_b:
            check(load().vfd_loss_backward(kind, dtype_code(a.dtype), a.data_ptr(), ptr(b), bconst, gout.data_ptr(), ptr(ga),
                                           ptr(gb), rows, C, 1.0, pos_weight, stream()), "loss_backward")
        return ga, gb, None, None, None, None


def _loss(kind, a, b, pos_weight=2.0):
    if isinstance(b, ClTensor):
        if tuple(a.t.shape) != tuple(b.t.shape):
            raise RuntimeError("loss: shapes differ: %s vs %s" % (a.shape, b.shape))
        return _Loss.apply(a.t, b.t, kind, a.C, 0.0, float(pos_weight))
    return _Loss.apply(a.t, None, kind, a.C, float(b), float(pos_weight))


def l2_loss(a, b):
    """lib/utils.py:59-63 (size_average=True): mean((a-b)^2).  `b` is a ClTensor or a Python float."""
    return _loss(_lib.LOSS_L2, a, b)


def l1_loss(a, b):
    """nn.L1Loss() (models/ganomaly.py:438)."""
    return _loss(_lib.LOSS_L1, a, b)


def bce_loss(a, b):
    """nn.BCELoss() mean (models/mygannet.py:267); `b` may be a constant label (1.0 / 0.0)."""
    return _loss(_lib.LOSS_BCE, a, b)


def weighted_bce(a, b, pos_weight=2):
    """lib/utils.py:65-71."""
    return _loss(_lib.LOSS_WBCE, a, b, pos_weight if pos_weight is not None else 1.0)
