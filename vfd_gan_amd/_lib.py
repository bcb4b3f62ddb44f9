"""ctypes binding of libvfdgan_hip.so (C ABI: include/vfdgan_hip.h).

The product path has NO fallback: if the HIP library is missing or a call fails, this raises.  PyTorch is used
only for device memory (tensors -> raw pointers) and the current HIP stream.
"""
import ctypes
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libvfdgan_hip.so")

F32, BF16, FP8 = 0, 1, 2
ACT_NONE, ACT_LRELU, ACT_SIGMOID, ACT_TANH = 0, 1, 2, 3
LOSS_L2, LOSS_L1, LOSS_BCE, LOSS_WBCE = 0, 1, 2, 3

c_int, c_i64, c_f32, c_vp, c_sz, c_u64 = (ctypes.c_int, ctypes.c_int64, ctypes.c_float, ctypes.c_void_p,
                                          ctypes.c_size_t, ctypes.c_uint64)


class ConvDesc(ctypes.Structure):
    """Mirror of `vfd_conv_desc` (include/vfdgan_hip.h)."""
    _fields_ = [(n, ctypes.c_int32) for n in
                ("N", "Di", "Hi", "Wi", "Cin", "Do", "Ho", "Wo", "Cout", "kd", "kh", "kw", "sd", "sh", "sw",
                 "pd", "ph", "pw", "transposed", "dtype", "act")] + [("slope", ctypes.c_float)]


# name -> (restype, argtypes); every symbol include/vfdgan_hip.h declares
SIGNATURES = {
    "vfd_abi_version": (c_int, []),
    "vfd_last_error": (ctypes.c_char_p, []),
    "vfd_ncs_to_nsc": (c_int, [c_int, c_vp, c_vp, c_i64, c_int, c_i64, c_vp]),
    "vfd_nsc_to_ncs": (c_int, [c_int, c_vp, c_vp, c_i64, c_int, c_i64, c_vp]),
    "vfd_unflatten": (c_int, [c_int, c_vp, c_vp, c_i64, c_int, c_i64, c_vp]),
    "vfd_flatten": (c_int, [c_int, c_vp, c_vp, c_i64, c_int, c_i64, c_vp]),
    "vfd_pack_filter": (c_int, [c_int, c_vp, c_vp, c_int, c_int, c_int, c_int, c_vp]),
    "vfd_pack_filter_blocks": (c_i64, [c_int, c_int, c_int, c_int]),
    "vfd_pack_filters": (c_int, [c_vp, c_int, c_i64, c_vp]),
    "vfd_conv_forward": (c_int, [ctypes.POINTER(ConvDesc), c_vp, c_vp, c_vp, c_vp, c_vp, c_sz, c_vp, c_sz, c_vp]),
    "vfd_conv_bn_backward_supported": (c_int, [ctypes.POINTER(ConvDesc)]),
    "vfd_conv_set_bn_handover_min_channels": (c_int, [c_int]),
    "vfd_conv_forward_bn_backward": (c_int, [ctypes.POINTER(ConvDesc), c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_int, c_f32,
                                             c_vp, c_sz, c_vp]),
    "vfd_amax": (c_int, [c_int, c_vp, c_i64, c_int, c_vp, c_vp]),
    "vfd_quantize_fp8": (c_int, [c_int, c_vp, c_vp, c_i64, c_int, c_vp, c_vp, c_vp]),
    "vfd_pack_filter_fp8": (c_int, [c_vp, c_vp, c_int, c_int, c_int, c_int, c_vp, c_vp, c_vp]),
    "vfd_dequantize_fp8": (c_int, [c_vp, c_vp, c_i64, c_vp, c_vp]),
    "vfd_conv_forward_fp8": (c_int, [ctypes.POINTER(ConvDesc), c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_sz, c_vp]),
    "vfd_conv_forward_mul": (c_int, [ctypes.POINTER(ConvDesc), c_vp, c_vp, c_vp, c_vp, c_vp, c_int, c_f32, c_vp]),
    "vfd_conv_workspace": (c_int, [ctypes.POINTER(ConvDesc), c_int, ctypes.POINTER(c_sz)]),
    "vfd_conv_kernel_name": (c_int, [ctypes.POINTER(ConvDesc), c_int, ctypes.c_char_p, c_sz]),
    "vfd_conv_set_halo_mode": (c_int, [c_int]),
    "vfd_conv_set_tile_sub": (c_int, [c_int]),
    "vfd_wgrad_workspace": (c_int, [ctypes.POINTER(ConvDesc), ctypes.POINTER(ctypes.c_int32),
                                    ctypes.POINTER(c_sz)]),
    "vfd_conv_wgrad": (c_int, [ctypes.POINTER(ConvDesc), c_vp, c_vp, c_vp, c_sz, c_vp]),
    "vfd_wgrad_reduce": (c_int, [ctypes.POINTER(ConvDesc), c_vp, c_vp, c_f32, c_vp]),
    "vfd_wgrad_reduce_bias": (c_int, [ctypes.POINTER(ConvDesc), c_vp, c_vp, c_f32, c_vp, c_int, c_int, c_vp, c_vp]),
    "vfd_wgrad_set_halo_mode": (c_int, [c_int]),
    "vfd_wgrad_kernel_name": (c_int, [ctypes.POINTER(ConvDesc), ctypes.c_char_p, c_sz]),
    "vfd_bias_grad_workspace": (c_sz, [c_int]),
    "vfd_bias_grad": (c_int, [c_int, c_vp, c_vp, c_i64, c_int, c_f32, c_vp, c_vp]),
    "vfd_bn_workspace": (c_sz, [c_i64, c_int]),
    "vfd_bn_stats": (c_int, [c_int, c_vp, c_i64, c_int, c_f32, c_f32, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp]),
    "vfd_bn_stats_from_sums": (c_int, [c_vp, c_i64, c_int, c_f32, c_f32, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp]),
    "vfd_bn_running_update": (c_int, [c_vp, c_vp, c_i64, c_int, c_f32, c_f32, c_vp, c_vp, c_vp, c_vp]),
    "vfd_bn_act_forward": (c_int, [c_int, c_vp, c_vp, c_i64, c_int, c_vp, c_vp, c_vp, c_vp, c_int, c_f32, c_vp]),
    "vfd_bn_act_forward_sums": (c_int, [c_int, c_vp, c_vp, c_i64, c_int, c_vp, c_f32, c_f32, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp,
                                        c_int, c_f32, c_vp]),
    "vfd_bn_backward_apply_sums": (c_int, [c_int, c_vp, c_vp, c_vp, c_i64, c_int, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp,
                                           c_vp]),
    "vfd_bn_act_backward_sums": (c_int, [c_int, c_vp, c_vp, c_vp, c_i64, c_int, c_vp, c_vp, c_vp, c_vp, c_int, c_f32, c_vp, c_vp, c_vp,
                                         c_vp, c_vp, c_vp, c_vp]),
    "vfd_bn_act_pool_forward_sums": (c_int, [c_int, c_vp, c_vp, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_vp, c_f32, c_f32, c_vp, c_vp, c_vp,
                                             c_vp, c_vp, c_vp, c_vp, c_int, c_f32, c_vp, c_vp]),
    "vfd_bn_act_pool_backward_sums": (c_int, [c_int, c_vp, c_vp, c_vp, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_vp, c_vp, c_vp, c_vp, c_int,
                                              c_f32, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp]),
    "vfd_bn_act_backward": (c_int, [c_int, c_vp, c_vp, c_vp, c_i64, c_int, c_vp, c_vp, c_vp, c_vp, c_int, c_f32,
                                    c_vp, c_vp, c_vp, c_vp, c_vp, c_vp]),
    "vfd_act_forward": (c_int, [c_int, c_vp, c_vp, c_i64, c_int, c_int, c_f32, c_vp]),
    "vfd_act_backward": (c_int, [c_int, c_vp, c_vp, c_vp, c_i64, c_int, c_int, c_f32, c_vp]),
    "vfd_avgpool_forward": (c_int, [c_int, c_vp, c_vp] + [c_int] * 8 + [c_vp]),
    "vfd_avgpool_backward": (c_int, [c_int, c_vp, c_vp] + [c_int] * 8 + [c_vp]),
    "vfd_upsample2x_forward": (c_int, [c_int, c_vp, c_vp] + [c_int] * 5 + [c_vp]),
    "vfd_upsample2x_backward": (c_int, [c_int, c_vp, c_vp] + [c_int] * 5 + [c_vp]),
    "vfd_upsample_forward": (c_int, [c_int, c_vp, c_vp] + [c_int] * 8 + [c_vp]),
    "vfd_upsample_backward": (c_int, [c_int, c_vp, c_vp] + [c_int] * 8 + [c_vp]),
    "vfd_maxpool_forward": (c_int, [c_int, c_vp, c_vp, c_vp] + [c_int] * 14 + [c_vp]),
    "vfd_maxpool_backward": (c_int, [c_int, c_vp, c_vp, c_vp] + [c_int] * 14 + [c_vp]),
    "vfd_add": (c_int, [c_int, c_vp, c_vp, c_vp, c_vp, c_i64, c_vp]),
    "vfd_zero": (c_int, [c_vp, ctypes.c_size_t, c_vp]),
    "vfd_weighted_sum4": (c_int, [c_vp, c_vp, c_vp, c_vp, c_f32, c_f32, c_f32, c_f32, c_int, c_vp, c_vp]),
    "vfd_scale4": (c_int, [c_vp, c_f32, c_f32, c_f32, c_f32, c_int, c_vp, c_vp]),
    "vfd_upsample2x_cat_forward": (c_int, [c_int, c_vp, c_vp, c_vp, c_int, c_int, c_int, c_int, c_int, c_int, c_vp]),
    "vfd_upsample2x_cat_backward": (c_int, [c_int, c_vp, c_vp, c_vp, c_int, c_int, c_int, c_int, c_int, c_int, c_vp]),
    "vfd_morph_open5x5": (c_int, [c_vp, c_vp, c_vp, c_i64, c_int, c_int, c_f32, c_int, c_vp]),
    "vfd_concat_channels": (c_int, [c_int, c_vp, c_vp, c_vp, c_i64, c_int, c_int, c_vp]),
    "vfd_split_channels": (c_int, [c_int, c_vp, c_vp, c_vp, c_i64, c_int, c_int, c_vp]),
    "vfd_broadcast_channel": (c_int, [c_int, c_vp, c_vp, c_i64, c_int, c_vp]),
    "vfd_dropout_forward": (c_int, [c_int, c_vp, c_vp, c_vp, c_vp, c_i64, c_f32, c_u64, c_vp, c_vp]),
    "vfd_dropout_backward": (c_int, [c_int, c_vp, c_vp, c_vp, c_i64, c_f32, c_vp]),
    "vfd_loss_workspace": (c_sz, [c_i64, c_int]),
    "vfd_loss_forward": (c_int, [c_int, c_int, c_vp, c_vp, c_f32, c_vp, c_i64, c_int, c_f32, c_vp, c_vp]),
    "vfd_loss_backward": (c_int, [c_int, c_int, c_vp, c_vp, c_f32, c_vp, c_vp, c_vp, c_i64, c_int, c_f32, c_f32,
                                  c_vp]),
    "vfd_adam_step_dev": (c_int, [c_vp, c_vp, c_vp, c_vp, c_i64, c_f32, c_f32, c_f32, c_f32, c_vp, c_vp, c_f32, c_vp]),
    "vfd_adam_step": (c_int, [c_vp, c_vp, c_vp, c_vp, c_i64, c_f32, c_f32, c_f32, c_f32, ctypes.c_int32, c_f32,
                              c_vp]),
}

_lib = None


class HipLibraryError(RuntimeError):
    pass


def load():
    """Load the shared library (once).  Raises HipLibraryError when it is missing: there is no CPU fallback."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise HipLibraryError(
            "libvfdgan_hip.so not found at %s - build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "or `make -C vfd_gan_amd/csrc` (there is no CPU fallback on the product path)" % LIB_PATH)
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is missing: fail loudly
        fn.restype = res
        fn.argtypes = args
    if lib.vfd_abi_version() != 1:
        raise HipLibraryError("libvfdgan_hip.so ABI version %d != 1" % lib.vfd_abi_version())
    _lib = lib
    return lib


def check(rc, what=""):
    if rc != 0:
        msg = load().vfd_last_error()
        raise RuntimeError("libvfdgan_hip %s failed (rc=%d): %s" % (what, rc, msg.decode() if msg else "?"))


def dtype_code(dt):
    if dt == torch.float32:
        return F32
    if dt == torch.bfloat16:
        return BF16
    raise TypeError("unsupported compute dtype %r (float32 or bfloat16)" % (dt,))


def stream():
    """Raw hipStream_t of torch's current stream (so launches are ordered with torch's allocator and graphs)."""
    return torch.cuda.current_stream().cuda_stream


def ptr(t):
    return 0 if t is None else t.data_ptr()


def require_device(t):
    if not t.is_cuda:
        raise HipLibraryError("vfd_gan_amd ops run on the HIP device only (got a %s tensor); "
                              "there is no CPU fallback" % t.device)
