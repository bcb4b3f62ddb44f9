"""Times vfd_upsample_backward at mygan's four decoder joints (8 clips): the gather-form trilinear backward against its HBM time."""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from vfd_gan_amd import _lib  # noqa: E402

lib = _lib.load()
dev = torch.device("cuda", 0)
st = torch.cuda.current_stream().cuda_stream
for (N, D, H, W, C) in ((8, 8, 112, 112, 64), (8, 4, 56, 56, 128), (8, 2, 28, 28, 256), (8, 1, 14, 14, 512)):
    dy = torch.randn((N, 2 * D, 2 * H, 2 * W, C), device=dev).bfloat16()
    dx = torch.empty((N, D, H, W, C), device=dev, dtype=torch.bfloat16)
    for _ in range(3):
        _lib.check(lib.vfd_upsample_backward(1, dy.data_ptr(), dx.data_ptr(), N, D, H, W, C, 2, 2, 2, st))
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        _lib.check(lib.vfd_upsample_backward(1, dy.data_ptr(), dx.data_ptr(), N, D, H, W, C, 2, 2, 2, st))
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 100
    gb = (dy.numel() + dx.numel()) * 2 / 1e9
    print("upsample_backward %dx%dx%dx%d c%d: %8.1f us  (%.2f GB -> %.0f GB/s)" % (N, D, H, W, C, us, gb, gb / us * 1e6))
    # the decoder joint: dy is the leading C channels of a (C + C/2)-channel concatenation (models/mygannet.py:78-94)
    Cb = C // 2
    dcat = torch.randn((N, 2 * D, 2 * H, 2 * W, C + Cb), device=dev).bfloat16()
    dskip = torch.empty((N, 2 * D, 2 * H, 2 * W, Cb), device=dev, dtype=torch.bfloat16)
    for mode in ("0",):
        for _ in range(2):
            _lib.check(lib.vfd_upsample2x_cat_backward(1, dcat.data_ptr(), dx.data_ptr(), dskip.data_ptr(), N, D, H, W, C, Cb, st))
        torch.cuda.synchronize()
        e0.record()
        for _ in range(10):
            _lib.check(lib.vfd_upsample2x_cat_backward(1, dcat.data_ptr(), dx.data_ptr(), dskip.data_ptr(), N, D, H, W, C, Cb, st))
        e1.record()
        torch.cuda.synchronize()
        print("   cat_backward (+ skip copy), row pitch %d B: %8.1f us" % ((C + Cb) * 2, e0.elapsed_time(e1) * 100))
        break
