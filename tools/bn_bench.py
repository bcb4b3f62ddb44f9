#!/usr/bin/env python3
"""Timing of the BatchNorm(+activation) passes alone, on layer shapes of the three bench workloads: the three-launch backward
(vfd_bn_act_backward), the two-launch one (vfd_bn_act_backward_sums, with and without the fused conv-bias column sum) and
the forward with / without the statistics fold."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from vfd_gan_amd import _lib, functional as F  # noqa: E402
from vfd_gan_amd._lib import check, ptr, stream  # noqa: E402

SHAPES = [("ganomaly 512x56x56 c64", 512 * 56 * 56, 64), ("ganomaly 512x28x28 c128", 512 * 28 * 28, 128),
          ("ganomaly 512x14x14 c256", 512 * 14 * 14, 256), ("anogan 32x16x112x112 c64", 32 * 16 * 112 * 112, 64),
          ("anogan 32x8x56x56 c128", 32 * 8 * 56 * 56, 128), ("mygan 8x16x224x224 c32", 8 * 16 * 224 * 224, 32),
          ("mygan 8x8x112x112 c115", 8 * 8 * 112 * 112, 115)]


def timeit(fn, iters=20):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


def main():
    dev = torch.device("cuda", 0)
    lib = _lib.load()
    dt = torch.bfloat16
    dtc = _lib.dtype_code(dt)
    print("%-28s %9s %9s | %9s %9s %9s   (us; GB/s of the 2-launch backward: 4 reads + 1 write)" % ("shape", "fwd", "fwd+fold", "bwd 3", "bwd 2", "bwd 2+cs"))
    for name, rows, C in SHAPES:
        Cp = F.cpad(C)
        x = torch.randn(rows, Cp, device=dev).to(dt)
        dy = torch.randn(rows, Cp, device=dev).to(dt)
        y, dx = torch.empty_like(x), torch.empty_like(x)
        mean, rstd = torch.zeros(C, device=dev), torch.ones(C, device=dev)
        gamma, beta = torch.rand(C, device=dev) + 0.5, torch.randn(C, device=dev)
        dg, db, cs = torch.empty(C, device=dev), torch.empty(C, device=dev), torch.zeros(F.stats_buffer_numel(C), device=dev)
        sums = F.new_stats_buffer(C, dev)      # forward statistics: float64
        bsums = torch.zeros(F.stats_buffer_numel(C), device=dev)
        ws = torch.empty(lib.vfd_bn_workspace(rows, C), dtype=torch.uint8, device=dev)
        st = stream()

        def fwd():
            check(lib.vfd_bn_act_forward(dtc, x.data_ptr(), y.data_ptr(), rows, C, mean.data_ptr(), rstd.data_ptr(), gamma.data_ptr(),
                                         beta.data_ptr(), 1, 0.2, st), "f")

        def fwd_fold():
            check(lib.vfd_bn_act_forward_sums(dtc, x.data_ptr(), y.data_ptr(), rows, C, sums.data_ptr(), 1e-5, 0.1, mean.data_ptr(),
                                              rstd.data_ptr(), None, None, None, gamma.data_ptr(), beta.data_ptr(), 1, 0.2, st), "ff")

        def bwd3():
            check(lib.vfd_bn_act_backward(dtc, x.data_ptr(), dy.data_ptr(), dx.data_ptr(), rows, C, mean.data_ptr(), rstd.data_ptr(),
                                          gamma.data_ptr(), beta.data_ptr(), 1, 0.2, dg.data_ptr(), db.data_ptr(), None, None,
                                          ws.data_ptr(), st), "b3")

        def bwd2(c=None):
            check(lib.vfd_bn_act_backward_sums(dtc, x.data_ptr(), dy.data_ptr(), dx.data_ptr(), rows, C, mean.data_ptr(), rstd.data_ptr(),
                                               gamma.data_ptr(), beta.data_ptr(), 1, 0.2, bsums.data_ptr(), dg.data_ptr(), db.data_ptr(),
                                               None, None, c, st), "b2")

        t = [timeit(fwd), timeit(fwd_fold), timeit(bwd3), timeit(bwd2), timeit(lambda: bwd2(cs.data_ptr()))]
        gbs = 5 * rows * Cp * 2 / (t[3] * 1e-6) / 1e9
        print("%-28s %9.1f %9.1f | %9.1f %9.1f %9.1f   %6.0f GB/s" % (name, t[0], t[1], t[2], t[3], t[4], gbs))
        del x, dy, y, dx


if __name__ == "__main__":
    main()
