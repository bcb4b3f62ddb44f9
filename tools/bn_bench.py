#!/usr/bin/env python3
"""Streaming rate of the BatchNorm(+activation) kernels on the bench workload's four BatchNorm geometries (bf16)."""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from vfd_gan_amd import _lib  # noqa: E402


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
    lib = _lib.load()
    dev = torch.device("cuda", 0)
    st = _lib.stream()
    print("%-16s %10s %8s | %10s %8s" % ("layer", "fwd us", "TB/s", "bwd us", "TB/s"))
    for C, hw in ((64, 56), (128, 28), (256, 14), (512, 7)):
        rows = 512 * hw * hw
        x = torch.randn(rows, C, device=dev).bfloat16()
        dy = torch.randn(rows, C, device=dev).bfloat16()
        y = torch.empty_like(x)
        dx = torch.empty_like(x)
        mean = torch.zeros(C, device=dev)
        rstd = torch.ones(C, device=dev)
        gamma = torch.ones(C, device=dev)
        beta = torch.zeros(C, device=dev)
        dg = torch.zeros(C, device=dev)
        db = torch.zeros(C, device=dev)
        ws = torch.empty(lib.vfd_bn_workspace(rows, C), dtype=torch.uint8, device=dev)
        nbytes = rows * C * 2
        tf = timeit(lambda: lib.vfd_bn_act_forward(_lib.BF16, x.data_ptr(), y.data_ptr(), rows, C, mean.data_ptr(), rstd.data_ptr(),
                                                   gamma.data_ptr(), beta.data_ptr(), 1, 0.2, st))
        tb = timeit(lambda: lib.vfd_bn_act_backward(_lib.BF16, x.data_ptr(), dy.data_ptr(), dx.data_ptr(), rows, C, mean.data_ptr(),
                                                    rstd.data_ptr(), gamma.data_ptr(), beta.data_ptr(), 1, 0.2, dg.data_ptr(),
                                                    db.data_ptr(), 0, 0, ws.data_ptr(), st))
        print("%-16s %10.1f %8.2f | %10.1f %8.2f" % ("%dch @%d" % (C, hw), tf, 2 * nbytes / tf / 1e6, tb, 5 * nbytes / tb / 1e6))


if __name__ == "__main__":
    main()
