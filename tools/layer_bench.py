#!/usr/bin/env python3
"""Per-layer timing of the conv family on the bench workload's geometries (ganomaly 16x112x112, 512 frames):
forward, data gradient and filter gradient of every distinct layer, TFLOP/s against the bf16 MFMA peak."""
import argparse
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from vfd_gan_amd import _lib, functional as F  # noqa: E402

# 3-D layers of anogan at 16x112x112 (BASELINE configs[2]): name, Cin, Cout, (D,H,W) in, k, s, p, transposed
LAYERS3D = [
    ("D.l1 32->64 k3 @16x112", 32, 64, (16, 112, 112), 3, 1, 1, False),
    ("D.l1 64->64 k3 @16x112", 64, 64, (16, 112, 112), 3, 1, 1, False),
    ("D.l2 64->128 k3 @8x56", 64, 128, (8, 56, 56), 3, 1, 1, False),
    ("D.l2 128->128 k3 @8x56", 128, 128, (8, 56, 56), 3, 1, 1, False),
    ("D.l2 128->256 k3 @4x28", 128, 256, (4, 28, 28), 3, 1, 1, False),
    ("G.l3 128->64 convT k3s1 @8x56", 128, 64, (8, 56, 56), 3, 1, 1, True),
    ("G.l3 64->64 k3 @8x56", 64, 64, (8, 56, 56), 3, 1, 1, False),
]

LAYERS = [
    # name, Cin, Cout, H(in), k, s, p, transposed
    ("enc.init 3->64 k4s2 @112", 3, 64, 112, 4, 2, 1, False),
    ("enc.pyr 64->128 k4s2 @56", 64, 128, 56, 4, 2, 1, False),
    ("enc.pyr 128->256 k4s2 @28", 128, 256, 28, 4, 2, 1, False),
    ("enc.pyr 256->512 k4s2 @14", 256, 512, 14, 4, 2, 1, False),
    ("enc.final 512->100 k7 @7", 512, 100, 7, 7, 1, 0, False),
    ("D.cls 512->1 k7 @7", 512, 1, 7, 7, 1, 0, False),
    ("dec.init 100->512 k7 convT @1", 100, 512, 1, 7, 1, 0, True),
    ("dec.pyr 512->256 convT @7", 512, 256, 7, 4, 2, 1, True),
    ("dec.pyr 256->128 convT @14", 256, 128, 14, 4, 2, 1, True),
    ("dec.pyr 128->64 convT @28", 128, 64, 28, 4, 2, 1, True),
    ("dec.final 64->3 convT @56", 64, 3, 56, 4, 2, 1, True),
]


def timeit(fn, iters):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3  # us


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=512)
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--dtype", default="bf16")
    ap.add_argument("--only", default="")
    ap.add_argument("--set", default="ganomaly", choices=["ganomaly", "anogan"], help="ganomaly: 2-D pyramid, 512 frames; anogan: 3-D layers, 32 clips")
    ap.add_argument("--fp8", action="store_true", help="also time the e4m3 forward / data gradient of every layer with >= 16 channels")
    ap.add_argument("--halo", type=int, default=0, help="vfd_conv_set_halo_mode: 0 default rules, 1 never (conv_igemm), 2 always")
    a = ap.parse_args()
    dt = torch.bfloat16 if a.dtype == "bf16" else torch.float32
    dev = torch.device("cuda", 0)
    lib = _lib.load()
    lib.vfd_conv_set_halo_mode(a.halo)
    N = a.frames if a.set == "ganomaly" else 32
    layers = LAYERS if a.set == "ganomaly" else LAYERS3D
    tot = {"fwd": 0.0, "dgrad": 0.0, "wgrad": 0.0}
    print("%-34s %10s %8s | %10s %8s | %10s %8s" % ("layer", "fwd us", "TF/s", "dgrad us", "TF/s", "wgrad us", "TF/s"))
    for name, cin, cout, h, k, s, p, tr in layers:
        if a.only and a.only not in name:
            continue
        dims = (1, h, h) if isinstance(h, int) else h
        nd3 = not isinstance(h, int)
        kk, ss, pp = ((k, k, k), (s, s, s), (p, p, p)) if nd3 else ((1, k, k), (1, s, s), (0, p, p))
        odims = tuple(((dims[i] - 1) * ss[i] - 2 * pp[i] + kk[i]) if tr else ((dims[i] + 2 * pp[i] - kk[i]) // ss[i] + 1) for i in range(3))
        x = torch.randn((N,) + dims + (F.cpad(cin),), device=dev).to(dt)
        x[..., cin:] = 0
        gy = torch.randn((N,) + odims + (F.cpad(cout),), device=dev).to(dt)
        gy[..., cout:] = 0
        w = torch.nn.Parameter(torch.randn(((cin, cout) if tr else (cout, cin)) + (kk if nd3 else kk[1:]), device=dev) * 0.05)
        T = kk[0] * kk[1] * kk[2]
        A, B = (cin, cout) if tr else (cout, cin)
        pk_f = F._packed_filter(w, dt, bool(tr), A, B, T)
        pk_d = F._packed_filter(w, dt, not tr, A, B, T)
        y = torch.empty_like(gy)
        gx = torch.empty_like(x)
        d_f = F._make_desc(N, dims, cin, odims, cout, kk, ss, pp, tr, dt)
        d_d = F._make_desc(N, odims, cout, dims, cin, kk, ss, pp, not tr, dt)
        nsplit, nbytes = ctypes.c_int32(), ctypes.c_size_t()
        _lib.check(lib.vfd_wgrad_workspace(ctypes.byref(d_f), ctypes.byref(nsplit), ctypes.byref(nbytes)))
        ws = torch.empty(nbytes.value, dtype=torch.uint8, device=dev)
        gw = torch.empty_like(w)
        st = _lib.stream()
        px_in, px_out = dims[0] * dims[1] * dims[2], odims[0] * odims[1] * odims[2]
        flops = 2.0 * N * (px_in if tr else px_out) * T * cin * cout
        def wsp(d):
            need = ctypes.c_size_t()
            _lib.check(lib.vfd_conv_workspace(ctypes.byref(d), 0, ctypes.byref(need)))
            return torch.empty(max(need.value, 16), dtype=torch.uint8, device=dev), need.value
        wf, nf = wsp(d_f)
        wd, ndb = wsp(d_d)
        t_f = timeit(lambda: lib.vfd_conv_forward(ctypes.byref(d_f), x.data_ptr(), pk_f.data_ptr(), 0, y.data_ptr(), 0, 0, wf.data_ptr(), nf, st), a.iters)
        t_d = timeit(lambda: lib.vfd_conv_forward(ctypes.byref(d_d), gy.data_ptr(), pk_d.data_ptr(), 0, gx.data_ptr(), 0, 0, wd.data_ptr(), ndb, st), a.iters)

        def wg():
            lib.vfd_conv_wgrad(ctypes.byref(d_f), x.data_ptr(), gy.data_ptr(), ws.data_ptr(), nbytes.value, st)
            lib.vfd_wgrad_reduce(ctypes.byref(d_f), ws.data_ptr(), gw.data_ptr(), 0.0, st)
        t_w = timeit(wg, a.iters)
        tot["fwd"] += t_f; tot["dgrad"] += t_d; tot["wgrad"] += t_w
        extra = ""
        if a.fp8 and cin >= 16 and cout >= 16:
            # the same layer with e4m3 operands (forward and data gradient; the quantisation passes timed on their own)
            xq, xs = F.quantize_fp8(x, cin)
            gq, gs = F.quantize_fp8(gy, cout)
            wq_f, ws_f = F.pack_filter_fp8(w, bool(tr), A, B, T)
            wq_d, ws_d = F.pack_filter_fp8(w, not tr, A, B, T)
            t8f = timeit(lambda: F.conv_fp8(xq, xs, wq_f, ws_f, None, N, dims, cin, odims, cout, kk, ss, pp, tr), a.iters)
            t8d = timeit(lambda: F.conv_fp8(gq, gs, wq_d, ws_d, None, N, odims, cout, dims, cin, kk, ss, pp, not tr), a.iters)
            t8q = timeit(lambda: F.quantize_fp8(x, cin), a.iters)
            tot["fp8 fwd"] = tot.get("fp8 fwd", 0.0) + t8f
            tot["fp8 dgrad"] = tot.get("fp8 dgrad", 0.0) + t8d
            extra = "  || fp8 fwd %8.1f us %7.1f TF/s | dgrad %8.1f us %7.1f TF/s | quantise x %6.1f us" % (t8f, flops / t8f / 1e6, t8d, flops / t8d / 1e6, t8q)
        print("%-34s %10.1f %8.1f | %10.1f %8.1f | %10.1f %8.1f   (split %d)%s" % (name, t_f, flops / t_f / 1e6, t_d, flops / t_d / 1e6,
                                                                     t_w, flops / t_w / 1e6, nsplit.value, extra))
    print("totals us:", {k: round(v, 1) for k, v in tot.items()})


if __name__ == "__main__":
    main()
