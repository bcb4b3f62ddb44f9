// AvgPool3d (kernel == stride, no padding; also the "global" pools of SDisc/TDisc) and the trilinear x2
// up-sampling (align_corners=True) of the mygan decoder, forward and backward, on channels-last blocks.
// One thread per output granule (8 channels), 16-byte loads/stores; gather form in both directions, so the
// backward passes are deterministic (no atomics).
#include "common.hpp"

namespace {

constexpr int PL_THREADS = 256;
static inline unsigned pl_blocks(long long n) {
  long long b = (n + PL_THREADS - 1) / PL_THREADS;
  if (b > 256LL * 32) b = 256LL * 32;
  if (b < 1) b = 1;
  return (unsigned)b;
}

template <typename T>
__global__ void avgpool_fwd_kernel(const T* __restrict__ x, T* __restrict__ y, int N, int D, int H, int W, int Cp, int kd, int kh,
                                   int kw) {
  const int GR = Cp >> 3, Do = D / kd, Ho = H / kh, Wo = W / kw;
  const long long total = (long long)N * Do * Ho * Wo * GR;
  const float inv = 1.f / (float)(kd * kh * kw);
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    long long q = i;
    const int g = (int)(q % GR); q /= GR;
    const int ow = (int)(q % Wo); q /= Wo;
    const int oh = (int)(q % Ho); q /= Ho;
    const int od = (int)(q % Do); q /= Do;
    const int n = (int)q;
    float s[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int a = 0; a < kd; ++a)
      for (int b = 0; b < kh; ++b)
        for (int c = 0; c < kw; ++c) {
          const size_t pix = ((size_t)(n * D + od * kd + a) * H + oh * kh + b) * W + ow * kw + c;
          float v[8];
          load8(x + pix * Cp + g * 8, v);
#pragma unroll
          for (int k = 0; k < 8; ++k) s[k] += v[k];
        }
#pragma unroll
    for (int k = 0; k < 8; ++k) s[k] *= inv;
    store8(y + i * 8, s);
  }
}

// Large windows (the "global" pools of SDisc / TDisc: (nfr,1,1) and (1,isize,isize)): one workgroup per output
// granule, threads stride over the window, fixed-order LDS tree (deterministic).
template <typename T>
__global__ __launch_bounds__(256) void avgpool_reduce_kernel(const T* __restrict__ x, T* __restrict__ y, int N, int D, int H, int W,
                                                             int Cp, int kd, int kh, int kw) {
  const int GR = Cp >> 3, Do = D / kd, Ho = H / kh, Wo = W / kw;
  long long q = blockIdx.x;
  const int g = (int)(q % GR); q /= GR;
  const int ow = (int)(q % Wo); q /= Wo;
  const int oh = (int)(q % Ho); q /= Ho;
  const int od = (int)(q % Do); q /= Do;
  const int n = (int)q;
  const int win = kd * kh * kw;
  float s[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  for (int t = threadIdx.x; t < win; t += 256) {
    const int c = t % kw, b = (t / kw) % kh, a = t / (kw * kh);
    const size_t pix = ((size_t)(n * D + od * kd + a) * H + oh * kh + b) * W + ow * kw + c;
    float v[8];
    load8(x + pix * Cp + g * 8, v);
#pragma unroll
    for (int k = 0; k < 8; ++k) s[k] += v[k];
  }
  __shared__ float sh[8][4];
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const float t = wave_sum(s[k]);
    if ((threadIdx.x & 63) == 0) sh[k][threadIdx.x >> 6] = t;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    const float inv = 1.f / (float)win;
    float o[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) o[k] = (((sh[k][0] + sh[k][1]) + sh[k][2]) + sh[k][3]) * inv;
    store8(y + (size_t)blockIdx.x * 8, o);
  }
}

template <typename T>
__global__ void avgpool_bwd_kernel(const T* __restrict__ dy, T* __restrict__ dx, int N, int D, int H, int W, int Cp, int kd, int kh,
                                   int kw) {
  const int GR = Cp >> 3, Do = D / kd, Ho = H / kh, Wo = W / kw;
  const long long total = (long long)N * D * H * W * GR;
  const float inv = 1.f / (float)(kd * kh * kw);
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    long long q = i;
    const int g = (int)(q % GR); q /= GR;
    const int iw = (int)(q % W); q /= W;
    const int ih = (int)(q % H); q /= H;
    const int id = (int)(q % D); q /= D;
    const int n = (int)q;
    const int od = id / kd, oh = ih / kh, ow = iw / kw;
    float v[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (od < Do && oh < Ho && ow < Wo) {
      const size_t pix = ((size_t)(n * Do + od) * Ho + oh) * Wo + ow;
      load8(dy + pix * Cp + g * 8, v);
#pragma unroll
      for (int k = 0; k < 8; ++k) v[k] *= inv;
    }
    store8(dx + i * 8, v);
  }
}

// torch upsample_trilinear3d, align_corners=True: src = o * (I-1)/(O-1) computed in float32
__device__ __forceinline__ void up_src(int o, int I, int O, int& i0, int& i1, float& l1) {
  const float r = (O > 1) ? (float)(I - 1) / (float)(O - 1) : 0.f;
  const float s = r * (float)o;
  i0 = (int)s;
  if (i0 > I - 1) i0 = I - 1;
  i1 = i0 + ((i0 < I - 1) ? 1 : 0);
  l1 = s - (float)i0;
}

// (Do, Ho, Wo) = per-dimension output extents: 2x the input for the U-Net's Upsample(2), or (D, 2H, 2W) for the (1,2,2) scale of
// models/xception.py:84 — an extent equal to the input's makes that dimension the identity (src = o, weight 1)
template <typename T>
__global__ void upsample2x_fwd_kernel(const T* __restrict__ x, T* __restrict__ y, int N, int D, int H, int W, int Cp, int Do, int Ho, int Wo) {
  const int GR = Cp >> 3;
  const long long total = (long long)N * Do * Ho * Wo * GR;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    long long q = i;
    const int g = (int)(q % GR); q /= GR;
    const int ow = (int)(q % Wo); q /= Wo;
    const int oh = (int)(q % Ho); q /= Ho;
    const int od = (int)(q % Do); q /= Do;
    const int n = (int)q;
    int d0, d1, h0, h1, w0, w1;
    float ld, lh, lw;
    up_src(od, D, Do, d0, d1, ld);
    up_src(oh, H, Ho, h0, h1, lh);
    up_src(ow, W, Wo, w0, w1, lw);
    float s[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int c = 0; c < 2; ++c) {
          const float wgt = (a ? ld : 1.f - ld) * (b ? lh : 1.f - lh) * (c ? lw : 1.f - lw);
          const size_t pix = ((size_t)(n * D + (a ? d1 : d0)) * H + (b ? h1 : h0)) * W + (c ? w1 : w0);
          float v[8];
          load8(x + pix * Cp + g * 8, v);
#pragma unroll
          for (int k = 0; k < 8; ++k) s[k] += wgt * v[k];
        }
    store8(y + i * 8, s);
  }
}

// weight with which output index o (of O) reads input index i (of I) along one dimension
__device__ __forceinline__ float up_weight(int o, int i, int I, int O) {
  int i0, i1;
  float l1;
  up_src(o, I, O, i0, i1, l1);
  float w = 0.f;
  if (i0 == i) w += 1.f - l1;
  if (i1 == i) w += l1;
  return w;
}

// torch.cat([Upsample(x), skip], dim=1) in one pass (the U-Net decoder joint, models/mygannet.py:78-94): output granule g of
// pixel o is the interpolation of x for g < Cap/8, else the skip tensor's granule g - Cap/8.  Reads the (small) x and the skip
// once, writes the concatenation once; the up-sampled tensor itself never exists.  Cap = channels of x, a multiple of 8.
template <typename T>
__global__ void upsample2x_cat_fwd_kernel(const T* __restrict__ x, const T* __restrict__ skip, T* __restrict__ y, int N, int D, int H,
                                          int W, int Cap, int Cbp) {
  const int GA = Cap >> 3, GR = (Cap + Cbp) >> 3, Do = 2 * D, Ho = 2 * H, Wo = 2 * W;
  const long long total = (long long)N * Do * Ho * Wo * GR;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    long long q = i;
    const int g = (int)(q % GR); q /= GR;
    float s[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (g >= GA) {
      load8(skip + q * Cbp + (g - GA) * 8, s);      // q = output pixel index
    } else {
      const int ow = (int)(q % Wo); q /= Wo;
      const int oh = (int)(q % Ho); q /= Ho;
      const int od = (int)(q % Do); q /= Do;
      const int n = (int)q;
      int d0, d1, h0, h1, w0, w1;
      float ld, lh, lw;
      up_src(od, D, Do, d0, d1, ld);
      up_src(oh, H, Ho, h0, h1, lh);
      up_src(ow, W, Wo, w0, w1, lw);
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
          for (int c = 0; c < 2; ++c) {
            const float wgt = (a ? ld : 1.f - ld) * (b ? lh : 1.f - lh) * (c ? lw : 1.f - lw);
            const size_t pix = ((size_t)(n * D + (a ? d1 : d0)) * H + (b ? h1 : h0)) * W + (c ? w1 : w0);
            float v[8];
            load8(x + pix * Cap + g * 8, v);
#pragma unroll
            for (int k = 0; k < 8; ++k) s[k] += wgt * v[k];
          }
    }
    store8(y + i * 8, s);
  }
}

// channels [c0, c0 + 8*ng) of a [rows][srcCp] block -> a dense [rows][8*ng] block (the skip half of a concatenation's gradient)
template <typename T>
__global__ void slice_copy_kernel(const T* __restrict__ src, T* __restrict__ dst, long long rows, int srcCp, int g0, int ng) {
  const long long total = rows * ng;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const long long r = i / ng;
    const int g = (int)(i - r * ng);
    float v[8];
    load8(src + r * srcCp + (g0 + g) * 8, v);
    store8(dst + i * 8, v);
  }
}

// dyCp: row length of dy (>= Cp: the gradient may be the leading channels of a wider, concatenated tensor)
template <typename T>
__global__ void upsample2x_bwd_kernel(const T* __restrict__ dy, T* __restrict__ dx, int N, int D, int H, int W, int Cp, int dyCp, int Do, int Ho, int Wo) {
  const int GR = Cp >> 3;
  const int fd = Do / D, fh = Ho / H, fw = Wo / W;     // 1 or 2 per dimension
  const long long total = (long long)N * D * H * W * GR;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    long long q = i;
    const int g = (int)(q % GR); q /= GR;
    const int iw = (int)(q % W); q /= W;
    const int ih = (int)(q % H); q /= H;
    const int id = (int)(q % D); q /= D;
    const int n = (int)q;
    float s[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    // src(o) in (i-1, i+1)  =>  o in [f i - 2, f i + 3]  (f = 2: 1/r = 2 + 1/(I-1); f = 1: o == i, inside the range)
    for (int od = max(0, fd * id - 2); od <= min(Do - 1, fd * id + 3); ++od) {
      const float wd = up_weight(od, id, D, Do);
      if (wd == 0.f) continue;
      for (int oh = max(0, fh * ih - 2); oh <= min(Ho - 1, fh * ih + 3); ++oh) {
        const float wh = up_weight(oh, ih, H, Ho);
        if (wh == 0.f) continue;
        for (int ow = max(0, fw * iw - 2); ow <= min(Wo - 1, fw * iw + 3); ++ow) {
          const float ww = up_weight(ow, iw, W, Wo);
          if (ww == 0.f) continue;
          const size_t pix = ((size_t)(n * Do + od) * Ho + oh) * Wo + ow;
          float v[8];
          load8(dy + pix * dyCp + g * 8, v);
          const float wgt = wd * wh * ww;
#pragma unroll
          for (int k = 0; k < 8; ++k) s[k] += wgt * v[k];
        }
      }
    }
    store8(dx + i * 8, s);
  }
}

// MaxPool3d (models/xception.py:57: kernel (1,3,3), stride (1,s,s), padding (0,1,1)), any kernel of <= 255 taps with padding.
// torch semantics: the window is clipped to the input, the FIRST maximum in (d,h,w) scan order wins (strict >), a NaN wins.
// The forward also stores, per output element, the window-local index of its maximum (1 byte), so that the backward is a
// gather: an input element receives dy of every window whose stored index names it (deterministic, no atomics).
struct PoolGeom { int kd, kh, kw, sd, sh, sw, pd, ph, pw; };

template <typename T>
__global__ void maxpool_fwd_kernel(const T* __restrict__ x, T* __restrict__ y, uint8_t* __restrict__ idx, int N, int D, int H, int W,
                                   int Cp, int Do, int Ho, int Wo, PoolGeom g_) {
  const int GR = Cp >> 3;
  const long long total = (long long)N * Do * Ho * Wo * GR;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    long long q = i;
    const int g = (int)(q % GR); q /= GR;
    const int ow = (int)(q % Wo); q /= Wo;
    const int oh = (int)(q % Ho); q /= Ho;
    const int od = (int)(q % Do); q /= Do;
    const int n = (int)q;
    float m[8];
    uint32_t mi[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) { m[k] = -INFINITY; mi[k] = 0xffu; }
    for (int a = 0; a < g_.kd; ++a) {
      const int id = od * g_.sd - g_.pd + a;
      if ((unsigned)id >= (unsigned)D) continue;
      for (int b = 0; b < g_.kh; ++b) {
        const int ih = oh * g_.sh - g_.ph + b;
        if ((unsigned)ih >= (unsigned)H) continue;
        for (int c = 0; c < g_.kw; ++c) {
          const int iw = ow * g_.sw - g_.pw + c;
          if ((unsigned)iw >= (unsigned)W) continue;
          const uint32_t t = (uint32_t)((a * g_.kh + b) * g_.kw + c);
          float v[8];
          load8(x + ((((size_t)n * D + id) * H + ih) * W + iw) * Cp + g * 8, v);
#pragma unroll
          for (int k = 0; k < 8; ++k)
            if (v[k] > m[k] || v[k] != v[k] || mi[k] == 0xffu) { m[k] = v[k]; mi[k] = t; }
        }
      }
    }
    store8(y + i * 8, m);
    uint2 o;
    o.x = mi[0] | (mi[1] << 8) | (mi[2] << 16) | (mi[3] << 24);
    o.y = mi[4] | (mi[5] << 8) | (mi[6] << 16) | (mi[7] << 24);
    *reinterpret_cast<uint2*>(idx + i * 8) = o;
  }
}

template <typename T>
__global__ void maxpool_bwd_kernel(const T* __restrict__ dy, const uint8_t* __restrict__ idx, T* __restrict__ dx, int N, int D, int H,
                                   int W, int Cp, int Do, int Ho, int Wo, PoolGeom g_) {
  const int GR = Cp >> 3;
  const long long total = (long long)N * D * H * W * GR;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    long long q = i;
    const int g = (int)(q % GR); q /= GR;
    const int iw = (int)(q % W); q /= W;
    const int ih = (int)(q % H); q /= H;
    const int id = (int)(q % D); q /= D;
    const int n = (int)q;
    float s[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    // windows that contain this element: o*s - p <= i <= o*s - p + k - 1
    const int od0 = max(0, (id + g_.pd - g_.kd + g_.sd) / g_.sd), od1 = min(Do - 1, (id + g_.pd) / g_.sd);
    const int oh0 = max(0, (ih + g_.ph - g_.kh + g_.sh) / g_.sh), oh1 = min(Ho - 1, (ih + g_.ph) / g_.sh);
    const int ow0 = max(0, (iw + g_.pw - g_.kw + g_.sw) / g_.sw), ow1 = min(Wo - 1, (iw + g_.pw) / g_.sw);
    for (int od = od0; od <= od1; ++od)
      for (int oh = oh0; oh <= oh1; ++oh)
        for (int ow = ow0; ow <= ow1; ++ow) {
          const uint32_t t = (uint32_t)(((id - (od * g_.sd - g_.pd)) * g_.kh + (ih - (oh * g_.sh - g_.ph))) * g_.kw + (iw - (ow * g_.sw - g_.pw)));
          const size_t o = ((((size_t)n * Do + od) * Ho + oh) * Wo + ow) * Cp + g * 8;
          const uint2 w = *reinterpret_cast<const uint2*>(idx + o);
          float v[8];
          load8(dy + o, v);
#pragma unroll
          for (int k = 0; k < 8; ++k) {
            const uint32_t mk = ((k < 4 ? w.x : w.y) >> (8 * (k & 3))) & 0xffu;
            if (mk == t) s[k] += v[k];
          }
        }
    store8(dx + i * 8, s);
  }
}

}  // namespace

#define POOL_ARGS_OK(name) \
  VFD_REQUIRE(dtype == VFD_F32 || dtype == VFD_BF16, name ": bad dtype"); \
  VFD_REQUIRE(N > 0 && D > 0 && H > 0 && W > 0 && C > 0, name ": bad dims")

extern "C" int vfd_avgpool_forward(int dtype, const void* x, void* y, int N, int D, int H, int W, int C, int kd, int kh, int kw,
                                   void* stream) {
  POOL_ARGS_OK("avgpool_forward");
  VFD_REQUIRE(x && y && kd > 0 && kh > 0 && kw > 0 && D >= kd && H >= kh && W >= kw, "avgpool_forward: bad kernel");
  const int Cp = cpad(C);
  const long long total = (long long)N * (D / kd) * (H / kh) * (W / kw) * (Cp >> 3);
  if (kd * kh * kw >= 256 && total < 0x7fffffffLL) {     // global pools: few outputs, huge windows
    if (dtype == VFD_BF16)
      hipLaunchKernelGGL(avgpool_reduce_kernel<bf16_t>, dim3((unsigned)total), dim3(256), 0, as_stream(stream), (const bf16_t*)x, (bf16_t*)y, N, D, H, W, Cp, kd, kh, kw);
    else
      hipLaunchKernelGGL(avgpool_reduce_kernel<float>, dim3((unsigned)total), dim3(256), 0, as_stream(stream), (const float*)x, (float*)y, N, D, H, W, Cp, kd, kh, kw);
    VFD_CHECK_LAUNCH("avgpool_reduce");
    return VFD_OK;
  }
  if (dtype == VFD_BF16)
    hipLaunchKernelGGL(avgpool_fwd_kernel<bf16_t>, dim3(pl_blocks(total)), dim3(PL_THREADS), 0, as_stream(stream), (const bf16_t*)x, (bf16_t*)y, N, D, H, W, Cp, kd, kh, kw);
  else
    hipLaunchKernelGGL(avgpool_fwd_kernel<float>, dim3(pl_blocks(total)), dim3(PL_THREADS), 0, as_stream(stream), (const float*)x, (float*)y, N, D, H, W, Cp, kd, kh, kw);
  VFD_CHECK_LAUNCH("avgpool_forward");
  return VFD_OK;
}

extern "C" int vfd_avgpool_backward(int dtype, const void* dy, void* dx, int N, int D, int H, int W, int C, int kd, int kh, int kw,
                                    void* stream) {
  POOL_ARGS_OK("avgpool_backward");
  VFD_REQUIRE(dy && dx && kd > 0 && kh > 0 && kw > 0 && D >= kd && H >= kh && W >= kw, "avgpool_backward: bad kernel");
  const int Cp = cpad(C);
  const long long total = (long long)N * D * H * W * (Cp >> 3);
  if (dtype == VFD_BF16)
    hipLaunchKernelGGL(avgpool_bwd_kernel<bf16_t>, dim3(pl_blocks(total)), dim3(PL_THREADS), 0, as_stream(stream), (const bf16_t*)dy, (bf16_t*)dx, N, D, H, W, Cp, kd, kh, kw);
  else
    hipLaunchKernelGGL(avgpool_bwd_kernel<float>, dim3(pl_blocks(total)), dim3(PL_THREADS), 0, as_stream(stream), (const float*)dy, (float*)dx, N, D, H, W, Cp, kd, kh, kw);
  VFD_CHECK_LAUNCH("avgpool_backward");
  return VFD_OK;
}

extern "C" int vfd_upsample2x_forward(int dtype, const void* x, void* y, int N, int D, int H, int W, int C, void* stream) {
  POOL_ARGS_OK("upsample2x_forward");
  VFD_REQUIRE(x && y, "upsample2x_forward: null pointer");
  const int Cp = cpad(C);
  const long long total = (long long)N * D * H * W * 8 * (Cp >> 3);
  if (dtype == VFD_BF16)
    hipLaunchKernelGGL(upsample2x_fwd_kernel<bf16_t>, dim3(pl_blocks(total)), dim3(PL_THREADS), 0, as_stream(stream), (const bf16_t*)x, (bf16_t*)y, N, D, H, W, Cp, 2 * D, 2 * H, 2 * W);
  else
    hipLaunchKernelGGL(upsample2x_fwd_kernel<float>, dim3(pl_blocks(total)), dim3(PL_THREADS), 0, as_stream(stream), (const float*)x, (float*)y, N, D, H, W, Cp, 2 * D, 2 * H, 2 * W);
  VFD_CHECK_LAUNCH("upsample2x_forward");
  return VFD_OK;
}

// Upsample(scale_factor=(fd,fh,fw), mode='trilinear', align_corners=True) with every factor 1 or 2 (models/xception.py:84: (1,2,2))
extern "C" int vfd_upsample_forward(int dtype, const void* x, void* y, int N, int D, int H, int W, int C, int fd, int fh, int fw, void* stream) {
  POOL_ARGS_OK("upsample_forward");
  VFD_REQUIRE(x && y && (fd == 1 || fd == 2) && (fh == 1 || fh == 2) && (fw == 1 || fw == 2), "upsample_forward: scale factors must be 1 or 2");
  const int Cp = cpad(C);
  const long long total = (long long)N * D * fd * H * fh * W * fw * (Cp >> 3);
  if (dtype == VFD_BF16)
    hipLaunchKernelGGL(upsample2x_fwd_kernel<bf16_t>, dim3(pl_blocks(total)), dim3(PL_THREADS), 0, as_stream(stream), (const bf16_t*)x, (bf16_t*)y, N, D, H, W, Cp, fd * D, fh * H, fw * W);
  else
    hipLaunchKernelGGL(upsample2x_fwd_kernel<float>, dim3(pl_blocks(total)), dim3(PL_THREADS), 0, as_stream(stream), (const float*)x, (float*)y, N, D, H, W, Cp, fd * D, fh * H, fw * W);
  VFD_CHECK_LAUNCH("upsample_forward");
  return VFD_OK;
}

extern "C" int vfd_upsample_backward(int dtype, const void* dy, void* dx, int N, int D, int H, int W, int C, int fd, int fh, int fw, void* stream) {
  POOL_ARGS_OK("upsample_backward");
  VFD_REQUIRE(dy && dx && (fd == 1 || fd == 2) && (fh == 1 || fh == 2) && (fw == 1 || fw == 2), "upsample_backward: scale factors must be 1 or 2");
  const int Cp = cpad(C);
  const long long total = (long long)N * D * H * W * (Cp >> 3);
  if (dtype == VFD_BF16)
    hipLaunchKernelGGL(upsample2x_bwd_kernel<bf16_t>, dim3(pl_blocks(total)), dim3(PL_THREADS), 0, as_stream(stream), (const bf16_t*)dy, (bf16_t*)dx, N, D, H, W, Cp, Cp, fd * D, fh * H, fw * W);
  else
    hipLaunchKernelGGL(upsample2x_bwd_kernel<float>, dim3(pl_blocks(total)), dim3(PL_THREADS), 0, as_stream(stream), (const float*)dy, (float*)dx, N, D, H, W, Cp, Cp, fd * D, fh * H, fw * W);
  VFD_CHECK_LAUNCH("upsample_backward");
  return VFD_OK;
}

static int maxpool_geom(int D, int H, int W, const PoolGeom& g, int& Do, int& Ho, int& Wo) {
  VFD_REQUIRE(g.kd > 0 && g.kh > 0 && g.kw > 0 && g.sd > 0 && g.sh > 0 && g.sw > 0 && g.pd >= 0 && g.ph >= 0 && g.pw >= 0, "maxpool: bad kernel / stride / padding");
  VFD_REQUIRE(g.kd * g.kh * g.kw <= 255 && 2 * g.pd <= g.kd && 2 * g.ph <= g.kh && 2 * g.pw <= g.kw, "maxpool: at most 255 taps, padding at most half the kernel");
  VFD_REQUIRE(D + 2 * g.pd >= g.kd && H + 2 * g.ph >= g.kh && W + 2 * g.pw >= g.kw, "maxpool: kernel larger than the padded input");
  Do = (D + 2 * g.pd - g.kd) / g.sd + 1; Ho = (H + 2 * g.ph - g.kh) / g.sh + 1; Wo = (W + 2 * g.pw - g.kw) / g.sw + 1;     // floor mode
  return VFD_OK;
}

// y [N,Do,Ho,Wo,CPAD(C)], idx (uint8, same shape): window-local index of each output's maximum, for vfd_maxpool_backward
extern "C" int vfd_maxpool_forward(int dtype, const void* x, void* y, void* idx, int N, int D, int H, int W, int C, int kd, int kh, int kw,
                                   int sd, int sh, int sw, int pd, int ph, int pw, void* stream) {
  POOL_ARGS_OK("maxpool_forward");
  VFD_REQUIRE(x && y && idx, "maxpool_forward: null pointer");
  const PoolGeom g = {kd, kh, kw, sd, sh, sw, pd, ph, pw};
  int Do, Ho, Wo;
  const int rc = maxpool_geom(D, H, W, g, Do, Ho, Wo);
  if (rc != VFD_OK) return rc;
  const int Cp = cpad(C);
  const long long total = (long long)N * Do * Ho * Wo * (Cp >> 3);
  if (dtype == VFD_BF16)
    hipLaunchKernelGGL(maxpool_fwd_kernel<bf16_t>, dim3(pl_blocks(total)), dim3(PL_THREADS), 0, as_stream(stream), (const bf16_t*)x, (bf16_t*)y, (uint8_t*)idx, N, D, H, W, Cp, Do, Ho, Wo, g);
  else
    hipLaunchKernelGGL(maxpool_fwd_kernel<float>, dim3(pl_blocks(total)), dim3(PL_THREADS), 0, as_stream(stream), (const float*)x, (float*)y, (uint8_t*)idx, N, D, H, W, Cp, Do, Ho, Wo, g);
  VFD_CHECK_LAUNCH("maxpool_forward");
  return VFD_OK;
}

extern "C" int vfd_maxpool_backward(int dtype, const void* dy, const void* idx, void* dx, int N, int D, int H, int W, int C, int kd, int kh,
                                    int kw, int sd, int sh, int sw, int pd, int ph, int pw, void* stream) {
  POOL_ARGS_OK("maxpool_backward");
  VFD_REQUIRE(dy && idx && dx, "maxpool_backward: null pointer");
  const PoolGeom g = {kd, kh, kw, sd, sh, sw, pd, ph, pw};
  int Do, Ho, Wo;
  const int rc = maxpool_geom(D, H, W, g, Do, Ho, Wo);
  if (rc != VFD_OK) return rc;
  const int Cp = cpad(C);
  const long long total = (long long)N * D * H * W * (Cp >> 3);
  if (dtype == VFD_BF16)
    hipLaunchKernelGGL(maxpool_bwd_kernel<bf16_t>, dim3(pl_blocks(total)), dim3(PL_THREADS), 0, as_stream(stream), (const bf16_t*)dy, (const uint8_t*)idx, (bf16_t*)dx, N, D, H, W, Cp, Do, Ho, Wo, g);
  else
    hipLaunchKernelGGL(maxpool_bwd_kernel<float>, dim3(pl_blocks(total)), dim3(PL_THREADS), 0, as_stream(stream), (const float*)dy, (const uint8_t*)idx, (float*)dx, N, D, H, W, Cp, Do, Ho, Wo, g);
  VFD_CHECK_LAUNCH("maxpool_backward");
  return VFD_OK;
}

extern "C" int vfd_upsample2x_backward(int dtype, const void* dy, void* dx, int N, int D, int H, int W, int C, void* stream) {
  POOL_ARGS_OK("upsample2x_backward");
  VFD_REQUIRE(dy && dx, "upsample2x_backward: null pointer");
  const int Cp = cpad(C);
  const long long total = (long long)N * D * H * W * (Cp >> 3);
  if (dtype == VFD_BF16)
    hipLaunchKernelGGL(upsample2x_bwd_kernel<bf16_t>, dim3(pl_blocks(total)), dim3(PL_THREADS), 0, as_stream(stream), (const bf16_t*)dy, (bf16_t*)dx, N, D, H, W, Cp, Cp, 2 * D, 2 * H, 2 * W);
  else
    hipLaunchKernelGGL(upsample2x_bwd_kernel<float>, dim3(pl_blocks(total)), dim3(PL_THREADS), 0, as_stream(stream), (const float*)dy, (float*)dx, N, D, H, W, Cp, Cp, 2 * D, 2 * H, 2 * W);
  VFD_CHECK_LAUNCH("upsample2x_backward");
  return VFD_OK;
}

extern "C" int vfd_upsample2x_cat_forward(int dtype, const void* x, const void* skip, void* y, int N, int D, int H, int W, int Ca, int Cb,
                                          void* stream) {
  VFD_REQUIRE(dtype == VFD_F32 || dtype == VFD_BF16, "upsample2x_cat_forward: bad dtype");
  VFD_REQUIRE(x && skip && y && N > 0 && D > 0 && H > 0 && W > 0 && Ca > 0 && Cb > 0, "upsample2x_cat_forward: bad arguments");
  VFD_REQUIRE(Ca % 8 == 0, "upsample2x_cat_forward: the up-sampled tensor's channel count (%d) must be a multiple of 8", Ca);
  const int Cbp = cpad(Cb);
  const long long total = (long long)N * D * H * W * 8 * ((Ca + Cbp) >> 3);
  if (dtype == VFD_BF16)
    hipLaunchKernelGGL(upsample2x_cat_fwd_kernel<bf16_t>, dim3(pl_blocks(total)), dim3(PL_THREADS), 0, as_stream(stream), (const bf16_t*)x, (const bf16_t*)skip, (bf16_t*)y, N, D, H, W, Ca, Cbp);
  else
    hipLaunchKernelGGL(upsample2x_cat_fwd_kernel<float>, dim3(pl_blocks(total)), dim3(PL_THREADS), 0, as_stream(stream), (const float*)x, (const float*)skip, (float*)y, N, D, H, W, Ca, Cbp);
  VFD_CHECK_LAUNCH("upsample2x_cat_forward");
  return VFD_OK;
}

// gradient of the above: dx = transpose of the interpolation applied to the first Ca channels of dcat (read in place), dskip =
// channels [Ca, Ca + CPAD(Cb)) of dcat as a dense block
extern "C" int vfd_upsample2x_cat_backward(int dtype, const void* dcat, void* dx, void* dskip, int N, int D, int H, int W, int Ca, int Cb,
                                           void* stream) {
  VFD_REQUIRE(dtype == VFD_F32 || dtype == VFD_BF16, "upsample2x_cat_backward: bad dtype");
  VFD_REQUIRE(dcat && dx && dskip && N > 0 && D > 0 && H > 0 && W > 0 && Ca > 0 && Cb > 0 && Ca % 8 == 0, "upsample2x_cat_backward: bad arguments");
  const int Cbp = cpad(Cb), Ccat = Ca + Cbp;
  const long long total = (long long)N * D * H * W * (Ca >> 3);
  const long long orows = (long long)N * D * H * W * 8;
  const long long tot2 = orows * (Cbp >> 3);
  hipStream_t st = as_stream(stream);
  if (dtype == VFD_BF16) {
    hipLaunchKernelGGL(upsample2x_bwd_kernel<bf16_t>, dim3(pl_blocks(total)), dim3(PL_THREADS), 0, st, (const bf16_t*)dcat, (bf16_t*)dx, N, D, H, W, Ca, Ccat, 2 * D, 2 * H, 2 * W);
    hipLaunchKernelGGL(slice_copy_kernel<bf16_t>, dim3(pl_blocks(tot2)), dim3(PL_THREADS), 0, st, (const bf16_t*)dcat, (bf16_t*)dskip, orows, Ccat, Ca >> 3, Cbp >> 3);
  } else {
    hipLaunchKernelGGL(upsample2x_bwd_kernel<float>, dim3(pl_blocks(total)), dim3(PL_THREADS), 0, st, (const float*)dcat, (float*)dx, N, D, H, W, Ca, Ccat, 2 * D, 2 * H, 2 * W);
    hipLaunchKernelGGL(slice_copy_kernel<float>, dim3(pl_blocks(tot2)), dim3(PL_THREADS), 0, st, (const float*)dcat, (float*)dskip, orows, Ccat, Ca >> 3, Cbp >> 3);
  }
  VFD_CHECK_LAUNCH("upsample2x_cat_backward");
  return VFD_OK;
}
