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

template <typename T>
__global__ void upsample2x_fwd_kernel(const T* __restrict__ x, T* __restrict__ y, int N, int D, int H, int W, int Cp) {
  const int GR = Cp >> 3, Do = 2 * D, Ho = 2 * H, Wo = 2 * W;
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
__global__ void upsample2x_bwd_kernel(const T* __restrict__ dy, T* __restrict__ dx, int N, int D, int H, int W, int Cp, int dyCp) {
  const int GR = Cp >> 3, Do = 2 * D, Ho = 2 * H, Wo = 2 * W;
  const long long total = (long long)N * D * H * W * GR;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    long long q = i;
    const int g = (int)(q % GR); q /= GR;
    const int iw = (int)(q % W); q /= W;
    const int ih = (int)(q % H); q /= H;
    const int id = (int)(q % D); q /= D;
    const int n = (int)q;
    float s[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    // src(o) in (i-1, i+1)  =>  o in [2i-2, 2i+3]  because 1/r = 2 + 1/(I-1)
    for (int od = max(0, 2 * id - 2); od <= min(Do - 1, 2 * id + 3); ++od) {
      const float wd = up_weight(od, id, D, Do);
      if (wd == 0.f) continue;
      for (int oh = max(0, 2 * ih - 2); oh <= min(Ho - 1, 2 * ih + 3); ++oh) {
        const float wh = up_weight(oh, ih, H, Ho);
        if (wh == 0.f) continue;
        for (int ow = max(0, 2 * iw - 2); ow <= min(Wo - 1, 2 * iw + 3); ++ow) {
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
    hipLaunchKernelGGL(upsample2x_fwd_kernel<bf16_t>, dim3(pl_blocks(total)), dim3(PL_THREADS), 0, as_stream(stream), (const bf16_t*)x, (bf16_t*)y, N, D, H, W, Cp);
  else
    hipLaunchKernelGGL(upsample2x_fwd_kernel<float>, dim3(pl_blocks(total)), dim3(PL_THREADS), 0, as_stream(stream), (const float*)x, (float*)y, N, D, H, W, Cp);
  VFD_CHECK_LAUNCH("upsample2x_forward");
  return VFD_OK;
}

extern "C" int vfd_upsample2x_backward(int dtype, const void* dy, void* dx, int N, int D, int H, int W, int C, void* stream) {
  POOL_ARGS_OK("upsample2x_backward");
  VFD_REQUIRE(dy && dx, "upsample2x_backward: null pointer");
  const int Cp = cpad(C);
  const long long total = (long long)N * D * H * W * (Cp >> 3);
  if (dtype == VFD_BF16)
    hipLaunchKernelGGL(upsample2x_bwd_kernel<bf16_t>, dim3(pl_blocks(total)), dim3(PL_THREADS), 0, as_stream(stream), (const bf16_t*)dy, (bf16_t*)dx, N, D, H, W, Cp, Cp);
  else
    hipLaunchKernelGGL(upsample2x_bwd_kernel<float>, dim3(pl_blocks(total)), dim3(PL_THREADS), 0, as_stream(stream), (const float*)dy, (float*)dx, N, D, H, W, Cp, Cp);
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
    hipLaunchKernelGGL(upsample2x_bwd_kernel<bf16_t>, dim3(pl_blocks(total)), dim3(PL_THREADS), 0, st, (const bf16_t*)dcat, (bf16_t*)dx, N, D, H, W, Ca, Ccat);
    hipLaunchKernelGGL(slice_copy_kernel<bf16_t>, dim3(pl_blocks(tot2)), dim3(PL_THREADS), 0, st, (const bf16_t*)dcat, (bf16_t*)dskip, orows, Ccat, Ca >> 3, Cbp >> 3);
  } else {
    hipLaunchKernelGGL(upsample2x_bwd_kernel<float>, dim3(pl_blocks(total)), dim3(PL_THREADS), 0, st, (const float*)dcat, (float*)dx, N, D, H, W, Ca, Ccat);
    hipLaunchKernelGGL(slice_copy_kernel<float>, dim3(pl_blocks(tot2)), dim3(PL_THREADS), 0, st, (const float*)dcat, (float*)dskip, orows, Ccat, Ca >> 3, Cbp >> 3);
  }
  VFD_CHECK_LAUNCH("upsample2x_cat_backward");
  return VFD_OK;
}
