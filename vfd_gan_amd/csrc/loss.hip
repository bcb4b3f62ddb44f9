// Mean-reduced losses of the GAN step: forward (two-stage deterministic reduction to ONE device float) and
// backward (element-wise gradient scaled by a DEVICE scalar, so no host sync sits between loss and backward).
//   l2_loss       lib/utils.py:59-63       mean((a-b)^2)
//   nn.L1Loss     models/ganomaly.py:438   mean(|a-b|)
//   nn.BCELoss    models/mygannet.py:267   -mean(b*max(log a,-100) + (1-b)*max(log(1-a),-100))
//   weighted_bce  lib/utils.py:65-71       a' = clamp(a, 1e-8, 1-1e-8) [in float32 the upper bound is 1.0];
//                                          -mean(b*log a' + pw*(1-b)*log(1-a'))
#include "common.hpp"

namespace {

constexpr int LOSS_THREADS = 256;
constexpr int LOSS_MAX_BLOCKS = 1024;

__device__ __forceinline__ float loss_term(int kind, float a, float b, float pw) {
  switch (kind) {
    case VFD_LOSS_L2: { const float d = a - b; return d * d; }
    case VFD_LOSS_L1: return fabsf(a - b);
    case VFD_LOSS_BCE: {
      const float la = fmaxf(logf(a), -100.f), l1a = fmaxf(logf(1.f - a), -100.f);
      return -(b * la + (1.f - b) * l1a);
    }
    default: {  // WBCE
      const float hi = 1.f - 1e-8f;  // == 1.0f in float32, as in the reference
      const float ac = fminf(fmaxf(a, 1e-8f), hi);
      return -(b * logf(ac) + pw * (1.f - b) * logf(1.f - ac));
    }
  }
}
// d term / d a  and  d term / d b
__device__ __forceinline__ void loss_grad(int kind, float a, float b, float pw, float& ga, float& gb) {
  switch (kind) {
    case VFD_LOSS_L2: { const float d = a - b; ga = 2.f * d; gb = -2.f * d; return; }
    case VFD_LOSS_L1: { const float d = a - b; const float s = d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f); ga = s; gb = -s; return; }
    case VFD_LOSS_BCE: {
      // torch: grad = (a - b) / max((1-a)*a, 1e-12)
      ga = (a - b) / fmaxf((1.f - a) * a, 1e-12f);
      gb = -(fmaxf(logf(a), -100.f) - fmaxf(logf(1.f - a), -100.f));
      return;
    }
    default: {
      const float hi = 1.f - 1e-8f;
      const bool inside = (a >= 1e-8f) && (a <= hi);  // clamp passes gradient only inside [min,max]
      const float ac = fminf(fmaxf(a, 1e-8f), hi);
      ga = inside ? -(b / ac - pw * (1.f - b) / (1.f - ac)) : 0.f;
      gb = -(logf(ac) - pw * logf(1.f - ac));
      return;
    }
  }
}

template <typename T>
__global__ __launch_bounds__(LOSS_THREADS) void loss_partial_kernel(int kind, const T* __restrict__ a, const T* __restrict__ b,
                                                                    float bconst, double* __restrict__ part, long long rows, int C,
                                                                    float pw) {
  const int Cp = (C + 7) & ~7, GR = Cp >> 3;
  const long long total = rows * GR;
  double acc = 0.0;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int g = (int)(i % GR);
    float va[8], vb[8];
    load8(a + i * 8, va);
    if (b != nullptr) load8(b + i * 8, vb);
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k)
      if (g * 8 + k < C) s += loss_term(kind, va[k], b != nullptr ? vb[k] : bconst, pw);
    acc += (double)s;
  }
  acc = wave_sum_d(acc);
  __shared__ double red[LOSS_THREADS / 64];
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    double t = 0;
    for (int w = 0; w < LOSS_THREADS / 64; ++w) t += red[w];
    part[blockIdx.x] = t;
  }
}

__global__ void loss_finalize_kernel(const double* __restrict__ part, int n, double inv_count, float* loss) {
  double t = 0;
  for (int i = threadIdx.x; i < n; i += 64) t += part[i];
  t = wave_sum_d(t);
  if (threadIdx.x == 0) *loss = (float)(t * inv_count);
}

template <typename T>
__global__ __launch_bounds__(LOSS_THREADS) void loss_backward_kernel(int kind, const T* __restrict__ a, const T* __restrict__ b,
                                                                     float bconst, const float* __restrict__ gout, T* __restrict__ ga_o,
                                                                     T* __restrict__ gb_o, long long rows, int C, float scale, float pw) {
  const int Cp = (C + 7) & ~7, GR = Cp >> 3;
  const long long total = rows * GR;
  const float s = scale * (gout != nullptr ? *gout : 1.f);
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int g = (int)(i % GR);
    float va[8], vb[8], oa[8], ob[8];
    load8(a + i * 8, va);
    if (b != nullptr) load8(b + i * 8, vb);
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      float da = 0.f, db = 0.f;
      if (g * 8 + k < C) loss_grad(kind, va[k], b != nullptr ? vb[k] : bconst, pw, da, db);
      oa[k] = da * s; ob[k] = db * s;
    }
    if (ga_o != nullptr) store8(ga_o + i * 8, oa);
    if (gb_o != nullptr) store8(gb_o + i * 8, ob);
  }
}

static inline unsigned loss_blocks(long long total) {
  long long b = (total + LOSS_THREADS - 1) / LOSS_THREADS;
  if (b > LOSS_MAX_BLOCKS) b = LOSS_MAX_BLOCKS;
  if (b < 1) b = 1;
  return (unsigned)b;
}

}  // namespace

extern "C" size_t vfd_loss_workspace(int64_t rows, int C) {
  (void)rows; (void)C;
  return (size_t)LOSS_MAX_BLOCKS * sizeof(double);
}

extern "C" int vfd_loss_forward(int kind, int dtype, const void* a, const void* b, float bconst, float* loss, int64_t rows, int C,
                                float pos_weight, void* ws, void* stream) {
  VFD_REQUIRE(kind >= VFD_LOSS_L2 && kind <= VFD_LOSS_WBCE, "loss: bad kind %d", kind);
  VFD_REQUIRE(dtype == VFD_F32 || dtype == VFD_BF16, "loss: bad dtype");
  VFD_REQUIRE(a && loss && ws && rows > 0 && C > 0, "loss: bad arguments");
  const long long total = (long long)rows * (cpad(C) >> 3);
  const unsigned nb = loss_blocks(total);
  double* part = reinterpret_cast<double*>(ws);
  if (dtype == VFD_BF16)
    hipLaunchKernelGGL(loss_partial_kernel<bf16_t>, dim3(nb), dim3(LOSS_THREADS), 0, as_stream(stream), kind, (const bf16_t*)a, (const bf16_t*)b, bconst, part, (long long)rows, C, pos_weight);
  else
    hipLaunchKernelGGL(loss_partial_kernel<float>, dim3(nb), dim3(LOSS_THREADS), 0, as_stream(stream), kind, (const float*)a, (const float*)b, bconst, part, (long long)rows, C, pos_weight);
  VFD_CHECK_LAUNCH("loss_partial");
  hipLaunchKernelGGL(loss_finalize_kernel, dim3(1), dim3(64), 0, as_stream(stream), part, (int)nb, 1.0 / ((double)rows * C), loss);
  VFD_CHECK_LAUNCH("loss_finalize");
  return VFD_OK;
}

extern "C" int vfd_loss_backward(int kind, int dtype, const void* a, const void* b, float bconst, const float* gout, void* grad_a,
                                 void* grad_b, int64_t rows, int C, float scale, float pos_weight, void* stream) {
  VFD_REQUIRE(kind >= VFD_LOSS_L2 && kind <= VFD_LOSS_WBCE, "loss_backward: bad kind %d", kind);
  VFD_REQUIRE(dtype == VFD_F32 || dtype == VFD_BF16, "loss_backward: bad dtype");
  VFD_REQUIRE(a && (grad_a || grad_b) && rows > 0 && C > 0, "loss_backward: bad arguments");
  const long long total = (long long)rows * (cpad(C) >> 3);
  long long nb = (total + LOSS_THREADS - 1) / LOSS_THREADS;
  if (nb > 4096) nb = 4096;
  const float s = scale / (float)((double)rows * C);
  if (dtype == VFD_BF16)
    hipLaunchKernelGGL(loss_backward_kernel<bf16_t>, dim3((unsigned)nb), dim3(LOSS_THREADS), 0, as_stream(stream), kind, (const bf16_t*)a, (const bf16_t*)b, bconst, gout, (bf16_t*)grad_a, (bf16_t*)grad_b, (long long)rows, C, s, pos_weight);
  else
    hipLaunchKernelGGL(loss_backward_kernel<float>, dim3((unsigned)nb), dim3(LOSS_THREADS), 0, as_stream(stream), kind, (const float*)a, (const float*)b, bconst, gout, (float*)grad_a, (float*)grad_b, (long long)rows, C, s, pos_weight);
  VFD_CHECK_LAUNCH("loss_backward");
  return VFD_OK;
}
