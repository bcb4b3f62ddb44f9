// fp8 (OCP e4m3fn) operand preparation for the convolution kernels (BASELINE configs[4]: "fp8 weights/activations (CDNA4 fp8
// MFMA)"): per-tensor CURRENT scaling — the tensor's own max |x| is taken first (vfd_amax), the quantiser multiplies by
// 448 / amax — so nothing saturates and no amax history has to be carried between steps.  Scales live in device memory
// and are read by the consuming kernels (vfd_conv_forward_fp8): a captured step follows them.
#include "common.hpp"

namespace {

constexpr float E4M3_MAX = 448.f;

__device__ __forceinline__ float scale_of(float amax) { return amax > 0.f ? E4M3_MAX / amax : 1.f; }

template <typename T>
__global__ __launch_bounds__(256) void amax_kernel(const T* __restrict__ x, long long n8, float* __restrict__ amax) {
  float m = 0.f;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (long long)gridDim.x * blockDim.x) {
    float v[8];
    load8(x + i * 8, v);
#pragma unroll
    for (int k = 0; k < 8; ++k) m = fmaxf(m, fabsf(v[k]));
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
  __shared__ float sh[4];
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) {
    m = fmaxf(fmaxf(sh[0], sh[1]), fmaxf(sh[2], sh[3]));
    // non-negative floats order like their bit patterns; NaN (sign 0) orders above everything and so survives as NaN
    atomicMax(reinterpret_cast<unsigned int*>(amax), __float_as_uint(m));
  }
}

// [rows][Cp8] (T) -> [rows][Cp16] e4m3: one thread per 16-channel output granule (two input granules; the second may be
// past Cp8 when C's padded widths differ: zeros)
template <typename T>
__global__ __launch_bounds__(256) void quantize_fp8_kernel(const T* __restrict__ x, fp8_t* __restrict__ q, long long rows, int Cp8,
                                                           int Cp16, const float* __restrict__ amax, float* __restrict__ scale_out) {
  const float sc = amax != nullptr ? scale_of(amax[0]) : 1.f;
  if (blockIdx.x == 0 && threadIdx.x == 0 && scale_out != nullptr) scale_out[0] = sc;
  const int G = Cp16 >> 4;
  const long long total = rows * G;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const long long r = i / G;
    const int g = (int)(i - r * G);
    float a[8], b[8];
    load8(x + r * Cp8 + g * 16, a);
    if (g * 16 + 8 < Cp8) load8(x + r * Cp8 + g * 16 + 8, b);
    else {
#pragma unroll
      for (int k = 0; k < 8; ++k) b[k] = 0.f;
    }
    uint4 o;
    o.x = pack4_fp8(a[0] * sc, a[1] * sc, a[2] * sc, a[3] * sc);
    o.y = pack4_fp8(a[4] * sc, a[5] * sc, a[6] * sc, a[7] * sc);
    o.z = pack4_fp8(b[0] * sc, b[1] * sc, b[2] * sc, b[3] * sc);
    o.w = pack4_fp8(b[4] * sc, b[5] * sc, b[6] * sc, b[7] * sc);
    *reinterpret_cast<uint4*>(q + r * Cp16 + g * 16) = o;
  }
}

__global__ __launch_bounds__(256) void amax_f32_flat_kernel(const float* __restrict__ x, long long n, float* __restrict__ amax) {
  float m = 0.f;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) m = fmaxf(m, fabsf(x[i]));
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
  if ((threadIdx.x & 63) == 0) atomicMax(reinterpret_cast<unsigned int*>(amax), __float_as_uint(m));
}

// packed[r][t][c] (e4m3, c padded to 16) = w[..] * scale, same index map as pack_filter_kernel (elementwise.hip)
__global__ __launch_bounds__(256) void pack_filter_fp8_kernel(const float* __restrict__ w, fp8_t* __restrict__ out, int A, int B, int Tn,
                                                              int tr, const float* __restrict__ amax, float* __restrict__ scale_out) {
  const float sc = scale_of(amax[0]);
  if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) scale_out[0] = sc;
  const int Cc = tr ? A : B, Ccp = (Cc + 15) & ~15;
  const int r = blockIdx.x;
  for (int i = blockIdx.y * 256 + threadIdx.x; i < Tn * Ccp; i += gridDim.y * 256) {
    const int t = i / Ccp, c = i - t * Ccp;
    float v = 0.f;
    if (c < Cc) {
      const size_t ab = tr ? (size_t)c * B + r : (size_t)r * B + c;
      v = w[ab * Tn + t] * sc;
    }
    Elem<fp8_t>::st(out + ((size_t)r * Tn + t) * Ccp + c, v);
  }
}

__global__ void dequantize_fp8_kernel(const fp8_t* __restrict__ q, float* __restrict__ y, long long n, const float* __restrict__ scale) {
  const float inv = 1.f / scale[0];
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) y[i] = Elem<fp8_t>::ld(q + i) * inv;
}

}  // namespace



extern "C" int vfd_amax(int dtype, const void* x, int64_t rows, int C, float* amax, void* stream) {
  VFD_REQUIRE(dtype == VFD_F32 || dtype == VFD_BF16, "amax: bad dtype");
  VFD_REQUIRE(x && amax && rows > 0 && C > 0, "amax: bad arguments");
  hipStream_t st = as_stream(stream);
  if (hipMemsetAsync(amax, 0, sizeof(float), st) != hipSuccess) { vfd_set_error("amax: memset failed"); return VFD_ELAUNCH; }
  const long long n8 = (long long)rows * (cpad(C) >> 3);
  long long nb = (n8 + 255) / 256;
  if (nb > 2048) nb = 2048;
  if (dtype == VFD_BF16) hipLaunchKernelGGL(amax_kernel<bf16_t>, dim3((unsigned)nb), dim3(256), 0, st, (const bf16_t*)x, n8, amax);
  else hipLaunchKernelGGL(amax_kernel<float>, dim3((unsigned)nb), dim3(256), 0, st, (const float*)x, n8, amax);
  VFD_CHECK_LAUNCH("amax");
  return VFD_OK;
}

extern "C" int vfd_quantize_fp8(int dtype, const void* x, void* q, int64_t rows, int C, const float* amax, float* scale_out,
                                void* stream) {
  VFD_REQUIRE(dtype == VFD_F32 || dtype == VFD_BF16, "quantize_fp8: bad dtype");
  VFD_REQUIRE(x && q && rows > 0 && C > 0, "quantize_fp8: bad arguments");
  VFD_REQUIRE((((uintptr_t)x | (uintptr_t)q) & 15) == 0, "quantize_fp8: tensors must be 16-byte aligned");
  const long long total = (long long)rows * (cpad16(C) >> 4);
  long long nb = (total + 255) / 256;
  if (nb > 4096) nb = 4096;
  hipStream_t st = as_stream(stream);
  if (dtype == VFD_BF16)
    hipLaunchKernelGGL(quantize_fp8_kernel<bf16_t>, dim3((unsigned)nb), dim3(256), 0, st, (const bf16_t*)x, (fp8_t*)q, (long long)rows, cpad(C), cpad16(C), amax, scale_out);
  else
    hipLaunchKernelGGL(quantize_fp8_kernel<float>, dim3((unsigned)nb), dim3(256), 0, st, (const float*)x, (fp8_t*)q, (long long)rows, cpad(C), cpad16(C), amax, scale_out);
  VFD_CHECK_LAUNCH("quantize_fp8");
  return VFD_OK;
}

extern "C" int vfd_pack_filter_fp8(const float* w, void* packed, int A, int B, int T, int transpose_ab, float* amax, float* scale_out,
                                   void* stream) {
  VFD_REQUIRE(w && packed && amax && scale_out && A > 0 && B > 0 && T > 0, "pack_filter_fp8: bad arguments");
  hipStream_t st = as_stream(stream);
  if (hipMemsetAsync(amax, 0, sizeof(float), st) != hipSuccess) { vfd_set_error("pack_filter_fp8: memset failed"); return VFD_ELAUNCH; }
  const long long n = (long long)A * B * T;
  long long nb = (n + 255) / 256;
  if (nb > 1024) nb = 1024;
  hipLaunchKernelGGL(amax_f32_flat_kernel, dim3((unsigned)nb), dim3(256), 0, st, w, n, amax);
  VFD_CHECK_LAUNCH("pack_filter_fp8(amax)");
  const int R = transpose_ab ? B : A, Cc = transpose_ab ? A : B;
  const long long per = (long long)T * cpad16(Cc);
  int gy = (int)((per + 255) / 256);
  if (gy > 64) gy = 64;
  hipLaunchKernelGGL(pack_filter_fp8_kernel, dim3((unsigned)R, (unsigned)gy), dim3(256), 0, st, w, (fp8_t*)packed, A, B, T, transpose_ab,
                     (const float*)amax, scale_out);
  VFD_CHECK_LAUNCH("pack_filter_fp8");
  return VFD_OK;
}

extern "C" int vfd_dequantize_fp8(const void* q, float* y, int64_t n, const float* scale, void* stream) {
  VFD_REQUIRE(q && y && scale && n > 0, "dequantize_fp8: bad arguments");
  long long nb = (n + 255) / 256;
  if (nb > 4096) nb = 4096;
  hipLaunchKernelGGL(dequantize_fp8_kernel, dim3((unsigned)nb), dim3(256), 0, as_stream(stream), (const fp8_t*)q, y, (long long)n, scale);
  VFD_CHECK_LAUNCH("dequantize_fp8");
  return VFD_OK;
}
