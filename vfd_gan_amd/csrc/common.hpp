// Shared device/host helpers for libvfdgan_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>

#include "../../include/vfdgan_hip.h"

typedef uint16_t bf16_t;  // bfloat16 bit pattern
struct fp8_t { uint8_t v; };   // OCP e4m3fn bit pattern (gfx950's native fp8: max 448, no infinities)

typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
typedef __attribute__((ext_vector_type(2))) uint32_t u32x2;

// ---- error plumbing -------------------------------------------------------------------------------
void vfd_set_error(const char* fmt, ...);
#define VFD_REQUIRE(cond, ...)       \
  do {                               \
    if (!(cond)) {                   \
      vfd_set_error(__VA_ARGS__);    \
      return VFD_EINVAL;             \
    }                                \
  } while (0)
#define VFD_CHECK_LAUNCH(name)                                        \
  do {                                                                \
    hipError_t e__ = hipGetLastError();                               \
    if (e__ != hipSuccess) {                                          \
      vfd_set_error("%s: %s", name, hipGetErrorString(e__));          \
      return VFD_ELAUNCH;                                             \
    }                                                                 \
  } while (0)

// conv_small.hip: 1 = handled (or, with query, would be handled), 0 = not a thin-channel shape, < 0 = launch error
int vfd_conv_small_try(const vfd_conv_desc* d, const void* x, const void* packed, const float* bias, void* y, double* stats,
                       bool query, hipStream_t st);

// Gradient hand-over carried by the epilogue of a data-gradient convolution (conv_epilogue.hpp): the consumer layer's
// dgrad leaves already multiplied by the derivative of the producer's activation.
//   src only            : y *= act'(src), src = the producer's activation OUTPUT (conv -> act -> conv chains)
//   src + bn_mean ...   : src = the producer BatchNorm's INPUT x; with xh = (x - mean) * rstd the epilogue stores
//                         g = y * act'(gamma * xh + beta) and adds the per-channel sums of g and g * xh into bn_sums
//                         ([VFD_STATS_REPLICAS][2][Cop], zeroed by the caller): BatchNorm's backward reduce pass
struct MulP {
  const void* src;
  int act;
  float slope;
  const float* bn_mean;
  const float* bn_rstd;
  const float* bn_gamma;   // null: 1
  const float* bn_beta;    // null: 0
  float* bn_sums;
};
static inline MulP no_mul() { MulP m; m.src = nullptr; m.act = 0; m.slope = 0.f; m.bn_mean = m.bn_rstd = m.bn_gamma = m.bn_beta = nullptr; m.bn_sums = nullptr; return m; }

// conv_halo.hip: same return convention; unit-input-stride layers with <= 64 output channels (bf16)
int vfd_conv_halo_try(const vfd_conv_desc* d, const void* x, const void* packed, const float* bias, void* y, double* stats,
                      const MulP& mul, bool query, hipStream_t st);

// conv_wgrad_halo.hip: halo-tiled filter gradient of stride-1 k(1|3)x3x3 layers (bf16); 1 = eligible / launched
struct WhGeomOut { int Cs, Cg, nrt, nct, nhb, nwb, nsplit, ncols, nkc, g32; long long nblocks, per_split; };
int vfd_wgrad_halo_geom(const vfd_conv_desc* d, int* nsplit, size_t* bytes, WhGeomOut* out);
int vfd_wgrad_halo_launch(const vfd_conv_desc* d, const void* x, const void* dy, void* ws, hipStream_t st);

static inline int cpad(int c) { return (c + 7) & ~7; }
static inline int cpad16(int c) { return (c + 15) & ~15; }      // fp8 tensors: 16 channels per 16-byte granule
static inline int64_t cdiv64(int64_t a, int64_t b) { return (a + b - 1) / b; }

// ---- division by a launch-constant: q = floor(m / d) for 0 <= m < 2^31 --------------------------------------
// M = ceil(2^p / d), p = 31 + ceil(log2 d): m*M >> p is exact for every m < 2^31 (error term M*d - 2^p < 2^(p-31)).
struct FastDiv {
  uint32_t magic;
  uint32_t shift;
  uint32_t d;
};
static inline FastDiv make_fastdiv(uint32_t d) {
  FastDiv f;
  if (d == 0) d = 1;
  uint32_t l = 0;
  while ((1ull << l) < d) ++l;
  const uint32_t p = 31 + l;
  f.magic = (uint32_t)(((1ull << p) + d - 1) / d);
  f.shift = p;
  f.d = d;
  return f;
}
__device__ __forceinline__ uint32_t fdiv(uint32_t m, const FastDiv& f) {
  return (uint32_t)(((unsigned long long)m * f.magic) >> f.shift);
}
// q = m / d, r = m % d
__device__ __forceinline__ void fdivmod(uint32_t m, const FastDiv& f, uint32_t& q, uint32_t& r) {
  q = fdiv(m, f);
  r = m - q * f.d;
}

// ---- per-dimension description of the taps of one output class of a (transposed) convolution -----------------
//     regular    : in = q*s - p + t            (t = tap, all k taps)          out = q
//     transposed : in = q + c0 - t,  k = k0 + t*s, k0 = (r+p)%s, c0 = (r+p-k0)/s,  out = q*s + r
struct DimClass {
  int nk;   // number of taps
  int k0;   // first filter index
  int ks;   // filter index step
  int c0;   // input coordinate offset
  int cs;   // input coordinate step per tap (+1 regular, -1 transposed)
  int a;    // input coordinate multiplier of q
  int so;   // output coordinate multiplier of q
  int r;    // output coordinate offset
  int Q;    // number of q along this dim
  FastDiv fq;  // division by Q (pixel index decomposition)
};

static inline DimClass make_dim_host(int transposed, int r, int k, int s, int p, int O) {
  DimClass d;
  if (!transposed) {
    d.nk = k; d.k0 = 0; d.ks = 1; d.c0 = -p; d.cs = 1; d.a = s; d.so = 1; d.r = 0; d.Q = O;
  } else {
    d.k0 = (r + p) % s;
    d.nk = (d.k0 < k) ? (k - d.k0 + s - 1) / s : 0;
    d.ks = s;
    d.c0 = (r + p - d.k0) / s;
    d.cs = -1;
    d.a = 1;
    d.so = s;
    d.r = r;
    d.Q = (O > r) ? (O - r + s - 1) / s : 0;
  }
  d.fq = make_fastdiv((uint32_t)d.Q);
  return d;
}


// ---- bf16 <-> f32 ---------------------------------------------------------------------------------
__device__ __forceinline__ float bf2f(bf16_t v) { return __uint_as_float(((uint32_t)v) << 16); }
__device__ __forceinline__ bf16_t f2bf(float f) {
  // plain cast: hipcc emits v_cvt_pk_bf16_f32 (RNE, NaN-preserving) on gfx950
  __bf16 b = (__bf16)f;
  return __builtin_bit_cast(bf16_t, b);
}
// (two single conversions + shift + or: kept for the fp8 256 x 256 tile, which sits exactly at its 128 registers and spills one
// with the packed form below)
__device__ __forceinline__ uint32_t pack2bf_2cvt(float lo, float hi) {
  return (uint32_t)f2bf(lo) | ((uint32_t)f2bf(hi) << 16);
}
__device__ __forceinline__ uint32_t pack2bf(float lo, float hi) {
  // ONE v_cvt_pk_bf16_f32 for the pair (found in the emitted ISA, round 3: the epilogues converted every value on its own)
  typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
  typedef float f32x2_t __attribute__((ext_vector_type(2)));
  const f32x2_t v = {lo, hi};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2_t));
}

template <typename T> struct Elem;
template <> struct Elem<float> {
  static constexpr int VEC = 4;  // elements per 16 bytes
  __device__ static __forceinline__ float ld(const float* p) { return *p; }
  __device__ static __forceinline__ void st(float* p, float v) { *p = v; }
};
template <> struct Elem<bf16_t> {
  static constexpr int VEC = 8;
  __device__ static __forceinline__ float ld(const bf16_t* p) { return bf2f(*p); }
  __device__ static __forceinline__ void st(bf16_t* p, float v) { *p = f2bf(v); }
};
// e4m3 through the hardware converters (v_cvt_pk_fp8_f32: round to nearest even, saturating at +-448; NaN stays NaN)
__device__ __forceinline__ uint32_t pack4_fp8(float a, float b, float c, float d) {
  int w = 0;
  w = __builtin_amdgcn_cvt_pk_fp8_f32(a, b, w, false);
  w = __builtin_amdgcn_cvt_pk_fp8_f32(c, d, w, true);
  return (uint32_t)w;
}
template <> struct Elem<fp8_t> {
  static constexpr int VEC = 16;
  __device__ static __forceinline__ float ld(const fp8_t* p) { return __builtin_amdgcn_cvt_f32_fp8((int)p->v, 0); }
  __device__ static __forceinline__ void st(fp8_t* p, float v) { p->v = (uint8_t)(pack4_fp8(v, 0.f, 0.f, 0.f) & 0xffu); }
};
// a convolution's OUTPUT element type for an operand type: fp8 operands accumulate in f32 and leave as bf16 (the
// BatchNorm / activation pass that follows re-quantises)
template <typename T> struct OutOf { typedef T type; };
template <> struct OutOf<fp8_t> { typedef bf16_t type; };

// 8 consecutive channels (one CPAD granule) <-> 8 floats
__device__ __forceinline__ void load8(const float* p, float (&v)[8]) {
  const float4 a = *reinterpret_cast<const float4*>(p);
  const float4 b = *reinterpret_cast<const float4*>(p + 4);
  v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
}
__device__ __forceinline__ void load8(const bf16_t* p, float (&v)[8]) {
  const uint4 a = *reinterpret_cast<const uint4*>(p);
  v[0] = __uint_as_float(a.x << 16); v[1] = __uint_as_float(a.x & 0xffff0000u);
  v[2] = __uint_as_float(a.y << 16); v[3] = __uint_as_float(a.y & 0xffff0000u);
  v[4] = __uint_as_float(a.z << 16); v[5] = __uint_as_float(a.z & 0xffff0000u);
  v[6] = __uint_as_float(a.w << 16); v[7] = __uint_as_float(a.w & 0xffff0000u);
}
__device__ __forceinline__ void store8(float* p, const float (&v)[8]) {
  *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
  *reinterpret_cast<float4*>(p + 4) = make_float4(v[4], v[5], v[6], v[7]);
}
__device__ __forceinline__ void store8(bf16_t* p, const float (&v)[8]) {
  uint4 a;
  a.x = pack2bf(v[0], v[1]); a.y = pack2bf(v[2], v[3]); a.z = pack2bf(v[4], v[5]); a.w = pack2bf(v[6], v[7]);
  *reinterpret_cast<uint4*>(p) = a;
}

// ---- activations -----------------------------------------------------------------------------------
__device__ __forceinline__ float act_apply(float x, int act, float slope) {
  switch (act) {
    case VFD_ACT_LRELU: return x > 0.f ? x : x * slope;
    case VFD_ACT_SIGMOID: return 1.f / (1.f + __expf(-x));
    case VFD_ACT_TANH: return tanhf(x);
    default: return x;
  }
}
// bf16-output kernels: v_exp_f32 / v_rcp_f32 forms (relative error ~1e-6, far inside half a bf16 ulp = 2^-9)
__device__ __forceinline__ float fast_sigmoid(float x) { return __builtin_amdgcn_rcpf(1.f + __expf(-x)); }
__device__ __forceinline__ float fast_tanh(float x) {
  x = fminf(fmaxf(x, -15.f), 15.f);
  return 1.f - 2.f * __builtin_amdgcn_rcpf(__expf(2.f * x) + 1.f);
}
// the same over a register array with ONE (wave-uniform) dispatch: inside unrolled epilogue loops the per-value switch
// above is replicated per element, branches included
template <bool FAST, int N>
__device__ __forceinline__ void act_apply_n(float (&v)[N], int act, float slope) {
  if (act == VFD_ACT_LRELU) {
#pragma unroll
    for (int i = 0; i < N; ++i) v[i] = v[i] > 0.f ? v[i] : v[i] * slope;
  } else if (act == VFD_ACT_SIGMOID) {
#pragma unroll
    for (int i = 0; i < N; ++i) v[i] = FAST ? fast_sigmoid(v[i]) : 1.f / (1.f + __expf(-v[i]));
  } else if (act == VFD_ACT_TANH) {
#pragma unroll
    for (int i = 0; i < N; ++i) v[i] = FAST ? fast_tanh(v[i]) : tanhf(v[i]);
  }
}
// derivative expressed from the OUTPUT y = act(x)
__device__ __forceinline__ float act_grad_from_out(float y, int act, float slope) {
  switch (act) {
    case VFD_ACT_LRELU: return y > 0.f ? 1.f : slope;  // slope > 0 keeps the sign; slope == 0: y==0 -> 0
    case VFD_ACT_SIGMOID: return y * (1.f - y);
    case VFD_ACT_TANH: return 1.f - y * y;
    default: return 1.f;
  }
}
// derivative expressed from the INPUT x (pre-activation)
__device__ __forceinline__ float act_grad_from_in(float x, int act, float slope) {
  switch (act) {
    case VFD_ACT_LRELU: return x > 0.f ? 1.f : slope;
    case VFD_ACT_SIGMOID: { const float y = 1.f / (1.f + __expf(-x)); return y * (1.f - y); }
    case VFD_ACT_TANH: { const float y = tanhf(x); return 1.f - y * y; }
    default: return 1.f;
  }
}

// ---- wave / block reductions -------------------------------------------------------------------------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
// Sum over the 16 lanes of a DPP row (lanes with the same lane >> 4: the 16 pixels that hold one channel quad in the MFMA
// accumulator layout); every lane of the row receives it.  Four rotate-and-add steps on the VALU (v_add_f32 with a DPP row_ror
// operand) instead of the four ds_bpermute_b32 round trips through the LDS crossbar that __shfl_xor lowers to (128 of them per
// wave and tile in the statistics epilogue).  Association: ((x_i + x_{i+8}) + (x_{i+4} + x_{i+12})) + ... — fixed.
__device__ __forceinline__ float row16_sum(float v) {
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x128, 0xf, 0xf, false));      // row_ror:8
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x124, 0xf, 0xf, false));      // row_ror:4
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x122, 0xf, 0xf, false));      // row_ror:2
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x121, 0xf, 0xf, false));      // row_ror:1
  return v;
}
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// ---- LDS-DMA issued from inline asm ---------------------------------------------------------------------
// global_load_lds_dwordx4: lane l's 16 bytes at `gsrc` (per-lane source) land at LDS byte address lds_dst + 16*l
// (wave-uniform destination, passed in M0).  The ring kernels count their DMAs by hand (s_waitcnt vmcnt(N) across a raw
// s_barrier); issued through __builtin_amdgcn_global_load_lds the compiler ALSO tracks them and, for LDS reads it cannot
// disambiguate (the ds_read_b64_tr_b16 builtin carries no alias information), emits `s_waitcnt vmcnt(0)` in front of
// the first read of every K-step, which drains the ring (found by tools/isa_audit.py in every bf16 conv_wgrad
// instantiation of round 1).  From inline asm the DMA is invisible to that bookkeeping; the "memory" clobber keeps the
// compiler's own LDS accesses on their side of the statement.  M0 is written in the same statement that reads it
// (the compiler does not preserve M0 around asm); the audit checks that the kernel has no other use of M0.
typedef __attribute__((address_space(3))) char lds_char_t;
__device__ __forceinline__ uint32_t lds_addr(const void* p) {
  return (uint32_t)(uintptr_t)(lds_char_t*)p;
}
__device__ __forceinline__ void dma16_to_lds(const void* gsrc, uint32_t lds_dst_uniform) {
  // readfirstlane: a no-op for a value the compiler already holds in an SGPR; where it cannot prove uniformity it would
  // otherwise hand the "s" operand a VGPR
  const uint32_t dst = __builtin_amdgcn_readfirstlane(lds_dst_uniform);
  asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(gsrc), "s"(dst) : "memory");
}

static inline hipStream_t as_stream(void* s) { return reinterpret_cast<hipStream_t>(s); }
