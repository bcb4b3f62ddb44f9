// HBM-bound plumbing kernels: boundary layout conversion, filter packing, activations, channel concat /
// split / broadcast, dropout, bias gradient, Adam.  All are 16-byte-per-lane streaming kernels over the
// channels-last blocks [rows][Cp]; "granule" = 8 consecutive channels of one row.
#include "common.hpp"

namespace {

constexpr int EW_THREADS = 256;
static inline unsigned ew_blocks(long long n_items) {
  long long b = (n_items + EW_THREADS - 1) / EW_THREADS;
  const long long cap = 256LL * 16;  // 256 CUs x 16 resident workgroups, grid-stride beyond
  if (b > cap) b = cap;
  if (b < 1) b = 1;
  return (unsigned)b;
}

// ---- boundary layout ---------------------------------------------------------------------------------
// src f32 [N][C][S] -> dst T [N][S][Cp].  One thread per (n, s, granule); reads are coalesced along s for
// each of the 8 channels (a wave reads 8 runs of 64 consecutive floats), writes are 16 B per lane.
template <typename T>
__global__ void ncs_to_nsc_kernel(const float* __restrict__ src, T* __restrict__ dst, long long N, int C, long long S) {
  const int Cp = (C + 7) & ~7, GR = Cp >> 3;
  const long long total = N * S * GR;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const long long s = i % S;
    const long long ng = i / S;
    const int g = (int)(ng % GR);
    const long long n = ng / GR;
    float v[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int c = g * 8 + k;
      v[k] = (c < C) ? src[(n * C + c) * S + s] : 0.f;
    }
    store8(dst + (n * S + s) * Cp + g * 8, v);
  }
}
template <typename T>
__global__ void nsc_to_ncs_kernel(const T* __restrict__ src, float* __restrict__ dst, long long N, int C, long long S) {
  const int Cp = (C + 7) & ~7, GR = Cp >> 3;
  const long long total = N * S * GR;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const long long s = i % S;
    const long long ng = i / S;
    const int g = (int)(ng % GR);
    const long long n = ng / GR;
    float v[8];
    load8(src + (n * S + s) * Cp + g * 8, v);
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int c = g * 8 + k;
      if (c < C) dst[(n * C + c) * S + s] = v[k];
    }
  }
}

// [N][C][S] <-> [N][S][Cp] in the SAME dtype: the `x.view(N, C, D, H, W)` / `x.view(N, -1)` reshapes of the
// reference (models/anogan.py:76,115) expressed on channels-last blocks.
template <typename T>
__global__ void tt_ncs_to_nsc_kernel(const T* __restrict__ src, T* __restrict__ dst, long long N, int C, long long S) {
  const int Cp = (C + 7) & ~7, GR = Cp >> 3;
  const long long total = N * S * GR;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const long long s = i % S;
    const long long ng = i / S;
    const int g = (int)(ng % GR);
    const long long n = ng / GR;
    float v[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int c = g * 8 + k;
      v[k] = (c < C) ? Elem<T>::ld(src + (n * C + c) * S + s) : 0.f;
    }
    store8(dst + (n * S + s) * Cp + g * 8, v);
  }
}
template <typename T>
__global__ void tt_nsc_to_ncs_kernel(const T* __restrict__ src, T* __restrict__ dst, long long N, int C, long long S) {
  const int Cp = (C + 7) & ~7, GR = Cp >> 3;
  const long long total = N * S * GR;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const long long s = i % S;
    const long long ng = i / S;
    const int g = (int)(ng % GR);
    const long long n = ng / GR;
    float v[8];
    load8(src + (n * S + s) * Cp + g * 8, v);
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int c = g * 8 + k;
      if (c < C) Elem<T>::st(dst + (n * C + c) * S + s, v[k]);
    }
  }
}

// ---- filter packing -----------------------------------------------------------------------------------
// w f32 [A][B][T] -> packed T [R][T][Ccp] ; transpose_ab=0: R=A,Cc=B ; 1: R=B,Cc=A.  One thread per element
// of the packed block (pad channels written as zero).
// One workgroup transposes a [64 contracted channels][<= 64 taps] block of one packed row through LDS: the reads walk
// contiguous taps per channel (the whole block is contiguous when transpose_ab == 0 and T <= 64), the writes 64
// consecutive channels per tap (128 B of bf16) — instead of one 4-byte gather per packed element.
template <typename T>
__global__ __launch_bounds__(256) void pack_filter_kernel(const float* __restrict__ w, T* __restrict__ out, int A, int B, int Tn, int tr) {
  __shared__ float tile[64][65];
  const int Cc = tr ? A : B, Ccp = (Cc + 7) & ~7;
  const int r = blockIdx.x, c0 = blockIdx.y * 64, t0 = blockIdx.z * 64;
  const int nc = min(64, Cc - c0);                // real channels in this block (<= 0 for a pad-only tail)
  const int nt = min(64, Tn - t0);
  for (int i = threadIdx.x; i < 64 * nt; i += 256) {
    const int c = i / nt, t = i - c * nt;
    float v = 0.f;
    if (c < nc) {
      const int cc = c0 + c;
      const size_t ab = tr ? (size_t)cc * B + r : (size_t)r * B + cc;
      v = w[ab * Tn + t0 + t];
    }
    tile[c][t] = v;
  }
  __syncthreads();
  const int ncp = min(64, Ccp - c0);              // channels to write, padding included
  for (int i = threadIdx.x; i < 64 * nt; i += 256) {
    const int t = i >> 6, c = i & 63;
    if (c < ncp) Elem<T>::st(out + ((size_t)r * Tn + t0 + t) * Ccp + c0 + c, tile[c][t]);
  }
}

// Every filter copy of one optimiser in ONE launch (after its fused Adam step): jobs[j] = {w, packed, A, B, T, transpose,
// dtype, first block}, 8 x int64 in device memory; a workgroup finds its job by bisection over the first-block column and
// does the same 64 x 64 tile as pack_filter_kernel.
template <typename T>
__device__ __forceinline__ void pack_tile(const float* __restrict__ w, T* __restrict__ out, int A, int B, int Tn, int tr, int r, int c0, int t0,
                                          float (*tile)[65]) {
  const int Cc = tr ? A : B, Ccp = (Cc + 7) & ~7;
  const int nc = min(64, Cc - c0);
  const int nt = min(64, Tn - t0);
  for (int i = threadIdx.x; i < 64 * nt; i += 256) {
    const int c = i / nt, t = i - c * nt;
    float v = 0.f;
    if (c < nc) {
      const int cc = c0 + c;
      const size_t ab = tr ? (size_t)cc * B + r : (size_t)r * B + cc;
      v = w[ab * Tn + t0 + t];
    }
    tile[c][t] = v;
  }
  __syncthreads();
  const int ncp = min(64, Ccp - c0);
  for (int i = threadIdx.x; i < 64 * nt; i += 256) {
    const int t = i >> 6, c = i & 63;
    if (c < ncp) Elem<T>::st(out + ((size_t)r * Tn + t0 + t) * Ccp + c0 + c, tile[c][t]);
  }
}
__global__ __launch_bounds__(256) void pack_filters_kernel(const long long* __restrict__ jobs, int njobs) {
  __shared__ float tile[64][65];
  const long long b = blockIdx.x;
  int lo = 0, hi = njobs - 1;
  while (lo < hi) {                 // last job whose first block is <= b
    const int mid = (lo + hi + 1) >> 1;
    if (jobs[mid * 8 + 7] <= b) lo = mid; else hi = mid - 1;
  }
  const long long* j = jobs + lo * 8;
  const int A = (int)j[2], B = (int)j[3], Tn = (int)j[4], tr = (int)j[5];
  const int Cc = tr ? A : B, ncb = (((Cc + 7) & ~7) + 63) / 64, ntb = (Tn + 63) / 64;
  long long l = b - j[7];
  const int tb = (int)(l % ntb); l /= ntb;
  const int cb = (int)(l % ncb); l /= ncb;
  const int r = (int)l;
  if (r >= (tr ? B : A)) return;
  if (j[6] == VFD_BF16) pack_tile<bf16_t>(reinterpret_cast<const float*>(j[0]), reinterpret_cast<bf16_t*>(j[1]), A, B, Tn, tr, r, cb * 64, tb * 64, tile);
  else pack_tile<float>(reinterpret_cast<const float*>(j[0]), reinterpret_cast<float*>(j[1]), A, B, Tn, tr, r, cb * 64, tb * 64, tile);
}

// ---- y = a + b (+ c): the residual joins of models/xception.py:68 (`x += skip`) and the sum of the gradients that come back to
// a tensor with several consumers (functional.fanout), summed in float32 and rounded ONCE (torch's own bf16 adds round per term)
template <typename T>
__global__ void add_kernel(const T* __restrict__ a, const T* __restrict__ b, const T* __restrict__ c, T* __restrict__ y, long long n8) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (long long)gridDim.x * blockDim.x) {
    float u[8], v[8];
    load8(a + i * 8, u);
    load8(b + i * 8, v);
#pragma unroll
    for (int k = 0; k < 8; ++k) u[k] += v[k];
    if (c != nullptr) {
      load8(c + i * 8, v);
#pragma unroll
      for (int k = 0; k < 8; ++k) u[k] += v[k];
    }
    store8(y + i * 8, u);
  }
}

// ---- zero fill (statistics / sum pools, gradient arenas) and the scalar arithmetic of a step's loss terms: the last
// torch element-wise operators on the hot path (round 2's profiles: 24 at::native launches per ganomaly step)
__global__ void zero_kernel(uint4* __restrict__ p, long long n16, char* __restrict__ tail, int ntail) {
  const uint4 z = make_uint4(0u, 0u, 0u, 0u);
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (long long)gridDim.x * blockDim.x) p[i] = z;
  if (blockIdx.x == 0 && (int)threadIdx.x < ntail) tail[threadIdx.x] = 0;
}
// out = sum_i w[i] * *t[i]  (err_g = w_adv * err_g_adv + w_con * err_g_con + w_enc * err_g_enc, models/ganomaly.py:487-490;
// err_d = (err_d_real + err_d_fake) * 0.5, :511); summed left to right in float32 like the torch expression
__global__ void weighted_sum4_kernel(const float* t0, const float* t1, const float* t2, const float* t3, float w0, float w1, float w2,
                                     float w3, int n, float* __restrict__ out) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  float s = *t0 * w0;
  if (n > 1) s += *t1 * w1;
  if (n > 2) s += *t2 * w2;
  if (n > 3) s += *t3 * w3;
  *out = s;
}
// ... and its backward: out[i] = *g * w[i]
__global__ void scale4_kernel(const float* __restrict__ g, float w0, float w1, float w2, float w3, int n, float* __restrict__ out) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  const float v = *g;
  out[0] = v * w0;
  if (n > 1) out[1] = v * w1;
  if (n > 2) out[2] = v * w2;
  if (n > 3) out[3] = v * w3;
}

// ---- activation ------------------------------------------------------------------------------------------
template <typename T>
__global__ void act_fwd_kernel(const T* __restrict__ x, T* __restrict__ y, long long rows, int C, int act, float slope) {
  const int Cp = (C + 7) & ~7, GR = Cp >> 3;
  const long long total = rows * GR;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int g = (int)(i % GR);
    float v[8];
    load8(x + i * 8, v);
#pragma unroll
    for (int k = 0; k < 8; ++k) v[k] = (g * 8 + k < C) ? act_apply(v[k], act, slope) : 0.f;
    store8(y + i * 8, v);
  }
}
template <typename T>
__global__ void act_bwd_kernel(const T* __restrict__ y, const T* __restrict__ dy, T* __restrict__ dx, long long rows, int C,
                               int act, float slope) {
  const int Cp = (C + 7) & ~7, GR = Cp >> 3;
  const long long total = rows * GR;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int g = (int)(i % GR);
    float v[8], d[8];
    load8(y + i * 8, v);
    load8(dy + i * 8, d);
#pragma unroll
    for (int k = 0; k < 8; ++k) d[k] = (g * 8 + k < C) ? d[k] * act_grad_from_out(v[k], act, slope) : 0.f;
    store8(dx + i * 8, d);
  }
}

// ---- channel plumbing -----------------------------------------------------------------------------------
template <typename T>
__global__ void concat_kernel(const T* __restrict__ a, const T* __restrict__ b, T* __restrict__ dst, long long rows, int Ca,
                              int Cb) {
  const int Cap = (Ca + 7) & ~7, Cbp = (Cb + 7) & ~7, Cd = Ca + Cb, Cdp = (Cd + 7) & ~7, GR = Cdp >> 3;
  const long long total = rows * GR;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int g = (int)(i % GR);
    const long long r = i / GR;
    float v[8];
    if ((Ca & 7) == 0) {  // granule-aligned: whole-granule copies
      if (g * 8 < Ca) load8(a + r * Cap + g * 8, v);
      else load8(b + r * Cbp + (g * 8 - Ca), v);  // b's own pad granule lanes are zero
    } else {
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const int c = g * 8 + k;
        v[k] = (c < Ca) ? Elem<T>::ld(a + r * Cap + c) : (c < Cd ? Elem<T>::ld(b + r * Cbp + (c - Ca)) : 0.f);
      }
    }
    store8(dst + r * Cdp + g * 8, v);
  }
}
template <typename T>
__global__ void split_kernel(const T* __restrict__ src, T* __restrict__ a, T* __restrict__ b, long long rows, int Ca, int Cb) {
  const int Cap = (Ca + 7) & ~7, Cbp = (Cb + 7) & ~7, Cd = Ca + Cb, Cdp = (Cd + 7) & ~7;
  const int GA = Cap >> 3, GB = Cbp >> 3, GR = GA + GB;
  const long long total = rows * GR;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int g = (int)(i % GR);
    const long long r = i / GR;
    float v[8];
    const bool isa = g < GA;
    const int gl = isa ? g : g - GA;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int cl = gl * 8 + k;
      const int c = isa ? cl : Ca + cl;
      const bool ok = isa ? (cl < Ca) : (cl < Cb);
      v[k] = ok ? Elem<T>::ld(src + r * Cdp + c) : 0.f;
    }
    if (isa) store8(a + r * Cap + gl * 8, v);
    else store8(b + r * Cbp + gl * 8, v);
  }
}
template <typename T>
__global__ void broadcast_kernel(const T* __restrict__ src, T* __restrict__ dst, long long rows, int reps) {
  const int Cdp = (reps + 7) & ~7, GR = Cdp >> 3;
  const long long total = rows * GR;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int g = (int)(i % GR);
    const long long r = i / GR;
    const float s = Elem<T>::ld(src + r * 8);
    float v[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) v[k] = (g * 8 + k < reps) ? s : 0.f;
    store8(dst + i * 8, v);
  }
}

// ---- dropout ------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t mix_hash(uint64_t seed, uint64_t idx) {
  uint64_t z = seed + 0x9E3779B97F4A7C15ull * (idx + 1);
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  z ^= z >> 31;
  return (uint32_t)(z >> 32);
}
template <typename T>
__global__ void dropout_fwd_kernel(const T* __restrict__ x, T* __restrict__ y, uint8_t* __restrict__ mout,
                                   const uint8_t* __restrict__ min, long long n8, float p, uint64_t seed,
                                   const long long* __restrict__ step_dev) {
  if (step_dev != nullptr) seed += 0xD1B54A32D192ED03ull * (uint64_t)(*step_dev);   // device-side step counter (graph replay)
  const float scale = 1.f / (1.f - p);
  const uint32_t thr = (uint32_t)((double)p * 4294967296.0);
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (long long)gridDim.x * blockDim.x) {
    float v[8];
    load8(x + i * 8, v);
    uint8_t m[8];
    if (min != nullptr) {
      const uint2 mm = *reinterpret_cast<const uint2*>(min + i * 8);
#pragma unroll
      for (int k = 0; k < 8; ++k) m[k] = (uint8_t)(((k < 4 ? mm.x : mm.y) >> (8 * (k & 3))) & 0xff);
    } else {
#pragma unroll
      for (int k = 0; k < 8; ++k) m[k] = mix_hash(seed, (uint64_t)i * 8 + k) >= thr ? 1 : 0;
    }
    uint2 mo = make_uint2(0, 0);
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      v[k] = m[k] ? v[k] * scale : 0.f;
      if (k < 4) mo.x |= (uint32_t)m[k] << (8 * k); else mo.y |= (uint32_t)m[k] << (8 * (k - 4));
    }
    store8(y + i * 8, v);
    if (mout != nullptr) *reinterpret_cast<uint2*>(mout + i * 8) = mo;
  }
}
template <typename T>
__global__ void dropout_bwd_kernel(const T* __restrict__ dy, T* __restrict__ dx, const uint8_t* __restrict__ mask, long long n8,
                                   float p) {
  const float scale = 1.f / (1.f - p);
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (long long)gridDim.x * blockDim.x) {
    float v[8];
    load8(dy + i * 8, v);
    const uint2 mm = *reinterpret_cast<const uint2*>(mask + i * 8);
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const uint32_t m = ((k < 4 ? mm.x : mm.y) >> (8 * (k & 3))) & 0xff;
      v[k] = m ? v[k] * scale : 0.f;
    }
    store8(dx + i * 8, v);
  }
}

// ---- bias gradient: db[c] = beta*db[c] + sum_rows dy[row][c] --------------------------------------------------
// Two-stage column sum (deterministic): workgroups of TX granule-lanes x TY row-lanes write per-workgroup partials,
// a finalize kernel (8 channels x 32 lanes) folds them.
template <typename T>
__global__ __launch_bounds__(256) void colsum_partial_kernel(const T* __restrict__ dy, float* __restrict__ part, long long rows,
                                                             int C, int TX, long long rpb) {
  const int Cp = (C + 7) & ~7, GR = Cp >> 3;
  const int tx = threadIdx.x % TX, ty = threadIdx.x / TX, TY = 256 / TX;
  const int g = blockIdx.x * TX + tx;
  float s[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  if (g < GR) {
    const long long rbeg = (long long)blockIdx.y * rpb;
    long long rend = rbeg + rpb;
    if (rend > rows) rend = rows;
    for (long long r = rbeg + ty; r < rend; r += TY) {
      float v[8];
      load8(dy + r * Cp + g * 8, v);
#pragma unroll
      for (int k = 0; k < 8; ++k) s[k] += v[k];
    }
  }
  __shared__ float sh[256][8 + 1];
#pragma unroll
  for (int k = 0; k < 8; ++k) sh[threadIdx.x][k] = s[k];
  __syncthreads();
  if (ty == 0 && g < GR) {
    for (int j = 1; j < TY; ++j)
#pragma unroll
      for (int k = 0; k < 8; ++k) s[k] += sh[j * TX + tx][k];
#pragma unroll
    for (int k = 0; k < 8; ++k) part[(size_t)blockIdx.y * Cp + g * 8 + k] = s[k];
  }
}
__global__ __launch_bounds__(256) void colsum_finalize_kernel(const float* __restrict__ part, int nparts, int C, float* db,
                                                              float beta) {
  const int Cp = (C + 7) & ~7;
  const int pl = threadIdx.x & 31;
  const int c = blockIdx.x * 8 + (threadIdx.x >> 5);
  double s = 0;
  if (c < C)
    for (int j = pl; j < nparts; j += 32) s += part[(size_t)j * Cp + c];
#pragma unroll
  for (int o = 16; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
  if (pl == 0 && c < C) db[c] = (beta != 0.f) ? beta * db[c] + (float)s : (float)s;
}

// ---- Adam -----------------------------------------------------------------------------------------------------
// torch.optim.Adam semantics (no amsgrad, no weight decay):
//   m = b1 m + (1-b1) g ; v = b2 v + (1-b2) g^2 ; p -= lr/(1-b1^t) * m / (sqrt(v)/sqrt(1-b2^t) + eps)
__global__ void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                            long long n, float lr, float b1, float b2, float eps, float bc1, float bc2_sqrt, float gscale) {
  const long long n4 = n >> 2;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long long)gridDim.x * blockDim.x) {
    float4 pp = reinterpret_cast<float4*>(p)[i];
    const float4 gg = reinterpret_cast<const float4*>(g)[i];
    float4 mm = reinterpret_cast<float4*>(m)[i];
    float4 vv = reinterpret_cast<float4*>(v)[i];
    float* P = &pp.x; const float* G = &gg.x; float* M = &mm.x; float* V = &vv.x;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const float gk = G[k] * gscale;
      M[k] = b1 * M[k] + (1.f - b1) * gk;
      V[k] = b2 * V[k] + (1.f - b2) * gk * gk;
      const float denom = sqrtf(V[k]) / bc2_sqrt + eps;
      P[k] -= (lr / bc1) * (M[k] / denom);
    }
    reinterpret_cast<float4*>(p)[i] = pp;
    reinterpret_cast<float4*>(m)[i] = mm;
    reinterpret_cast<float4*>(v)[i] = vv;
  }
  // tail (n not a multiple of 4)
  const long long t0 = n4 << 2;
  const long long i = t0 + (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) {
    const float gk = g[i] * gscale;
    const float mk = b1 * m[i] + (1.f - b1) * gk;
    const float vk = b2 * v[i] + (1.f - b2) * gk * gk;
    m[i] = mk; v[i] = vk;
    p[i] -= (lr / bc1) * (mk / (sqrtf(vk) / bc2_sqrt + eps));
  }
}

}  // namespace

#define DISPATCH_DTYPE(dtype, KERNEL, grid, st, ...)                                                 \
  do {                                                                                               \
    if ((dtype) == VFD_BF16) hipLaunchKernelGGL(KERNEL<bf16_t>, grid, dim3(EW_THREADS), 0, st, __VA_ARGS__); \
    else hipLaunchKernelGGL(KERNEL<float>, grid, dim3(EW_THREADS), 0, st, __VA_ARGS__);              \
  } while (0)
#define CHECK_DTYPE(dtype, name) VFD_REQUIRE((dtype) == VFD_F32 || (dtype) == VFD_BF16, name ": bad dtype %d", (int)(dtype))

extern "C" int vfd_ncs_to_nsc(int dtype, const float* src, void* dst, int64_t N, int C, int64_t S, void* stream) {
  CHECK_DTYPE(dtype, "ncs_to_nsc");
  VFD_REQUIRE(src && dst && N > 0 && C > 0 && S > 0, "ncs_to_nsc: bad arguments");
  const long long total = (long long)N * S * (cpad(C) >> 3);
  if (dtype == VFD_BF16)
    hipLaunchKernelGGL(ncs_to_nsc_kernel<bf16_t>, dim3(ew_blocks(total)), dim3(EW_THREADS), 0, as_stream(stream), src, (bf16_t*)dst, (long long)N, C, (long long)S);
  else
    hipLaunchKernelGGL(ncs_to_nsc_kernel<float>, dim3(ew_blocks(total)), dim3(EW_THREADS), 0, as_stream(stream), src, (float*)dst, (long long)N, C, (long long)S);
  VFD_CHECK_LAUNCH("ncs_to_nsc");
  return VFD_OK;
}

extern "C" int vfd_nsc_to_ncs(int dtype, const void* src, float* dst, int64_t N, int C, int64_t S, void* stream) {
  CHECK_DTYPE(dtype, "nsc_to_ncs");
  VFD_REQUIRE(src && dst && N > 0 && C > 0 && S > 0, "nsc_to_ncs: bad arguments");
  const long long total = (long long)N * S * (cpad(C) >> 3);
  if (dtype == VFD_BF16)
    hipLaunchKernelGGL(nsc_to_ncs_kernel<bf16_t>, dim3(ew_blocks(total)), dim3(EW_THREADS), 0, as_stream(stream), (const bf16_t*)src, dst, (long long)N, C, (long long)S);
  else
    hipLaunchKernelGGL(nsc_to_ncs_kernel<float>, dim3(ew_blocks(total)), dim3(EW_THREADS), 0, as_stream(stream), (const float*)src, dst, (long long)N, C, (long long)S);
  VFD_CHECK_LAUNCH("nsc_to_ncs");
  return VFD_OK;
}

extern "C" int vfd_unflatten(int dtype, const void* src, void* dst, int64_t N, int C, int64_t S, void* stream) {
  CHECK_DTYPE(dtype, "unflatten");
  VFD_REQUIRE(src && dst && N > 0 && C > 0 && S > 0 && ((C * S) & 7) == 0, "unflatten: bad arguments (C*S must be a multiple of 8)");
  const long long total = (long long)N * S * (cpad(C) >> 3);
  if (dtype == VFD_BF16)
    hipLaunchKernelGGL(tt_ncs_to_nsc_kernel<bf16_t>, dim3(ew_blocks(total)), dim3(EW_THREADS), 0, as_stream(stream), (const bf16_t*)src, (bf16_t*)dst, (long long)N, C, (long long)S);
  else
    hipLaunchKernelGGL(tt_ncs_to_nsc_kernel<float>, dim3(ew_blocks(total)), dim3(EW_THREADS), 0, as_stream(stream), (const float*)src, (float*)dst, (long long)N, C, (long long)S);
  VFD_CHECK_LAUNCH("unflatten");
  return VFD_OK;
}

extern "C" int vfd_flatten(int dtype, const void* src, void* dst, int64_t N, int C, int64_t S, void* stream) {
  CHECK_DTYPE(dtype, "flatten");
  VFD_REQUIRE(src && dst && N > 0 && C > 0 && S > 0 && ((C * S) & 7) == 0, "flatten: bad arguments (C*S must be a multiple of 8)");
  const long long total = (long long)N * S * (cpad(C) >> 3);
  if (dtype == VFD_BF16)
    hipLaunchKernelGGL(tt_nsc_to_ncs_kernel<bf16_t>, dim3(ew_blocks(total)), dim3(EW_THREADS), 0, as_stream(stream), (const bf16_t*)src, (bf16_t*)dst, (long long)N, C, (long long)S);
  else
    hipLaunchKernelGGL(tt_nsc_to_ncs_kernel<float>, dim3(ew_blocks(total)), dim3(EW_THREADS), 0, as_stream(stream), (const float*)src, (float*)dst, (long long)N, C, (long long)S);
  VFD_CHECK_LAUNCH("flatten");
  return VFD_OK;
}

extern "C" int vfd_pack_filter(int dtype, const float* w, void* packed, int A, int B, int T, int transpose_ab, void* stream) {
  CHECK_DTYPE(dtype, "pack_filter");
  VFD_REQUIRE(w && packed && A > 0 && B > 0 && T > 0, "pack_filter: bad arguments");
  const int R = transpose_ab ? B : A, Cc = transpose_ab ? A : B;
  VFD_REQUIRE((cpad(Cc) + 63) / 64 <= 65535 && (T + 63) / 64 <= 65535, "pack_filter: %d channels / %d taps exceed the launch limits", Cc, T);
  const dim3 grid((unsigned)R, (unsigned)((cpad(Cc) + 63) / 64), (unsigned)((T + 63) / 64));
  if (dtype == VFD_BF16)
    hipLaunchKernelGGL(pack_filter_kernel<bf16_t>, grid, dim3(256), 0, as_stream(stream), w, (bf16_t*)packed, A, B, T, transpose_ab);
  else
    hipLaunchKernelGGL(pack_filter_kernel<float>, grid, dim3(256), 0, as_stream(stream), w, (float*)packed, A, B, T, transpose_ab);
  VFD_CHECK_LAUNCH("pack_filter");
  return VFD_OK;
}

extern "C" int64_t vfd_pack_filter_blocks(int A, int B, int T, int transpose_ab) {
  const int R = transpose_ab ? B : A, Cc = transpose_ab ? A : B;
  return (int64_t)R * ((cpad(Cc) + 63) / 64) * ((T + 63) / 64);
}

extern "C" int vfd_pack_filters(const int64_t* jobs_dev, int njobs, int64_t total_blocks, void* stream) {
  VFD_REQUIRE(jobs_dev != nullptr && njobs > 0 && total_blocks > 0 && total_blocks < 0x7fffffffLL, "pack_filters: bad arguments");
  hipLaunchKernelGGL(pack_filters_kernel, dim3((unsigned)total_blocks), dim3(256), 0, as_stream(stream),
                     reinterpret_cast<const long long*>(jobs_dev), njobs);
  VFD_CHECK_LAUNCH("pack_filters");
  return VFD_OK;
}

extern "C" int vfd_add(int dtype, const void* a, const void* b, const void* c, void* y, int64_t n, void* stream) {
  CHECK_DTYPE(dtype, "add");
  VFD_REQUIRE(a && b && y && n > 0 && (n & 7) == 0, "add: bad arguments (element count must be a multiple of 8: channels-last blocks are)");
  const long long n8 = n >> 3;
  if (dtype == VFD_BF16)
    hipLaunchKernelGGL(add_kernel<bf16_t>, dim3(ew_blocks(n8)), dim3(EW_THREADS), 0, as_stream(stream), (const bf16_t*)a, (const bf16_t*)b, (const bf16_t*)c, (bf16_t*)y, n8);
  else
    hipLaunchKernelGGL(add_kernel<float>, dim3(ew_blocks(n8)), dim3(EW_THREADS), 0, as_stream(stream), (const float*)a, (const float*)b, (const float*)c, (float*)y, n8);
  VFD_CHECK_LAUNCH("add");
  return VFD_OK;
}

extern "C" int vfd_zero(void* p, size_t bytes, void* stream) {
  VFD_REQUIRE(p != nullptr || bytes == 0, "zero: null pointer");
  if (bytes == 0) return VFD_OK;
  VFD_REQUIRE((reinterpret_cast<uintptr_t>(p) & 15) == 0, "zero: the buffer must be 16-byte aligned");
  const long long n16 = (long long)(bytes >> 4);
  const int ntail = (int)(bytes & 15);
  hipLaunchKernelGGL(zero_kernel, dim3(ew_blocks(n16 > 0 ? n16 : 1)), dim3(EW_THREADS), 0, as_stream(stream), reinterpret_cast<uint4*>(p), n16,
                     reinterpret_cast<char*>(p) + (n16 << 4), ntail);
  VFD_CHECK_LAUNCH("zero");
  return VFD_OK;
}

extern "C" int vfd_weighted_sum4(const float* t0, const float* t1, const float* t2, const float* t3, float w0, float w1, float w2,
                                 float w3, int n, float* out, void* stream) {
  VFD_REQUIRE(n >= 1 && n <= 4 && t0 && out && (n < 2 || t1) && (n < 3 || t2) && (n < 4 || t3), "weighted_sum4: bad arguments");
  hipLaunchKernelGGL(weighted_sum4_kernel, dim3(1), dim3(64), 0, as_stream(stream), t0, t1, t2, t3, w0, w1, w2, w3, n, out);
  VFD_CHECK_LAUNCH("weighted_sum4");
  return VFD_OK;
}

extern "C" int vfd_scale4(const float* g, float w0, float w1, float w2, float w3, int n, float* out, void* stream) {
  VFD_REQUIRE(n >= 1 && n <= 4 && g && out, "scale4: bad arguments");
  hipLaunchKernelGGL(scale4_kernel, dim3(1), dim3(64), 0, as_stream(stream), g, w0, w1, w2, w3, n, out);
  VFD_CHECK_LAUNCH("scale4");
  return VFD_OK;
}

extern "C" int vfd_act_forward(int dtype, const void* x, void* y, int64_t rows, int C, int act, float slope, void* stream) {
  CHECK_DTYPE(dtype, "act_forward");
  VFD_REQUIRE(x && y && rows > 0 && C > 0, "act_forward: bad arguments");
  const long long total = (long long)rows * (cpad(C) >> 3);
  if (dtype == VFD_BF16)
    hipLaunchKernelGGL(act_fwd_kernel<bf16_t>, dim3(ew_blocks(total)), dim3(EW_THREADS), 0, as_stream(stream), (const bf16_t*)x, (bf16_t*)y, (long long)rows, C, act, slope);
  else
    hipLaunchKernelGGL(act_fwd_kernel<float>, dim3(ew_blocks(total)), dim3(EW_THREADS), 0, as_stream(stream), (const float*)x, (float*)y, (long long)rows, C, act, slope);
  VFD_CHECK_LAUNCH("act_forward");
  return VFD_OK;
}

extern "C" int vfd_act_backward(int dtype, const void* y, const void* dy, void* dx, int64_t rows, int C, int act, float slope,
                                void* stream) {
  CHECK_DTYPE(dtype, "act_backward");
  VFD_REQUIRE(y && dy && dx && rows > 0 && C > 0, "act_backward: bad arguments");
  const long long total = (long long)rows * (cpad(C) >> 3);
  if (dtype == VFD_BF16)
    hipLaunchKernelGGL(act_bwd_kernel<bf16_t>, dim3(ew_blocks(total)), dim3(EW_THREADS), 0, as_stream(stream), (const bf16_t*)y, (const bf16_t*)dy, (bf16_t*)dx, (long long)rows, C, act, slope);
  else
    hipLaunchKernelGGL(act_bwd_kernel<float>, dim3(ew_blocks(total)), dim3(EW_THREADS), 0, as_stream(stream), (const float*)y, (const float*)dy, (float*)dx, (long long)rows, C, act, slope);
  VFD_CHECK_LAUNCH("act_backward");
  return VFD_OK;
}

extern "C" int vfd_concat_channels(int dtype, const void* a, const void* b, void* dst, int64_t rows, int Ca, int Cb, void* stream) {
  CHECK_DTYPE(dtype, "concat");
  VFD_REQUIRE(a && b && dst && rows > 0 && Ca > 0 && Cb > 0, "concat: bad arguments");
  const long long total = (long long)rows * (cpad(Ca + Cb) >> 3);
  if (dtype == VFD_BF16)
    hipLaunchKernelGGL(concat_kernel<bf16_t>, dim3(ew_blocks(total)), dim3(EW_THREADS), 0, as_stream(stream), (const bf16_t*)a, (const bf16_t*)b, (bf16_t*)dst, (long long)rows, Ca, Cb);
  else
    hipLaunchKernelGGL(concat_kernel<float>, dim3(ew_blocks(total)), dim3(EW_THREADS), 0, as_stream(stream), (const float*)a, (const float*)b, (float*)dst, (long long)rows, Ca, Cb);
  VFD_CHECK_LAUNCH("concat");
  return VFD_OK;
}

extern "C" int vfd_split_channels(int dtype, const void* src, void* a, void* b, int64_t rows, int Ca, int Cb, void* stream) {
  CHECK_DTYPE(dtype, "split");
  VFD_REQUIRE(src && a && b && rows > 0 && Ca > 0 && Cb > 0, "split: bad arguments");
  const long long total = (long long)rows * ((cpad(Ca) + cpad(Cb)) >> 3);
  if (dtype == VFD_BF16)
    hipLaunchKernelGGL(split_kernel<bf16_t>, dim3(ew_blocks(total)), dim3(EW_THREADS), 0, as_stream(stream), (const bf16_t*)src, (bf16_t*)a, (bf16_t*)b, (long long)rows, Ca, Cb);
  else
    hipLaunchKernelGGL(split_kernel<float>, dim3(ew_blocks(total)), dim3(EW_THREADS), 0, as_stream(stream), (const float*)src, (float*)a, (float*)b, (long long)rows, Ca, Cb);
  VFD_CHECK_LAUNCH("split");
  return VFD_OK;
}

extern "C" int vfd_broadcast_channel(int dtype, const void* src, void* dst, int64_t rows, int reps, void* stream) {
  CHECK_DTYPE(dtype, "broadcast");
  VFD_REQUIRE(src && dst && rows > 0 && reps > 0, "broadcast: bad arguments");
  const long long total = (long long)rows * (cpad(reps) >> 3);
  if (dtype == VFD_BF16)
    hipLaunchKernelGGL(broadcast_kernel<bf16_t>, dim3(ew_blocks(total)), dim3(EW_THREADS), 0, as_stream(stream), (const bf16_t*)src, (bf16_t*)dst, (long long)rows, reps);
  else
    hipLaunchKernelGGL(broadcast_kernel<float>, dim3(ew_blocks(total)), dim3(EW_THREADS), 0, as_stream(stream), (const float*)src, (float*)dst, (long long)rows, reps);
  VFD_CHECK_LAUNCH("broadcast");
  return VFD_OK;
}

extern "C" int vfd_dropout_forward(int dtype, const void* x, void* y, uint8_t* mask_out, const uint8_t* mask_in, int64_t n,
                                   float p, uint64_t seed, const int64_t* step_dev, void* stream) {
  CHECK_DTYPE(dtype, "dropout_forward");
  VFD_REQUIRE(x && y && n > 0 && (n & 7) == 0, "dropout_forward: n must be a positive multiple of 8");
  VFD_REQUIRE(p >= 0.f && p < 1.f, "dropout_forward: p must be in [0,1)");
  const long long n8 = n >> 3;
  if (dtype == VFD_BF16)
    hipLaunchKernelGGL(dropout_fwd_kernel<bf16_t>, dim3(ew_blocks(n8)), dim3(EW_THREADS), 0, as_stream(stream), (const bf16_t*)x, (bf16_t*)y, mask_out, mask_in, n8, p, seed, (const long long*)step_dev);
  else
    hipLaunchKernelGGL(dropout_fwd_kernel<float>, dim3(ew_blocks(n8)), dim3(EW_THREADS), 0, as_stream(stream), (const float*)x, (float*)y, mask_out, mask_in, n8, p, seed, (const long long*)step_dev);
  VFD_CHECK_LAUNCH("dropout_forward");
  return VFD_OK;
}

extern "C" int vfd_dropout_backward(int dtype, const void* dy, void* dx, const uint8_t* mask, int64_t n, float p, void* stream) {
  CHECK_DTYPE(dtype, "dropout_backward");
  VFD_REQUIRE(dy && dx && mask && n > 0 && (n & 7) == 0, "dropout_backward: bad arguments");
  const long long n8 = n >> 3;
  if (dtype == VFD_BF16)
    hipLaunchKernelGGL(dropout_bwd_kernel<bf16_t>, dim3(ew_blocks(n8)), dim3(EW_THREADS), 0, as_stream(stream), (const bf16_t*)dy, (bf16_t*)dx, mask, n8, p);
  else
    hipLaunchKernelGGL(dropout_bwd_kernel<float>, dim3(ew_blocks(n8)), dim3(EW_THREADS), 0, as_stream(stream), (const float*)dy, (float*)dx, mask, n8, p);
  VFD_CHECK_LAUNCH("dropout_backward");
  return VFD_OK;
}

constexpr int COLSUM_MAX_PARTS = 1024;
extern "C" size_t vfd_bias_grad_workspace(int C) { return (size_t)COLSUM_MAX_PARTS * cpad(C) * sizeof(float); }

extern "C" int vfd_bias_grad(int dtype, const void* dy, float* db, int64_t rows, int C, float beta, void* ws, void* stream) {
  CHECK_DTYPE(dtype, "bias_grad");
  VFD_REQUIRE(dy && db && ws && rows > 0 && C > 0, "bias_grad: bad arguments");
  const int GR = cpad(C) >> 3;
  int tx = 1;
  while (tx < GR && tx < 256) tx <<= 1;
  const int ty = 256 / tx, gx = (GR + tx - 1) / tx;
  long long gy = (rows + (long long)ty * 8 - 1) / ((long long)ty * 8);
  const long long cap = COLSUM_MAX_PARTS / gx > 0 ? COLSUM_MAX_PARTS / gx : 1;
  if (gy > cap) gy = cap;
  if (gy < 1) gy = 1;
  const long long rpb = (rows + gy - 1) / gy;
  float* part = reinterpret_cast<float*>(ws);
  dim3 grid(gx, (unsigned)gy);
  if (dtype == VFD_BF16)
    hipLaunchKernelGGL(colsum_partial_kernel<bf16_t>, grid, dim3(256), 0, as_stream(stream), (const bf16_t*)dy, part, (long long)rows, C, tx, rpb);
  else
    hipLaunchKernelGGL(colsum_partial_kernel<float>, grid, dim3(256), 0, as_stream(stream), (const float*)dy, part, (long long)rows, C, tx, rpb);
  VFD_CHECK_LAUNCH("colsum_partial");
  hipLaunchKernelGGL(colsum_finalize_kernel, dim3((C + 7) / 8), dim3(256), 0, as_stream(stream), part, (int)gy, C, db, beta);
  VFD_CHECK_LAUNCH("colsum_finalize");
  return VFD_OK;
}

// hipGraph-replayable form: the step counter lives on the device.  state = {int32 step; float bc1; float bc2_sqrt;}
__global__ void adam_prepare_kernel(int* step, float* bc, float b1, float b2) {
  const int t = *step + 1;
  *step = t;
  bc[0] = (float)(1.0 - pow((double)b1, (double)t));
  bc[1] = (float)sqrt(1.0 - pow((double)b2, (double)t));
}
__global__ void adam_dev_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                                long long n, float lr, float b1, float b2, float eps, const float* __restrict__ bc, float gscale) {
  const float bc1 = bc[0], bc2_sqrt = bc[1];
  const long long n4 = (n + 3) >> 2;   // arenas are padded to a multiple of 64 floats
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long long)gridDim.x * blockDim.x) {
    float4 pp = reinterpret_cast<float4*>(p)[i];
    const float4 gg = reinterpret_cast<const float4*>(g)[i];
    float4 mm = reinterpret_cast<float4*>(m)[i];
    float4 vv = reinterpret_cast<float4*>(v)[i];
    float* P = &pp.x; const float* G = &gg.x; float* M = &mm.x; float* V = &vv.x;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const float gk = G[k] * gscale;
      M[k] = b1 * M[k] + (1.f - b1) * gk;
      V[k] = b2 * V[k] + (1.f - b2) * gk * gk;
      P[k] -= (lr / bc1) * (M[k] / (sqrtf(V[k]) / bc2_sqrt + eps));
    }
    reinterpret_cast<float4*>(p)[i] = pp;
    reinterpret_cast<float4*>(m)[i] = mm;
    reinterpret_cast<float4*>(v)[i] = vv;
  }
}

extern "C" int vfd_adam_step_dev(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n, float lr,
                                 float beta1, float beta2, float eps, int32_t* step_dev, float* bc_dev, float grad_scale,
                                 void* stream) {
  VFD_REQUIRE(param && grad && exp_avg && exp_avg_sq && step_dev && bc_dev && n > 0 && (n & 3) == 0,
              "adam_dev: bad arguments (n must be a multiple of 4)");
  VFD_REQUIRE((((uintptr_t)param | (uintptr_t)grad | (uintptr_t)exp_avg | (uintptr_t)exp_avg_sq) & 15) == 0,
              "adam_dev: arenas must be 16-byte aligned");
  hipLaunchKernelGGL(adam_prepare_kernel, dim3(1), dim3(1), 0, as_stream(stream), step_dev, bc_dev, beta1, beta2);
  VFD_CHECK_LAUNCH("adam_prepare");
  hipLaunchKernelGGL(adam_dev_kernel, dim3(ew_blocks(n / 4)), dim3(EW_THREADS), 0, as_stream(stream), param, grad, exp_avg,
                     exp_avg_sq, (long long)n, lr, beta1, beta2, eps, bc_dev, grad_scale);
  VFD_CHECK_LAUNCH("adam_dev");
  return VFD_OK;
}

extern "C" int vfd_adam_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n, float lr, float beta1,
                             float beta2, float eps, int32_t step, float grad_scale, void* stream) {
  VFD_REQUIRE(param && grad && exp_avg && exp_avg_sq && n > 0 && step >= 1, "adam: bad arguments");
  VFD_REQUIRE((((uintptr_t)param | (uintptr_t)grad | (uintptr_t)exp_avg | (uintptr_t)exp_avg_sq) & 15) == 0,
              "adam: arenas must be 16-byte aligned");
  const double bc1 = 1.0 - pow((double)beta1, (double)step);
  const double bc2 = 1.0 - pow((double)beta2, (double)step);
  hipLaunchKernelGGL(adam_kernel, dim3(ew_blocks((n + 3) / 4)), dim3(EW_THREADS), 0, as_stream(stream), param, grad, exp_avg,
                     exp_avg_sq, (long long)n, lr, beta1, beta2, eps, (float)bc1, (float)sqrt(bc2), grad_scale);
  VFD_CHECK_LAUNCH("adam");
  return VFD_OK;
}
