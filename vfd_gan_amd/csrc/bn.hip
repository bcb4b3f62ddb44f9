// Training-mode BatchNorm fused with the activation that follows it, on channels-last blocks [rows][Cp].
// HBM-bound: forward = 1 read (statistics) + 1 read + 1 write (normalise+activate); the statistics read
// disappears when the producing convolution accumulates sum / sum-of-squares in its epilogue.
// Backward = 2 reads (reduce) + 2 reads + 1 write (apply).
//
// Thread layout for every kernel here: a workgroup is TX granule-lanes x TY row-lanes (TX*TY = 256); a thread
// owns ONE granule (8 channels, 16/32 bytes) and walks rows, so per-channel parameters are loaded once.
// Reductions are two-stage and deterministic: workgroup partials in a workspace, then a finalize kernel.
#include "common.hpp"
#include <stdlib.h>

namespace {

struct Tiling {
  int TX, TY, gx, gy;
  long long rows_per_block;
};
// rows_per_thread: how many rows of its 8-channel granule a thread walks (streaming kernels: ROWS_U, one unrolled pass;
// fewer when that leaves under ~16 workgroups per CU, so that small tensors still fill the chip and large ones end
// without a long tail)
static Tiling make_tiling(long long rows, int C, int max_blocks, int rows_per_thread = 8) {
  Tiling t;
  const int GR = cpad(C) >> 3;
  int tx = 1;
  while (tx < GR && tx < 256) tx <<= 1;
  t.TX = tx; t.TY = 256 / tx;
  t.gx = (GR + tx - 1) / tx;
  long long gy = (rows + (long long)t.TY * rows_per_thread - 1) / ((long long)t.TY * rows_per_thread);
  long long cap = max_blocks / t.gx > 0 ? max_blocks / t.gx : 1;
  if (cap > 65535) cap = 65535;
  if (gy > cap) gy = cap;
  if (gy < 1) gy = 1;
  t.gy = (int)gy;
  t.rows_per_block = (rows + gy - 1) / gy;
  return t;
}
// streaming kernels: a fixed number of resident workgroups that each walk many rows — the per-workgroup preamble (a
// dependent chain of per-channel parameter loads) costs about as much as streaming 8 rows, so short-lived workgroups
// lose more to it than they gain in balance (measured: 16 K workgroups of 1-2 rows per thread ran the step 20 % slower)
static Tiling make_stream_tiling(long long rows, int C) {
  static const int nb = getenv("VFD_BN_STREAM_BLOCKS") ? atoi(getenv("VFD_BN_STREAM_BLOCKS")) : 512;
  return make_tiling(rows, C, nb, 4);
}
constexpr int BN_MAX_BLOCKS = 2048;  // partial workgroups per reduction

// ---- statistics: per-workgroup (count, mean, M2) per channel ----------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void bn_partial_kernel(const T* __restrict__ x, float* __restrict__ part, long long rows, int C,
                                                         int TX, long long rpb) {
  const int Cp = (C + 7) & ~7, GR = Cp >> 3;
  const int tx = threadIdx.x % TX, ty = threadIdx.x / TX, TY = 256 / TX;
  const int g = blockIdx.x * TX + tx;
  const long long rbeg = (long long)blockIdx.y * rpb;
  long long rend = rbeg + rpb;
  if (rend > rows) rend = rows;
  float cnt = 0.f, shift[8], s1[8], s2[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) { shift[k] = 0.f; s1[k] = 0.f; s2[k] = 0.f; }
  if (g < GR) {
    bool first = true;
    for (long long r = rbeg + ty; r < rend; r += TY) {
      float v[8];
      load8(x + r * Cp + g * 8, v);
      if (first) {
#pragma unroll
        for (int k = 0; k < 8; ++k) shift[k] = v[k];
        first = false;
      }
#pragma unroll
      for (int k = 0; k < 8; ++k) { const float d = v[k] - shift[k]; s1[k] += d; s2[k] += d * d; }
      cnt += 1.f;
    }
  }
  // per-thread (count, mean, M2), merged over the TY row-lanes through LDS with Chan's formula
  __shared__ float sh_cnt[256];
  __shared__ float sh_mean[256][8 + 1];
  __shared__ float sh_m2[256][8 + 1];
  sh_cnt[threadIdx.x] = cnt;
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const float mu = cnt > 0.f ? s1[k] / cnt : 0.f;
    sh_mean[threadIdx.x][k] = shift[k] + mu;
    sh_m2[threadIdx.x][k] = cnt > 0.f ? s2[k] - s1[k] * mu : 0.f;
  }
  __syncthreads();
  if (ty == 0 && g < GR) {
    float n = sh_cnt[tx], mean[8], m2[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) { mean[k] = sh_mean[tx][k]; m2[k] = sh_m2[tx][k]; }
    for (int j = 1; j < TY; ++j) {
      const int o = j * TX + tx;
      const float nb = sh_cnt[o];
      if (nb > 0.f) {
        const float nt = n + nb;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          const float d = sh_mean[o][k] - mean[k];
          mean[k] += d * (nb / nt);
          m2[k] += sh_m2[o][k] + d * d * (n * nb / nt);
        }
        n = nt;
      }
    }
    float* dst = part + (size_t)blockIdx.y * 3 * Cp;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      dst[g * 8 + k] = n;
      dst[Cp + g * 8 + k] = mean[k];
      dst[2 * Cp + g * 8 + k] = m2[k];
    }
  }
}

__device__ __forceinline__ void bn_write_stats(int c, double n, double mean, double m2, float eps, float momentum, float* mean_o,
                                               float* rstd_o, float* rmean, float* rvar) {
  const double var = n > 0 ? m2 / n : 0.0;
  mean_o[c] = (float)mean;
  rstd_o[c] = (float)(1.0 / sqrt(var + (double)eps));
  if (rmean != nullptr) rmean[c] = (1.f - momentum) * rmean[c] + momentum * (float)mean;
  if (rvar != nullptr) {
    const double unb = n > 1 ? m2 / (n - 1) : var;
    rvar[c] = (1.f - momentum) * rvar[c] + momentum * (float)unb;
  }
}

// One workgroup per 32 channels: 8 part-lanes per channel walk the workgroup partials, Chan-merge in LDS.
__global__ __launch_bounds__(256) void bn_finalize_kernel(const float* __restrict__ part, int nparts, int C, float eps,
                                                          float momentum, float* mean_o, float* rstd_o, float* rmean,
                                                          float* rvar, long long* nbt) {
  if (nbt != nullptr && blockIdx.x == 0 && threadIdx.x == 0) *nbt += 1;      // BatchNorm.num_batches_tracked
  const int Cp = (C + 7) & ~7;
  const int cl = threadIdx.x & 31, pl = threadIdx.x >> 5;
  const int c = blockIdx.x * 32 + cl;
  double n = 0, mean = 0, m2 = 0;
  if (c < C) {
    for (int j = pl; j < nparts; j += 8) {
      const float* src = part + (size_t)j * 3 * Cp;
      const double nb = src[c];
      if (nb > 0) {
        const double nt = n + nb, d = (double)src[Cp + c] - mean;
        mean += d * (nb / nt);
        m2 += (double)src[2 * Cp + c] + d * d * (n * nb / nt);
        n = nt;
      }
    }
  }
  __shared__ double sh[3][8][32];
  sh[0][pl][cl] = n; sh[1][pl][cl] = mean; sh[2][pl][cl] = m2;
  __syncthreads();
  if (pl == 0 && c < C) {
    for (int j = 1; j < 8; ++j) {
      const double nb = sh[0][j][cl];
      if (nb > 0) {
        const double nt = n + nb, d = sh[1][j][cl] - mean;
        mean += d * (nb / nt);
        m2 += sh[2][j][cl] + d * d * (n * nb / nt);
        n = nt;
      }
    }
    bn_write_stats(c, n, mean, m2, eps, momentum, mean_o, rstd_o, rmean, rvar);
  }
}

__global__ void bn_from_sums_kernel(const double* __restrict__ sums, long long rows, int C, float eps, float momentum,
                                    float* mean_o, float* rstd_o, float* rmean, float* rvar, long long* nbt) {
  if (nbt != nullptr && blockIdx.x == 0 && threadIdx.x == 0) *nbt += 1;      // BatchNorm.num_batches_tracked
  const int Cp = (C + 7) & ~7;
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  double s1 = 0, s2 = 0;
  for (int r = 0; r < VFD_STATS_REPLICAS; ++r) {      // replica rows written by the conv epilogues
    s1 += sums[(size_t)r * 2 * Cp + c];
    s2 += sums[(size_t)r * 2 * Cp + Cp + c];
  }
  const double n = (double)rows;
  const double mean = s1 / n;
  double m2 = s2 - s1 * mean;
  if (m2 < 0) m2 = 0;
  bn_write_stats(c, n, mean, m2, eps, momentum, mean_o, rstd_o, rmean, rvar);
}

// ---- forward apply --------------------------------------------------------------------------------------------
// The three streaming kernels below are templated on the activation (a runtime switch inside the unrolled channel
// loop is replicated, branches included, per element) and walk ROWS_U rows per thread and iteration with all loads
// issued before the first use: they are pure HBM streams and need the loads in flight, not the arithmetic.
constexpr int ROWS_U = 4;
template <int ACT> __device__ __forceinline__ float act_c(float x, float slope) {
  if constexpr (ACT == VFD_ACT_LRELU) return x > 0.f ? x : x * slope;
  else if constexpr (ACT == VFD_ACT_SIGMOID) return 1.f / (1.f + __expf(-x));
  else if constexpr (ACT == VFD_ACT_TANH) return tanhf(x);
  else return x;
}
template <int ACT> __device__ __forceinline__ float act_grad_in_c(float x, float slope) {
  if constexpr (ACT == VFD_ACT_LRELU) return x > 0.f ? 1.f : slope;
  else if constexpr (ACT == VFD_ACT_SIGMOID) { const float y = 1.f / (1.f + __expf(-x)); return y * (1.f - y); }
  else if constexpr (ACT == VFD_ACT_TANH) { const float y = tanhf(x); return 1.f - y * y; }
  else return 1.f;
}

// Sums of 8 consecutive channels (granule g) over the replica rows a conv epilogue wrote ([VFD_STATS_REPLICAS][2][Cp],
// common.hpp).  The TY row-lanes of a granule share the rows (lane ty loads rows ty, ty + TY, ...: 4 16-byte loads each —
// one thread loading all 8 rows either keeps 128 registers of loads in flight or pays 8 dependent L2 round trips), LDS
// brings them together and every lane adds the 8 partials in row order, in double: the same bits in every workgroup.
// EVERY thread of the workgroup must call it (one barrier).
__device__ __forceinline__ void fold_replicas8(const float* __restrict__ sums, int Cp, int g, bool live, int tx, int ty, int TX, int TY,
                                               float (*sh)[16 + 1], double (&s1)[8], double (&s2)[8]) {
  static_assert(VFD_STATS_REPLICAS == 8, "row-lane split below");
  float pa[16];
#pragma unroll
  for (int k = 0; k < 16; ++k) pa[k] = 0.f;
  if (live) {
    for (int r = ty; r < VFD_STATS_REPLICAS; r += TY) {
      const float4* a = reinterpret_cast<const float4*>(sums + (size_t)r * 2 * Cp + g * 8);
      const float4* b = reinterpret_cast<const float4*>(sums + (size_t)r * 2 * Cp + Cp + g * 8);
      const float4 a0 = a[0], a1 = a[1], b0 = b[0], b1 = b[1];
      if (TY >= VFD_STATS_REPLICAS) {      // one row per lane: the partial IS the row
        pa[0] = a0.x; pa[1] = a0.y; pa[2] = a0.z; pa[3] = a0.w; pa[4] = a1.x; pa[5] = a1.y; pa[6] = a1.z; pa[7] = a1.w;
        pa[8] = b0.x; pa[9] = b0.y; pa[10] = b0.z; pa[11] = b0.w; pa[12] = b1.x; pa[13] = b1.y; pa[14] = b1.z; pa[15] = b1.w;
      } else {
        pa[0] += a0.x; pa[1] += a0.y; pa[2] += a0.z; pa[3] += a0.w; pa[4] += a1.x; pa[5] += a1.y; pa[6] += a1.z; pa[7] += a1.w;
        pa[8] += b0.x; pa[9] += b0.y; pa[10] += b0.z; pa[11] += b0.w; pa[12] += b1.x; pa[13] += b1.y; pa[14] += b1.z; pa[15] += b1.w;
      }
    }
  }
#pragma unroll
  for (int k = 0; k < 16; ++k) sh[threadIdx.x][k] = pa[k];
  __syncthreads();
#pragma unroll
  for (int k = 0; k < 8; ++k) { s1[k] = 0; s2[k] = 0; }
  const int nl = TY < VFD_STATS_REPLICAS ? TY : VFD_STATS_REPLICAS;
  for (int j = 0; j < nl; ++j) {
#pragma unroll
    for (int k = 0; k < 8; ++k) { s1[k] += (double)sh[j * TX + tx][k]; s2[k] += (double)sh[j * TX + tx][8 + k]; }
  }
}

// The same for the FORWARD statistics, which the conv epilogues sum in double (conv_epilogue.hpp EpiP::stats: the variance is
// formed as E[x^2] - mean^2 below, and float32 sums lose (|mean| / sigma)^2 digits there); sum row and square row go through
// the LDS one after the other (18 KB instead of 35).  EVERY thread of the workgroup must call it (three barriers).
__device__ __forceinline__ void fold_replicas8_d(const double* __restrict__ sums, int Cp, int g, bool live, int tx, int ty, int TX, int TY,
                                                 double (*sh)[8 + 1], double (&s1)[8], double (&s2)[8]) {
  static_assert(VFD_STATS_REPLICAS == 8, "row-lane split below");
  const int nl = TY < VFD_STATS_REPLICAS ? TY : VFD_STATS_REPLICAS;
#pragma unroll
  for (int half = 0; half < 2; ++half) {
    double pa[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) pa[k] = 0.0;
    if (live) {
      for (int r = ty; r < VFD_STATS_REPLICAS; r += TY) {
        const double2* a = reinterpret_cast<const double2*>(sums + (size_t)r * 2 * Cp + (size_t)half * Cp + g * 8);
        const double2 a0 = a[0], a1 = a[1], a2 = a[2], a3 = a[3];
        pa[0] += a0.x; pa[1] += a0.y; pa[2] += a1.x; pa[3] += a1.y; pa[4] += a2.x; pa[5] += a2.y; pa[6] += a3.x; pa[7] += a3.y;
      }
    }
    if (half) __syncthreads();      // every lane is done with the first half's partials
#pragma unroll
    for (int k = 0; k < 8; ++k) sh[threadIdx.x][k] = pa[k];
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      double v = 0.0;
      for (int j = 0; j < nl; ++j) v += sh[j * TX + tx][k];
      if (half) s2[k] = v; else s1[k] = v;
    }
  }
}

// batch statistics handed over as epilogue sums (SUMS): the fold that bn_from_sums_kernel does in a launch of its own is
// repeated by every thread for its 8 channels (32 L2-resident 16-byte loads); the row-0 workgroups publish mean / rstd
// (saved for backward) and update the running statistics.
struct BnSumsArg {
  const double* sums;      // [VFD_STATS_REPLICAS][2][Cp] doubles (conv epilogue statistics)
  float eps, momentum;
  float* mean_o;
  float* rstd_o;
  float* rmean;
  float* rvar;
  long long* nbt;
};

// POOL variants (BatchNorm -> activation -> AvgPool3d with kernel = stride in {1,2}^3 and no other consumer of the
// activation: anogan's NetD, models/anogan.py:84-105; mygan's SDisc / TDisc, models/mygannet.py:132-133,174-175): the kernels walk the POOLED rows; a thread reads the 8 input rows of its pooled voxel.  The
// forward writes only the pooled tensor (the full-resolution activation is never needed again: BatchNorm's backward works
// from x), the backward passes take the POOLED gradient and form dy = gp / nq on the fly: per layer and direction the
// full-resolution activation / gradient tensor is neither written nor re-read (2 + 3 tensor passes fewer).
struct PoolArg {
  int D, H, W;      // INPUT extents, multiples of the pool extents
  int pd, ph, pw;   // pool kernel = stride, each 1 or 2 (AvgPool3d(2), (1,2,2), (2,1,1)): nq = pd*ph*pw inputs per pooled voxel
  void* full;       // forward: also write the full-resolution activation here (it has a second consumer, e.g. a U-Net skip);
                    // backward: the gradient that arrived for that full-resolution output, added to gp / nq.  Null: pool only
};
__device__ __forceinline__ long long pool_base_row(long long ro, const PoolArg& pa) {
  const int Wo = pa.W / pa.pw, Ho = pa.H / pa.ph, Do = pa.D / pa.pd;
  long long q = ro;
  const int wo = (int)(q % Wo); q /= Wo;
  const int ho = (int)(q % Ho); q /= Ho;
  const int dq = (int)(q % Do); q /= Do;      // q = n
  return ((q * pa.D + pa.pd * dq) * pa.H + pa.ph * ho) * pa.W + pa.pw * wo;
}
// row offset of input q (< nq) of a pooled voxel: q enumerates (i < pd, j < ph, k < pw), k fastest
__device__ __forceinline__ long long pool_off(int q, const PoolArg& pa) {
  const int k = q % pa.pw, j = (q / pa.pw) % pa.ph, i = q / (pa.pw * pa.ph);
  return ((long long)i * pa.H + j) * pa.W + k;
}

template <typename T, int ACT, bool SUMS, bool POOL = false>
__global__ __launch_bounds__(256) void bn_act_fwd_kernel(const T* __restrict__ x, T* __restrict__ y, long long rows, int C, int TX,
                                                         long long rpb, const float* __restrict__ mean,
                                                         const float* __restrict__ rstd, const float* __restrict__ gamma,
                                                         const float* __restrict__ beta, float slope, BnSumsArg sa, PoolArg pa = PoolArg()) {
  const int Cp = (C + 7) & ~7, GR = Cp >> 3;
  const int tx = threadIdx.x % TX, ty = threadIdx.x / TX, TY = 256 / TX;
  const int g = blockIdx.x * TX + tx;
  __shared__ double sh_fold[SUMS ? 256 : 1][8 + 1];
  double s1[8], s2[8];
  if constexpr (SUMS) {
    if (sa.nbt != nullptr && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) *sa.nbt += 1;      // BatchNorm.num_batches_tracked
    fold_replicas8_d(sa.sums, Cp, g, g < GR, tx, ty, TX, TY, sh_fold, s1, s2);
  }
  if (g >= GR) return;
  float sc[8], sf[8];
  if constexpr (SUMS) {
    const double n = (double)rows * (POOL ? (double)(pa.pd * pa.ph * pa.pw) : 1.0), inv_n = 1.0 / n;      // statistics are over the INPUT rows
    const bool publish = blockIdx.y == 0 && ty == 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int c = g * 8 + k;
      const double mu = s1[k] * inv_n;
      double m2 = s2[k] - s1[k] * mu;
      if (m2 < 0) m2 = 0;
      const float var = (float)(m2 * inv_n);
      const float rs = 1.f / sqrtf(var + sa.eps);
      if (c < C) {
        sc[k] = rs * (gamma ? gamma[c] : 1.f);
        sf[k] = (beta ? beta[c] : 0.f) - (float)mu * sc[k];
        if (publish) {
          sa.mean_o[c] = (float)mu;
          sa.rstd_o[c] = rs;
          if (sa.rmean != nullptr) sa.rmean[c] = (1.f - sa.momentum) * sa.rmean[c] + sa.momentum * (float)mu;
          if (sa.rvar != nullptr) sa.rvar[c] = (1.f - sa.momentum) * sa.rvar[c] + sa.momentum * (float)(n > 1 ? m2 / (n - 1) : m2 * inv_n);
        }
      } else { sc[k] = 0.f; sf[k] = 0.f; }
    }
  } else {
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int c = g * 8 + k;
      if (c < C) {
        sc[k] = rstd[c] * (gamma ? gamma[c] : 1.f);
        sf[k] = (beta ? beta[c] : 0.f) - mean[c] * sc[k];
      } else { sc[k] = 0.f; sf[k] = 0.f; }
    }
  }
  const long long rbeg = (long long)blockIdx.y * rpb;
  long long rend = rbeg + rpb;
  if (rend > rows) rend = rows;
  if constexpr (POOL) {
    for (long long r = rbeg + ty; r < rend; r += TY) {
      const long long b0 = pool_base_row(r, pa);
      const int nq = pa.pd * pa.ph * pa.pw;
      const float inv_q = 1.f / (float)nq;
      float v[8][8], o[8];
#pragma unroll
      for (int q = 0; q < 8; ++q)
        if (q < nq) load8(x + (b0 + pool_off(q, pa)) * Cp + g * 8, v[q]);
#pragma unroll
      for (int k = 0; k < 8; ++k) o[k] = 0.f;
      T* yfull = reinterpret_cast<T*>(pa.full);
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        if (q < nq) {
#pragma unroll
          for (int k = 0; k < 8; ++k) {
            v[q][k] = (g * 8 + k < C) ? act_c<ACT>(v[q][k] * sc[k] + sf[k], slope) : 0.f;
            o[k] += v[q][k];
          }
          if (yfull != nullptr) store8(yfull + (b0 + pool_off(q, pa)) * Cp + g * 8, v[q]);
        }
      }
#pragma unroll
      for (int k = 0; k < 8; ++k) o[k] *= inv_q;
      store8(y + r * Cp + g * 8, o);
    }
    return;
  }
  for (long long r = rbeg + ty; r < rend; r += (long long)ROWS_U * TY) {
    float v[ROWS_U][8];
#pragma unroll
    for (int u = 0; u < ROWS_U; ++u) {
      const long long ru = r + (long long)u * TY;
      load8(x + (ru < rend ? ru : r) * Cp + g * 8, v[u]);
    }
#pragma unroll
    for (int u = 0; u < ROWS_U; ++u) {
      const long long ru = r + (long long)u * TY;
#pragma unroll
      for (int k = 0; k < 8; ++k) v[u][k] = (g * 8 + k < C) ? act_c<ACT>(v[u][k] * sc[k] + sf[k], slope) : 0.f;
      if (ru < rend) store8(y + ru * Cp + g * 8, v[u]);
    }
  }
}

// ---- backward reduce: partial sums of g = dy*act'(z) and g*xhat per channel ---------------------------------------
// SUMS: the workgroup's partial sums are added into a replica row of a zeroed [VFD_STATS_REPLICAS][2][Cp] buffer (float
// atomics, one per channel and workgroup) instead of a slot of the partials workspace: no finalize launch, the apply
// kernel folds the replicas itself.
template <typename T, int ACT, bool SUMS, bool POOL = false>
__global__ __launch_bounds__(256) void bn_act_bwd_partial_kernel(const T* __restrict__ x, const T* __restrict__ dy,
                                                                 float* __restrict__ part, long long rows, int C, int TX,
                                                                 long long rpb, const float* __restrict__ mean,
                                                                 const float* __restrict__ rstd, const float* __restrict__ gamma,
                                                                 const float* __restrict__ beta, float slope, PoolArg pa = PoolArg()) {
  const int Cp = (C + 7) & ~7, GR = Cp >> 3;
  const int tx = threadIdx.x % TX, ty = threadIdx.x / TX, TY = 256 / TX;
  const int g = blockIdx.x * TX + tx;
  float sg[8], sgx[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) { sg[k] = 0.f; sgx[k] = 0.f; }
  if (g < GR) {
    float mu[8], rs[8], ga[8], be[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int c = g * 8 + k;
      const bool ok = c < C;
      mu[k] = ok ? mean[c] : 0.f; rs[k] = ok ? rstd[c] : 0.f;
      ga[k] = ok ? (gamma ? gamma[c] : 1.f) : 0.f; be[k] = ok ? (beta ? beta[c] : 0.f) : 0.f;
    }
    const long long rbeg = (long long)blockIdx.y * rpb;
    long long rend = rbeg + rpb;
    if (rend > rows) rend = rows;
    if constexpr (POOL) {
      for (long long r = rbeg + ty; r < rend; r += TY) {      // dy = the pooled gradient / 8 at each of the 8 input rows
        const long long b0 = pool_base_row(r, pa);
        const int nq = pa.pd * pa.ph * pa.pw;
        const float inv_q = 1.f / (float)nq;
        float v[8][8], gp[8];
        load8(dy + r * Cp + g * 8, gp);
        const T* gfull = reinterpret_cast<const T*>(pa.full);
#pragma unroll
        for (int q = 0; q < 8; ++q)
          if (q < nq) load8(x + (b0 + pool_off(q, pa)) * Cp + g * 8, v[q]);
#pragma unroll
        for (int q = 0; q < 8; ++q) {
          if (q < nq) {
            float gf[8];
            if (gfull != nullptr) load8(gfull + (b0 + pool_off(q, pa)) * Cp + g * 8, gf);
#pragma unroll
            for (int k = 0; k < 8; ++k) {
              const float xh = (v[q][k] - mu[k]) * rs[k];
              const float dyv = gp[k] * inv_q + (gfull != nullptr ? gf[k] : 0.f);
              const float gz = dyv * act_grad_in_c<ACT>(xh * ga[k] + be[k], slope);
              sg[k] += gz; sgx[k] += gz * xh;
            }
          }
        }
      }
    } else
    for (long long r = rbeg + ty; r < rend; r += (long long)ROWS_U * TY) {
      float v[ROWS_U][8], d[ROWS_U][8];
#pragma unroll
      for (int u = 0; u < ROWS_U; ++u) {
        const long long ru = r + (long long)u * TY;
        const long long rc = ru < rend ? ru : r;
        load8(x + rc * Cp + g * 8, v[u]);
        load8(dy + rc * Cp + g * 8, d[u]);
      }
#pragma unroll
      for (int u = 0; u < ROWS_U; ++u) {      // rows are added in the same order as an un-unrolled walk
        if (r + (long long)u * TY < rend) {
#pragma unroll
          for (int k = 0; k < 8; ++k) {
            const float xh = (v[u][k] - mu[k]) * rs[k];
            const float gz = d[u][k] * act_grad_in_c<ACT>(xh * ga[k] + be[k], slope);
            sg[k] += gz; sgx[k] += gz * xh;
          }
        }
      }
    }
  }
  __shared__ float sh[256][16 + 1];
#pragma unroll
  for (int k = 0; k < 8; ++k) { sh[threadIdx.x][k] = sg[k]; sh[threadIdx.x][8 + k] = sgx[k]; }
  __syncthreads();
  if (ty == 0 && g < GR) {
    float a[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) a[k] = sh[tx][k];
    for (int j = 1; j < TY; ++j)
#pragma unroll
      for (int k = 0; k < 16; ++k) a[k] += sh[j * TX + tx][k];
    if constexpr (SUMS) {
      float* dst = part + (size_t)(blockIdx.y % VFD_STATS_REPLICAS) * 2 * Cp;
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        if (g * 8 + k < C) { atomicAdd(dst + g * 8 + k, a[k]); atomicAdd(dst + Cp + g * 8 + k, a[8 + k]); }
      }
    } else {
      float* dst = part + (size_t)blockIdx.y * 2 * Cp;
#pragma unroll
      for (int k = 0; k < 8; ++k) { dst[g * 8 + k] = a[k]; dst[Cp + g * 8 + k] = a[8 + k]; }
    }
  }
}

// 8 channels x 32 part-lanes per workgroup: a lane folds every 32nd partial, 5 shuffles fold the lanes (fixed order).
__global__ __launch_bounds__(256) void bn_bwd_finalize_kernel(const float* __restrict__ part, int nparts, int C, float* dgamma,
                                                              float* dbeta, float* dgamma_acc, float* dbeta_acc) {
  const int Cp = (C + 7) & ~7;
  const int pl = threadIdx.x & 31;
  const int c = blockIdx.x * 8 + (threadIdx.x >> 5);
  double sg = 0, sgx = 0;
  if (c < C)
    for (int j = pl; j < nparts; j += 32) {
      sg += part[(size_t)j * 2 * Cp + c];
      sgx += part[(size_t)j * 2 * Cp + Cp + c];
    }
#pragma unroll
  for (int o = 16; o > 0; o >>= 1) {
    sg += __shfl_xor(sg, o, 64);
    sgx += __shfl_xor(sgx, o, 64);
  }
  if (pl == 0 && c < C) {
    dbeta[c] = (float)sg;
    dgamma[c] = (float)sgx;
    if (dbeta_acc != nullptr) dbeta_acc[c] += (float)sg;       // parameter-gradient arena (one writer per channel)
    if (dgamma_acc != nullptr) dgamma_acc[c] += (float)sgx;
  }
}

template <typename T, int ACT>
__global__ __launch_bounds__(256) void bn_act_bwd_apply_kernel(const T* __restrict__ x, const T* __restrict__ dy, T* __restrict__ dx,
                                                               long long rows, int C, int TX, long long rpb,
                                                               const float* __restrict__ mean, const float* __restrict__ rstd,
                                                               const float* __restrict__ gamma, const float* __restrict__ beta,
                                                               float slope, const float* __restrict__ dgamma,
                                                               const float* __restrict__ dbeta) {
  const int Cp = (C + 7) & ~7, GR = Cp >> 3;
  const int tx = threadIdx.x % TX, ty = threadIdx.x / TX, TY = 256 / TX;
  const int g = blockIdx.x * TX + tx;
  if (g >= GR) return;
  float mu[8], rs[8], ga[8], be[8], c1[8], c2[8];
  const float inv = 1.f / (float)rows;
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const int c = g * 8 + k;
    const bool ok = c < C;
    mu[k] = ok ? mean[c] : 0.f; rs[k] = ok ? rstd[c] : 0.f;
    ga[k] = ok ? (gamma ? gamma[c] : 1.f) : 0.f; be[k] = ok ? (beta ? beta[c] : 0.f) : 0.f;
    c1[k] = ok ? dbeta[c] * inv : 0.f; c2[k] = ok ? dgamma[c] * inv : 0.f;
  }
  const long long rbeg = (long long)blockIdx.y * rpb;
  long long rend = rbeg + rpb;
  if (rend > rows) rend = rows;
  for (long long r = rbeg + ty; r < rend; r += (long long)ROWS_U * TY) {
    float v[ROWS_U][8], d[ROWS_U][8];
#pragma unroll
    for (int u = 0; u < ROWS_U; ++u) {
      const long long ru = r + (long long)u * TY;
      const long long rc = ru < rend ? ru : r;
      load8(x + rc * Cp + g * 8, v[u]);
      load8(dy + rc * Cp + g * 8, d[u]);
    }
#pragma unroll
    for (int u = 0; u < ROWS_U; ++u) {
      const long long ru = r + (long long)u * TY;
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const float xh = (v[u][k] - mu[k]) * rs[k];
        const float gz = d[u][k] * act_grad_in_c<ACT>(xh * ga[k] + be[k], slope);
        d[u][k] = ga[k] * rs[k] * (gz - c1[k] - xh * c2[k]);
      }
      if (ru < rend) store8(dx + ru * Cp + g * 8, d[u]);
    }
  }
}

// ---- backward apply for a BatchNorm whose gradient arrives as g = dy * act'(z) with the sums of g and g * xhat in replica
// rows (written by the consumer convolution's data-gradient epilogue, conv_epilogue.hpp): the reduce pass and the finalize
// launch are gone; the row-0 workgroups publish dgamma / dbeta.
// ACT < 0: gz IS g (conv hand-over); ACT >= 0: gz is dy and g = dy * act'(gamma * xhat + beta) is formed here, as in
// bn_act_bwd_apply_kernel (the sums then come from bn_act_bwd_partial_kernel<.., SUMS>).
template <typename T, int ACT, bool POOL = false>
__global__ __launch_bounds__(256) void bn_bwd_apply_sums_kernel(const T* __restrict__ x, const T* __restrict__ gz, T* __restrict__ dx,
                                                                long long rows, int C, int TX, long long rpb,
                                                                const float* __restrict__ mean, const float* __restrict__ rstd,
                                                                const float* __restrict__ gamma, const float* __restrict__ beta,
                                                                float slope, const float* __restrict__ sums,
                                                                float* dgamma, float* dbeta, float* dgamma_acc, float* dbeta_acc,
                                                                float* colsum_acc, PoolArg pa = PoolArg()) {
  const int Cp = (C + 7) & ~7, GR = Cp >> 3;
  const int tx = threadIdx.x % TX, ty = threadIdx.x / TX, TY = 256 / TX;
  const int g = blockIdx.x * TX + tx;
  __shared__ float sh_cs[256][16 + 1];     // replica fold, then the column-sum fold
  double s1[8], s2[8];
  fold_replicas8(sums, Cp, g, g < GR, tx, ty, TX, TY, sh_cs, s1, s2);
  if (g >= GR) {
    if (colsum_acc != nullptr) { __syncthreads(); __syncthreads(); }      // the two barriers of the column-sum fold at the end
    return;
  }
  // dx = gamma*rstd*(g - mean(g) - xh*mean(g*xh)) with xh = (x - mu)*rstd, re-associated into dx = g*A + xc*B + D with
  // xc = x - mu, and z = gamma*xh + beta = xc*A + beta for the activation derivative: 5 constants per channel instead of
  // 7 (this kernel streams 4 rows x 2 tensors per thread; at 7 it needed 188 registers, i.e. 2 waves per SIMD).  The mean
  // is subtracted FIRST: round 2's x*B + D (D carrying mu*B) cancelled |mean|/sigma digits (8.6e-3 on dx at a ratio of 1e3,
  // tests/test_hip_primitives.py::test_batchnorm_epilogue_statistics_large_mean_over_sigma).
  float A[8], B[8], D[8], Mu[8], Be[8], cs[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) cs[k] = 0.f;
  const float inv = 1.f / ((float)rows * (POOL ? (float)(pa.pd * pa.ph * pa.pw) : 1.f));      // means are over the INPUT rows
  {
    const bool publish = blockIdx.y == 0 && ty == 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int c = g * 8 + k;
      const bool ok = c < C;
      const float sg = (float)s1[k], sgx = (float)s2[k];
      const float mu = ok ? mean[c] : 0.f, rs = ok ? rstd[c] : 0.f;
      const float ga = ok ? (gamma ? gamma[c] : 1.f) : 0.f, be = (ok && beta) ? beta[c] : 0.f;
      const float gr = ga * rs, c1 = ok ? sg * inv : 0.f, c2 = ok ? sgx * inv : 0.f;
      A[k] = gr;                 // = gamma * rstd: also the slope of z in xc
      B[k] = -rs * gr * c2;
      D[k] = -gr * c1;
      Mu[k] = mu;
      Be[k] = be;
      if (ok && publish) {
        dbeta[c] = sg;
        dgamma[c] = sgx;
        if (dbeta_acc != nullptr) dbeta_acc[c] += sg;       // parameter-gradient arena (one writer per channel)
        if (dgamma_acc != nullptr) dgamma_acc[c] += sgx;
      }
    }
  }
  const long long rbeg = (long long)blockIdx.y * rpb;
  long long rend = rbeg + rpb;
  if (rend > rows) rend = rows;
  if constexpr (POOL) {
    for (long long r = rbeg + ty; r < rend; r += TY) {
      const long long b0 = pool_base_row(r, pa);
      const int nq = pa.pd * pa.ph * pa.pw;
      const float inv_q = 1.f / (float)nq;
      float v[8][8], gp[8];
      load8(gz + r * Cp + g * 8, gp);
      const T* gfull = reinterpret_cast<const T*>(pa.full);
#pragma unroll
      for (int q = 0; q < 8; ++q)
        if (q < nq) load8(x + (b0 + pool_off(q, pa)) * Cp + g * 8, v[q]);
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        if (q < nq) {
          float o[8], gf[8];
          if (gfull != nullptr) load8(gfull + (b0 + pool_off(q, pa)) * Cp + g * 8, gf);
#pragma unroll
          for (int k = 0; k < 8; ++k) {
            float gzv = gp[k] * inv_q + (gfull != nullptr ? gf[k] : 0.f);
            const float xc = v[q][k] - Mu[k];
            if constexpr (ACT >= 0) gzv *= act_grad_in_c<ACT>(xc * A[k] + Be[k], slope);
            o[k] = gzv * A[k] + (xc * B[k] + D[k]);
            cs[k] += o[k];
          }
          store8(dx + (b0 + pool_off(q, pa)) * Cp + g * 8, o);
        }
      }
    }
  } else
  for (long long r = rbeg + ty; r < rend; r += (long long)ROWS_U * TY) {
    float v[ROWS_U][8], d[ROWS_U][8];
#pragma unroll
    for (int u = 0; u < ROWS_U; ++u) {
      const long long ru = r + (long long)u * TY;
      const long long rc = ru < rend ? ru : r;
      load8(x + rc * Cp + g * 8, v[u]);
      load8(gz + rc * Cp + g * 8, d[u]);
    }
#pragma unroll
    for (int u = 0; u < ROWS_U; ++u) {
      const long long ru = r + (long long)u * TY;
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        float gzv = d[u][k];
        v[u][k] -= Mu[k];
        if constexpr (ACT >= 0) gzv *= act_grad_in_c<ACT>(v[u][k] * A[k] + Be[k], slope);
        d[u][k] = gzv * A[k] + (v[u][k] * B[k] + D[k]);
      }
      if (ru < rend) {
        store8(dx + ru * Cp + g * 8, d[u]);
#pragma unroll
        for (int k = 0; k < 8; ++k) cs[k] += d[u][k];
      }
    }
  }
  // bias gradient of the convolution that feeds this BatchNorm: the column sums of dx, folded over the row-lanes and
  // added into a replica row of colsum_acc[VFD_STATS_REPLICAS][Cp] (one float atomic per channel and workgroup; straight
  // into the gradient, 512 workgroups per address, the atomics alone took 80 us) — the separate column-sum pass over dx
  // (vfd_bias_grad: one more read of the tensor and two launches) is gone; vfd_wgrad_reduce_bias folds the replicas
  if (colsum_acc != nullptr) {
    __syncthreads();      // every lane is done reading the replica fold from sh_cs
#pragma unroll
    for (int k = 0; k < 8; ++k) sh_cs[threadIdx.x][k] = cs[k];
    __syncthreads();
    if (ty == 0) {
      for (int j = 1; j < TY; ++j)
#pragma unroll
        for (int k = 0; k < 8; ++k) cs[k] += sh_cs[j * TX + tx][k];
#pragma unroll
      for (int k = 0; k < 8; ++k)
        if (g * 8 + k < C) atomicAdd(colsum_acc + (size_t)(blockIdx.y % VFD_STATS_REPLICAS) * Cp + g * 8 + k, cs[k]);
    }
  }
}

}  // namespace

// launch M(T, ACT) for the runtime (dtype, act) pair
#define BN_ACT_DISPATCH(M)                                                                  \
  do {                                                                                      \
    if (dtype == VFD_BF16) {                                                                \
      switch (act) {                                                                        \
        case VFD_ACT_LRELU: M(bf16_t, VFD_ACT_LRELU); break;                                \
        case VFD_ACT_SIGMOID: M(bf16_t, VFD_ACT_SIGMOID); break;                            \
        case VFD_ACT_TANH: M(bf16_t, VFD_ACT_TANH); break;                                  \
        default: M(bf16_t, VFD_ACT_NONE); break;                                            \
      }                                                                                     \
    } else {                                                                                \
      switch (act) {                                                                        \
        case VFD_ACT_LRELU: M(float, VFD_ACT_LRELU); break;                                 \
        case VFD_ACT_SIGMOID: M(float, VFD_ACT_SIGMOID); break;                             \
        case VFD_ACT_TANH: M(float, VFD_ACT_TANH); break;                                   \
        default: M(float, VFD_ACT_NONE); break;                                             \
      }                                                                                     \
    }                                                                                       \
  } while (0)

// One more momentum update of the running statistics with batch statistics that are already known (mean, rstd of a
// forward that would otherwise be repeated on identical input and weights): var = 1/rstd^2 - eps.
__global__ void bn_running_update_kernel(const float* __restrict__ mean, const float* __restrict__ rstd, long long rows, int C,
                                         float eps, float momentum, float* rmean, float* rvar, long long* nbt) {
  if (nbt != nullptr && blockIdx.x == 0 && threadIdx.x == 0) *nbt += 1;
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  const double rs = rstd[c];
  double var = 1.0 / (rs * rs) - (double)eps;
  if (var < 0) var = 0;
  const double n = (double)rows;
  if (rmean != nullptr) rmean[c] = (1.f - momentum) * rmean[c] + momentum * mean[c];
  if (rvar != nullptr) rvar[c] = (1.f - momentum) * rvar[c] + momentum * (float)(n > 1 ? var * n / (n - 1) : var);
}

extern "C" int vfd_bn_running_update(const float* mean, const float* rstd, int64_t rows, int C, float eps, float momentum,
                                     float* running_mean, float* running_var, int64_t* num_batches_tracked, void* stream) {
  VFD_REQUIRE(mean && rstd && rows > 0 && C > 0, "bn_running_update: bad arguments");
  hipLaunchKernelGGL(bn_running_update_kernel, dim3((C + 127) / 128), dim3(128), 0, as_stream(stream), mean, rstd, (long long)rows, C, eps,
                     momentum, running_mean, running_var, reinterpret_cast<long long*>(num_batches_tracked));
  VFD_CHECK_LAUNCH("bn_running_update");
  return VFD_OK;
}

extern "C" size_t vfd_bn_workspace(int64_t rows, int C) {
  (void)rows;
  return (size_t)BN_MAX_BLOCKS * 3 * cpad(C) * sizeof(float);
}

extern "C" int vfd_bn_stats(int dtype, const void* x, int64_t rows, int C, float eps, float momentum, float* mean, float* rstd,
                            float* running_mean, float* running_var, int64_t* num_batches_tracked, void* ws, void* stream) {
  VFD_REQUIRE(dtype == VFD_F32 || dtype == VFD_BF16, "bn_stats: bad dtype");
  VFD_REQUIRE(x && mean && rstd && ws && rows > 0 && C > 0, "bn_stats: bad arguments");
  const Tiling t = make_tiling(rows, C, BN_MAX_BLOCKS);
  dim3 grid(t.gx, t.gy);
  float* part = reinterpret_cast<float*>(ws);
  if (dtype == VFD_BF16)
    hipLaunchKernelGGL(bn_partial_kernel<bf16_t>, grid, dim3(256), 0, as_stream(stream), (const bf16_t*)x, part, (long long)rows, C, t.TX, t.rows_per_block);
  else
    hipLaunchKernelGGL(bn_partial_kernel<float>, grid, dim3(256), 0, as_stream(stream), (const float*)x, part, (long long)rows, C, t.TX, t.rows_per_block);
  VFD_CHECK_LAUNCH("bn_partial");
  hipLaunchKernelGGL(bn_finalize_kernel, dim3((C + 31) / 32), dim3(256), 0, as_stream(stream), part, t.gy, C, eps, momentum, mean,
                     rstd, running_mean, running_var, reinterpret_cast<long long*>(num_batches_tracked));
  VFD_CHECK_LAUNCH("bn_finalize");
  return VFD_OK;
}

extern "C" int vfd_bn_stats_from_sums(const double* stats, int64_t rows, int C, float eps, float momentum, float* mean, float* rstd,
                                      float* running_mean, float* running_var, int64_t* num_batches_tracked, void* stream) {
  VFD_REQUIRE(stats && mean && rstd && rows > 0 && C > 0, "bn_stats_from_sums: bad arguments");
  hipLaunchKernelGGL(bn_from_sums_kernel, dim3((C + 127) / 128), dim3(128), 0, as_stream(stream), stats, (long long)rows, C, eps,
                     momentum, mean, rstd, running_mean, running_var, reinterpret_cast<long long*>(num_batches_tracked));
  VFD_CHECK_LAUNCH("bn_from_sums");
  return VFD_OK;
}

extern "C" int vfd_bn_act_forward(int dtype, const void* x, void* y, int64_t rows, int C, const float* mean, const float* rstd,
                                  const float* gamma, const float* beta, int act, float slope, void* stream) {
  VFD_REQUIRE(dtype == VFD_F32 || dtype == VFD_BF16, "bn_act_forward: bad dtype");
  VFD_REQUIRE(x && y && mean && rstd && rows > 0 && C > 0, "bn_act_forward: bad arguments");
  const Tiling t = make_stream_tiling(rows, C);
  dim3 grid(t.gx, t.gy);
  BnSumsArg sa = {};
#define BN_FWD(T_, ACT_) hipLaunchKernelGGL((bn_act_fwd_kernel<T_, ACT_, false>), grid, dim3(256), 0, as_stream(stream), (const T_*)x, (T_*)y, (long long)rows, C, t.TX, t.rows_per_block, mean, rstd, gamma, beta, slope, sa)
  BN_ACT_DISPATCH(BN_FWD);
#undef BN_FWD
  VFD_CHECK_LAUNCH("bn_act_forward");
  return VFD_OK;
}

extern "C" int vfd_bn_act_forward_sums(int dtype, const void* x, void* y, int64_t rows, int C, const double* sums, float eps,
                                       float momentum, float* mean, float* rstd, float* running_mean, float* running_var,
                                       int64_t* num_batches_tracked, const float* gamma, const float* beta, int act, float slope,
                                       void* stream) {
  VFD_REQUIRE(dtype == VFD_F32 || dtype == VFD_BF16, "bn_act_forward_sums: bad dtype");
  VFD_REQUIRE(x && y && sums && mean && rstd && rows > 0 && C > 0, "bn_act_forward_sums: bad arguments");
  VFD_REQUIRE(((uintptr_t)sums & 15) == 0, "bn_act_forward_sums: the sums buffer must be 16-byte aligned");
  const Tiling t = make_stream_tiling(rows, C);
  dim3 grid(t.gx, t.gy);
  BnSumsArg sa;
  sa.sums = sums; sa.eps = eps; sa.momentum = momentum; sa.mean_o = mean; sa.rstd_o = rstd;
  sa.rmean = running_mean; sa.rvar = running_var; sa.nbt = reinterpret_cast<long long*>(num_batches_tracked);
#define BN_FWD(T_, ACT_) hipLaunchKernelGGL((bn_act_fwd_kernel<T_, ACT_, true>), grid, dim3(256), 0, as_stream(stream), (const T_*)x, (T_*)y, (long long)rows, C, t.TX, t.rows_per_block, nullptr, nullptr, gamma, beta, slope, sa)
  BN_ACT_DISPATCH(BN_FWD);
#undef BN_FWD
  VFD_CHECK_LAUNCH("bn_act_forward_sums");
  return VFD_OK;
}

extern "C" int vfd_bn_act_pool_forward_sums(int dtype, const void* x, void* y, int N, int D, int H, int W, int pd, int ph, int pw, int C, const double* sums,
                                            float eps, float momentum, float* mean, float* rstd, float* running_mean,
                                            float* running_var, int64_t* num_batches_tracked, const float* gamma, const float* beta,
                                            int act, float slope, void* y_full, void* stream) {
  VFD_REQUIRE(dtype == VFD_F32 || dtype == VFD_BF16, "bn_act_pool_forward_sums: bad dtype");
  VFD_REQUIRE(x && y && sums && mean && rstd && N > 0 && C > 0, "bn_act_pool_forward_sums: bad arguments");
  VFD_REQUIRE((pd == 1 || pd == 2) && (ph == 1 || ph == 2) && (pw == 1 || pw == 2), "bn_act_pool_forward_sums: pool extents must be 1 or 2");
  VFD_REQUIRE(D > 0 && H > 0 && W > 0 && D % pd == 0 && H % ph == 0 && W % pw == 0, "bn_act_pool_forward_sums: %dx%dx%d is not a multiple of the pool %dx%dx%d", D, H, W, pd, ph, pw);
  const long long orows = (long long)N * (D / pd) * (H / ph) * (W / pw);
  const Tiling t = make_stream_tiling(orows, C);
  dim3 grid(t.gx, t.gy);
  BnSumsArg sa;
  sa.sums = sums; sa.eps = eps; sa.momentum = momentum; sa.mean_o = mean; sa.rstd_o = rstd;
  sa.rmean = running_mean; sa.rvar = running_var; sa.nbt = reinterpret_cast<long long*>(num_batches_tracked);
  PoolArg pa; pa.D = D; pa.H = H; pa.W = W; pa.pd = pd; pa.ph = ph; pa.pw = pw; pa.full = y_full;
#define BN_FWDP(T_, ACT_) hipLaunchKernelGGL((bn_act_fwd_kernel<T_, ACT_, true, true>), grid, dim3(256), 0, as_stream(stream), (const T_*)x, (T_*)y, orows, C, t.TX, t.rows_per_block, nullptr, nullptr, gamma, beta, slope, sa, pa)
  BN_ACT_DISPATCH(BN_FWDP);
#undef BN_FWDP
  VFD_CHECK_LAUNCH("bn_act_pool_forward_sums");
  return VFD_OK;
}

extern "C" int vfd_bn_act_pool_backward_sums(int dtype, const void* x, const void* gpool, void* dx, int N, int D, int H, int W, int pd, int ph, int pw, int C,
                                             const float* mean, const float* rstd, const float* gamma, const float* beta, int act,
                                             float slope, float* sums, float* dgamma, float* dbeta, float* dgamma_acc,
                                             float* dbeta_acc, float* colsum_acc, const void* g_full, void* stream) {
  VFD_REQUIRE(dtype == VFD_F32 || dtype == VFD_BF16, "bn_act_pool_backward_sums: bad dtype");
  VFD_REQUIRE(x && gpool && dx && mean && rstd && sums && dgamma && dbeta && N > 0 && C > 0, "bn_act_pool_backward_sums: bad arguments");
  VFD_REQUIRE((pd == 1 || pd == 2) && (ph == 1 || ph == 2) && (pw == 1 || pw == 2), "bn_act_pool_backward_sums: pool extents must be 1 or 2");
  VFD_REQUIRE(D > 0 && H > 0 && W > 0 && D % pd == 0 && H % ph == 0 && W % pw == 0, "bn_act_pool_backward_sums: extents are not a multiple of the pool");
  const long long orows = (long long)N * (D / pd) * (H / ph) * (W / pw);
  static const int pnb = getenv("VFD_BN_PART_BLOCKS") ? atoi(getenv("VFD_BN_PART_BLOCKS")) : 512;
  const Tiling t = make_tiling(orows, C, pnb < BN_MAX_BLOCKS ? pnb : BN_MAX_BLOCKS, 1);
  dim3 grid(t.gx, t.gy);
  hipStream_t st = as_stream(stream);
  PoolArg pa; pa.D = D; pa.H = H; pa.W = W; pa.pd = pd; pa.ph = ph; pa.pw = pw; pa.full = const_cast<void*>(g_full);
#define BN_BWD_PP(T_, ACT_) hipLaunchKernelGGL((bn_act_bwd_partial_kernel<T_, ACT_, true, true>), grid, dim3(256), 0, st, (const T_*)x, (const T_*)gpool, sums, orows, C, t.TX, t.rows_per_block, mean, rstd, gamma, beta, slope, pa)
  BN_ACT_DISPATCH(BN_BWD_PP);
#undef BN_BWD_PP
  VFD_CHECK_LAUNCH("bn_act_pool_bwd_partial");
  const Tiling ta = make_tiling(orows, C, 512, 1);
  dim3 grid2(ta.gx, ta.gy);
#define BN_BWD_AP(T_, ACT_) hipLaunchKernelGGL((bn_bwd_apply_sums_kernel<T_, ACT_, true>), grid2, dim3(256), 0, st, (const T_*)x, (const T_*)gpool, (T_*)dx, orows, C, ta.TX, ta.rows_per_block, mean, rstd, gamma, beta, slope, sums, dgamma, dbeta, dgamma_acc, dbeta_acc, colsum_acc, pa)
  BN_ACT_DISPATCH(BN_BWD_AP);
#undef BN_BWD_AP
  VFD_CHECK_LAUNCH("bn_act_pool_bwd_apply");
  return VFD_OK;
}

extern "C" int vfd_bn_backward_apply_sums(int dtype, const void* x, const void* g, void* dx, int64_t rows, int C, const float* mean,
                                          const float* rstd, const float* gamma, const float* sums, float* dgamma, float* dbeta,
                                          float* dgamma_acc, float* dbeta_acc, float* colsum_acc, void* stream) {
  VFD_REQUIRE(dtype == VFD_F32 || dtype == VFD_BF16, "bn_backward_apply_sums: bad dtype");
  VFD_REQUIRE(x && g && dx && mean && rstd && sums && dgamma && dbeta && rows > 0 && C > 0, "bn_backward_apply_sums: bad arguments");
  VFD_REQUIRE(((uintptr_t)sums & 15) == 0, "bn_backward_apply_sums: the sums buffer must be 16-byte aligned");
  const Tiling t = make_stream_tiling(rows, C);
  dim3 grid(t.gx, t.gy);
  if (dtype == VFD_BF16)
    hipLaunchKernelGGL((bn_bwd_apply_sums_kernel<bf16_t, -1>), grid, dim3(256), 0, as_stream(stream), (const bf16_t*)x, (const bf16_t*)g, (bf16_t*)dx, (long long)rows, C, t.TX, t.rows_per_block, mean, rstd, gamma, nullptr, 0.f, sums, dgamma, dbeta, dgamma_acc, dbeta_acc, colsum_acc);
  else
    hipLaunchKernelGGL((bn_bwd_apply_sums_kernel<float, -1>), grid, dim3(256), 0, as_stream(stream), (const float*)x, (const float*)g, (float*)dx, (long long)rows, C, t.TX, t.rows_per_block, mean, rstd, gamma, nullptr, 0.f, sums, dgamma, dbeta, dgamma_acc, dbeta_acc, colsum_acc);
  VFD_CHECK_LAUNCH("bn_backward_apply_sums");
  return VFD_OK;
}

extern "C" int vfd_bn_act_backward_sums(int dtype, const void* x, const void* dy, void* dx, int64_t rows, int C, const float* mean,
                                        const float* rstd, const float* gamma, const float* beta, int act, float slope,
                                        float* sums, float* dgamma, float* dbeta, float* dgamma_acc, float* dbeta_acc,
                                        float* colsum_acc, void* stream) {
  VFD_REQUIRE(dtype == VFD_F32 || dtype == VFD_BF16, "bn_act_backward_sums: bad dtype");
  VFD_REQUIRE(x && dy && dx && mean && rstd && sums && dgamma && dbeta && rows > 0 && C > 0, "bn_act_backward_sums: bad arguments");
  VFD_REQUIRE(((uintptr_t)sums & 15) == 0, "bn_act_backward_sums: the sums buffer must be 16-byte aligned");
  static const int pnb = getenv("VFD_BN_PART_BLOCKS") ? atoi(getenv("VFD_BN_PART_BLOCKS")) : 512;
  const Tiling t = make_tiling(rows, C, pnb < BN_MAX_BLOCKS ? pnb : BN_MAX_BLOCKS);
  dim3 grid(t.gx, t.gy);
  hipStream_t st = as_stream(stream);
#define BN_BWD_PS(T_, ACT_) hipLaunchKernelGGL((bn_act_bwd_partial_kernel<T_, ACT_, true>), grid, dim3(256), 0, st, (const T_*)x, (const T_*)dy, sums, (long long)rows, C, t.TX, t.rows_per_block, mean, rstd, gamma, beta, slope)
  BN_ACT_DISPATCH(BN_BWD_PS);
#undef BN_BWD_PS
  VFD_CHECK_LAUNCH("bn_act_bwd_partial(sums)");
  const Tiling ta = make_stream_tiling(rows, C);
  dim3 grid2(ta.gx, ta.gy);
#define BN_BWD_AS(T_, ACT_) hipLaunchKernelGGL((bn_bwd_apply_sums_kernel<T_, ACT_>), grid2, dim3(256), 0, st, (const T_*)x, (const T_*)dy, (T_*)dx, (long long)rows, C, ta.TX, ta.rows_per_block, mean, rstd, gamma, beta, slope, sums, dgamma, dbeta, dgamma_acc, dbeta_acc, colsum_acc)
  BN_ACT_DISPATCH(BN_BWD_AS);
#undef BN_BWD_AS
  VFD_CHECK_LAUNCH("bn_bwd_apply_sums");
  return VFD_OK;
}

extern "C" int vfd_bn_act_backward(int dtype, const void* x, const void* dy, void* dx, int64_t rows, int C, const float* mean,
                                   const float* rstd, const float* gamma, const float* beta, int act, float slope, float* dgamma,
                                   float* dbeta, float* dgamma_acc, float* dbeta_acc, void* ws, void* stream) {
  VFD_REQUIRE(dtype == VFD_F32 || dtype == VFD_BF16, "bn_act_backward: bad dtype");
  VFD_REQUIRE(x && dy && dx && mean && rstd && dgamma && dbeta && ws && rows > 0 && C > 0, "bn_act_backward: bad arguments");
  static const int pnb = getenv("VFD_BN_PART_BLOCKS") ? atoi(getenv("VFD_BN_PART_BLOCKS")) : 512;
  const Tiling t = make_tiling(rows, C, pnb < BN_MAX_BLOCKS ? pnb : BN_MAX_BLOCKS);
  dim3 grid(t.gx, t.gy);
  float* part = reinterpret_cast<float*>(ws);
  hipStream_t st = as_stream(stream);
#define BN_BWD_P(T_, ACT_) hipLaunchKernelGGL((bn_act_bwd_partial_kernel<T_, ACT_, false>), grid, dim3(256), 0, st, (const T_*)x, (const T_*)dy, part, (long long)rows, C, t.TX, t.rows_per_block, mean, rstd, gamma, beta, slope)
  BN_ACT_DISPATCH(BN_BWD_P);
#undef BN_BWD_P
  VFD_CHECK_LAUNCH("bn_act_bwd_partial");
  hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3((C + 7) / 8), dim3(256), 0, st, part, t.gy, C, dgamma, dbeta, dgamma_acc, dbeta_acc);
  VFD_CHECK_LAUNCH("bn_bwd_finalize");
  const Tiling ta = make_stream_tiling(rows, C);
  dim3 grid2(ta.gx, ta.gy);
#define BN_BWD_A(T_, ACT_) hipLaunchKernelGGL((bn_act_bwd_apply_kernel<T_, ACT_>), grid2, dim3(256), 0, st, (const T_*)x, (const T_*)dy, (T_*)dx, (long long)rows, C, ta.TX, ta.rows_per_block, mean, rstd, gamma, beta, slope, dgamma, dbeta)
  BN_ACT_DISPATCH(BN_BWD_A);
#undef BN_BWD_A
  VFD_CHECK_LAUNCH("bn_act_bwd_apply");
  return VFD_OK;
}
