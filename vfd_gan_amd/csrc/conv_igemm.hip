// Implicit-GEMM convolution / transposed convolution on MFMA for gfx950.
//
// One kernel family serves nn.Conv3d, nn.ConvTranspose3d, nn.Conv2d, nn.ConvTranspose2d and nn.Linear of
// the reference (see include/vfdgan_hip.h for the call sites).  GEMM view, per output class:
//
//     Y[channel][pixel] = sum_K  Wp[channel][K] * Xcol[pixel][K]       K = (tap, input channel)
//
// Both operands are K-contiguous in memory (Wp is pre-packed [Cout][tap][Cip]; a pixel's input channels are
// contiguous in the channels-last activation block), so both LDS tiles are [row][64 bytes of K] and are read
// back as MFMA fragments with ds_read_b128.  Channels are the MFMA "row" dimension, so that every lane ends
// up with 4 consecutive output channels of ONE pixel: the epilogue packs them into one 8/16-byte store.
//
// Transposed convolutions (and the data gradient of strided convolutions) are run gather-form: output voxels
// are split into stride^3 parity classes (blockIdx.z); within a class only the taps congruent to (o+p) mod s
// contribute, so the class is a dense convolution with ceil(k/s) taps per dim and unit input stride.
//
//     regular    : in = q*s - p + t            (t = tap, all k taps)          out = q
//     transposed : in = q + c0 - t,  k = k0 + t*s, k0 = (r+p)%s, c0 = (r+p-k0)/s,  out = q*s + r
#include "common.hpp"

namespace {

struct ConvP {
  const void* x;
  const void* w;
  void* y;
  const float* bias;
  float* stats;  // [2][Cop] or null
  int N, Di, Hi, Wi, Cip;
  int Do, Ho, Wo, Cop, Cout;
  int kd, kh, kw, sd, sh, sw, pd, ph, pw;
  int transposed;
  int Kw;  // packed filter row length = kd*kh*kw*Cip
  int act;
  float slope;
  int ksplit;   // >1: K range split over blockIdx.z, raw f32 partial tiles go to ws[ksplit][M][Cop]
  float* ws;
};

struct DimClass {  // per-dimension description of the taps of one output class
  int nk;   // number of taps
  int k0;   // first filter index
  int ks;   // filter index step
  int c0;   // input coordinate offset
  int cs;   // input coordinate step per tap (+1 regular, -1 transposed)
  int a;    // input coordinate multiplier of q
  int so;   // output coordinate multiplier of q
  int r;    // output coordinate offset
  int Q;    // number of q along this dim
};

__device__ __forceinline__ DimClass make_dim(int transposed, int r, int k, int s, int p, int O) {
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
  return d;
}

// LDS tile: rows of 64 bytes (4 chunks of 16 B); chunk index XOR-swizzled by the row so that the ds_read_b128
// fragment reads (16 rows x 4 chunks per wave-instruction) are bank-conflict free.
__device__ __forceinline__ int lds_off(int row, int chunk) {
  const int g = (row >> 2) & 3;
  const int sw = (((g ^ (g >> 1)) & 1) << 1) | (g >> 1);
  return row * 64 + ((chunk ^ sw) << 4);
}

template <typename T> struct Mma;
template <> struct Mma<bf16_t> {
  // one K-step = 32 bf16 = one v_mfma_f32_16x16x32_bf16 per 16x16 tile
  template <int NI, int NJ>
  __device__ static __forceinline__ void step(const char* wt, const char* pt, int wrow0, int prow0, int lane,
                                              f32x4 (&acc)[NI][NJ]) {
    bf16x8 a[NI], b[NJ];
    const int r = lane & 15, ch = lane >> 4;
#pragma unroll
    for (int i = 0; i < NI; ++i) a[i] = *reinterpret_cast<const bf16x8*>(wt + lds_off(wrow0 + i * 16 + r, ch));
#pragma unroll
    for (int j = 0; j < NJ; ++j) b[j] = *reinterpret_cast<const bf16x8*>(pt + lds_off(prow0 + j * 16 + r, ch));
#pragma unroll
    for (int i = 0; i < NI; ++i)
#pragma unroll
      for (int j = 0; j < NJ; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
  }
};
template <> struct Mma<float> {
  // one K-step = 16 f32 = four v_mfma_f32_16x16x4_f32 per 16x16 tile (exact f32 FMA chain)
  template <int NI, int NJ>
  __device__ static __forceinline__ void step(const char* wt, const char* pt, int wrow0, int prow0, int lane,
                                              f32x4 (&acc)[NI][NJ]) {
    const int r = lane & 15, kq = lane >> 4;
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      float a[NI], b[NJ];
#pragma unroll
      for (int i = 0; i < NI; ++i) a[i] = *reinterpret_cast<const float*>(wt + lds_off(wrow0 + i * 16 + r, kk) + kq * 4);
#pragma unroll
      for (int j = 0; j < NJ; ++j) b[j] = *reinterpret_cast<const float*>(pt + lds_off(prow0 + j * 16 + r, kk) + kq * 4);
#pragma unroll
      for (int i = 0; i < NI; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i], b[j], acc[i][j], 0, 0, 0);
    }
  }
};

template <typename T, int WAVES_C, int WAVES_P, int NI, int NJ>
__global__ __launch_bounds__(256) void conv_igemm_kernel(const ConvP p) {
  constexpr int TILE_C = WAVES_C * NI * 16;
  constexpr int TILE_P = WAVES_P * NJ * 16;
  constexpr int VEC = Elem<T>::VEC;
  constexpr int BK = 4 * VEC;
  constexpr int WL = (TILE_C * 4 + 255) / 256;  // 16-byte chunks of the filter tile per thread
  constexpr int PL = (TILE_P * 4 + 255) / 256;  // 16-byte chunks of the pixel tile per thread
  static_assert(WAVES_C * WAVES_P == 4, "4 waves per workgroup");

  __shared__ __attribute__((aligned(16))) char smem[2 * (TILE_C + TILE_P) * 64];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wave_c0 = (wave % WAVES_C) * (NI * 16);
  const int wave_p0 = (wave / WAVES_C) * (NJ * 16);

  // ---- output class of this workgroup --------------------------------------------------------------
  int cls = blockIdx.z / p.ksplit;
  const int ksl = blockIdx.z - cls * p.ksplit;
  const int rw = p.transposed ? cls % p.sw : 0;
  if (p.transposed) cls /= p.sw;
  const int rh = p.transposed ? cls % p.sh : 0;
  if (p.transposed) cls /= p.sh;
  const int rd = p.transposed ? cls : 0;
  const DimClass dd = make_dim(p.transposed, rd, p.kd, p.sd, p.pd, p.Do);
  const DimClass dh = make_dim(p.transposed, rh, p.kh, p.sh, p.ph, p.Ho);
  const DimClass dw = make_dim(p.transposed, rw, p.kw, p.sw, p.pw, p.Wo);
  const long long Mcls = (long long)p.N * dd.Q * dh.Q * dw.Q;
  const long long m0 = (long long)blockIdx.x * TILE_P;
  if (m0 >= Mcls) return;  // uniform per workgroup
  const int n0 = blockIdx.y * TILE_C;
  const int ntaps = dd.nk * dh.nk * dw.nk;
  const int Kcls = ntaps * p.Cip;
  const int nsteps_all = (Kcls + BK - 1) / BK;
  const int steps_per = (nsteps_all + p.ksplit - 1) / p.ksplit;
  const int s_begin = ksl * steps_per;
  const int nsteps = max(0, min(nsteps_all - s_begin, steps_per));

  // ---- per-thread load bookkeeping ----------------------------------------------------------------------
  const int chunk = tid & 3;
  const int rbase = tid >> 2;  // 0..63
  // position of this thread's chunk inside the flattened K axis
  int kc, td, th, tw;
  {
    const int kflat = s_begin * BK + chunk * VEC;
    int t = kflat / p.Cip;
    kc = kflat - t * p.Cip;
    tw = (dw.nk > 0) ? t % dw.nk : 0;
    t = (dw.nk > 0) ? t / dw.nk : 0;
    th = (dh.nk > 0) ? t % dh.nk : 0;
    td = (dh.nk > 0) ? t / dh.nk : 0;
  }
  // pixel rows handled by this thread
  int pn[PL], pid[PL], pih[PL], piw[PL];
#pragma unroll
  for (int i = 0; i < PL; ++i) {
    const long long m = m0 + rbase + 64 * i;
    if (m < Mcls && (rbase + 64 * i) < TILE_P) {
      long long q = m;
      const int qw = (int)(q % dw.Q); q /= dw.Q;
      const int qh = (int)(q % dh.Q); q /= dh.Q;
      const int qd = (int)(q % dd.Q); q /= dd.Q;
      pn[i] = (int)q;
      pid[i] = qd * dd.a + dd.c0;
      pih[i] = qh * dh.a + dh.c0;
      piw[i] = qw * dw.a + dw.c0;
    } else {
      pn[i] = -1; pid[i] = 0; pih[i] = 0; piw[i] = 0;
    }
  }
  const T* __restrict__ xg = reinterpret_cast<const T*>(p.x);
  const T* __restrict__ wg = reinterpret_cast<const T*>(p.w);

  uint4 wreg[WL], preg[PL];
  auto load_global = [&]() {
    const bool kvalid = td < dd.nk && ntaps > 0;
    const int tapidx = ((dd.k0 + td * dd.ks) * p.kh + (dh.k0 + th * dh.ks)) * p.kw + (dw.k0 + tw * dw.ks);
#pragma unroll
    for (int i = 0; i < WL; ++i) {
      const int row = rbase + 64 * i;
      const int co = n0 + row;
      uint4 v = make_uint4(0, 0, 0, 0);
      if (kvalid && row < TILE_C && co < p.Cout)
        v = *reinterpret_cast<const uint4*>(wg + (size_t)co * p.Kw + (size_t)tapidx * p.Cip + kc);
      wreg[i] = v;
    }
    const int od = td * dd.cs, oh = th * dh.cs, ow = tw * dw.cs;
#pragma unroll
    for (int i = 0; i < PL; ++i) {
      const int id = pid[i] + od, ih = pih[i] + oh, iw = piw[i] + ow;
      uint4 v = make_uint4(0, 0, 0, 0);
      if (kvalid && pn[i] >= 0 && (unsigned)id < (unsigned)p.Di && (unsigned)ih < (unsigned)p.Hi &&
          (unsigned)iw < (unsigned)p.Wi) {
        const size_t pix = ((size_t)(pn[i] * p.Di + id) * p.Hi + ih) * p.Wi + iw;
        v = *reinterpret_cast<const uint4*>(xg + pix * p.Cip + kc);
      }
      preg[i] = v;
    }
  };
  auto advance_k = [&]() {
    kc += BK;
    while (kc >= p.Cip) {
      kc -= p.Cip;
      if (++tw >= dw.nk) {
        tw = 0;
        if (++th >= dh.nk) { th = 0; ++td; }
      }
    }
  };
  auto store_lds = [&](int buf) {
    char* wt = smem + buf * (TILE_C + TILE_P) * 64;
    char* pt = wt + TILE_C * 64;
#pragma unroll
    for (int i = 0; i < WL; ++i) {
      const int row = rbase + 64 * i;
      if (row < TILE_C) *reinterpret_cast<uint4*>(wt + lds_off(row, chunk)) = wreg[i];
    }
#pragma unroll
    for (int i = 0; i < PL; ++i) {
      const int row = rbase + 64 * i;
      if (row < TILE_P) *reinterpret_cast<uint4*>(pt + lds_off(row, chunk)) = preg[i];
    }
  };

  f32x4 acc[NI][NJ];
#pragma unroll
  for (int i = 0; i < NI; ++i)
#pragma unroll
    for (int j = 0; j < NJ; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  if (nsteps > 0) {
    load_global();
    store_lds(0);
    __syncthreads();
    for (int s = 0; s < nsteps; ++s) {
      const bool more = (s + 1) < nsteps;
      if (more) {
        advance_k();
        load_global();
      }
      const char* wt = smem + (s & 1) * (TILE_C + TILE_P) * 64;
      const char* pt = wt + TILE_C * 64;
      Mma<T>::template step<NI, NJ>(wt, pt, wave_c0, wave_p0, lane, acc);
      if (more) store_lds((s + 1) & 1);
      __syncthreads();
    }
  }

  const int cq = (lane >> 4) * 4;
  if (p.ksplit > 1) {
    // split-K: raw partial sums, [ksplit][M][Cop] float32; bias / activation / store happen in conv_splitk_finish
    float* slab = p.ws + (size_t)ksl * (size_t)Mcls * p.Cop;
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      const long long m = m0 + wave_p0 + j * 16 + (lane & 15);
#pragma unroll
      for (int i = 0; i < NI; ++i) {
        const int c = n0 + wave_c0 + i * 16 + cq;
        if (m < Mcls && c < p.Cop)
          *reinterpret_cast<float4*>(slab + (size_t)m * p.Cop + c) = make_float4(acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]);
      }
    }
    return;
  }

  // ---- epilogue: bias, statistics, activation, channels-last store --------------------------------------
  T* __restrict__ yg = reinterpret_cast<T*>(p.y);
  float ssum[NI][4], ssq[NI][4];
#pragma unroll
  for (int i = 0; i < NI; ++i)
#pragma unroll
    for (int r = 0; r < 4; ++r) { ssum[i][r] = 0.f; ssq[i][r] = 0.f; }

#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    const long long m = m0 + wave_p0 + j * 16 + (lane & 15);
    const bool mvalid = m < Mcls;
    size_t opix = 0;
    if (mvalid) {
      long long q = m;
      const int qw = (int)(q % dw.Q); q /= dw.Q;
      const int qh = (int)(q % dh.Q); q /= dh.Q;
      const int qd = (int)(q % dd.Q); q /= dd.Q;
      const int n = (int)q;
      opix = ((size_t)(n * p.Do + qd * dd.so + dd.r) * p.Ho + qh * dh.so + dh.r) * p.Wo + qw * dw.so + dw.r;
    }
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const int c = n0 + wave_c0 + i * 16 + cq;
      float v[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float t = acc[i][j][r];
        if (p.bias != nullptr && (c + r) < p.Cout) t += p.bias[c + r];
        if (p.stats != nullptr && mvalid) { ssum[i][r] += t; ssq[i][r] += t * t; }
        v[r] = act_apply(t, p.act, p.slope);
        if ((c + r) >= p.Cout) v[r] = 0.f;  // keep pad channels zero (sigmoid(0) != 0)
      }
      if (mvalid && c < p.Cop) {
        T* dst = yg + opix * p.Cop + c;
        if constexpr (sizeof(T) == 2) {
          uint2 o; o.x = pack2bf(v[0], v[1]); o.y = pack2bf(v[2], v[3]);
          *reinterpret_cast<uint2*>(dst) = o;
        } else {
          *reinterpret_cast<float4*>(dst) = make_float4(v[0], v[1], v[2], v[3]);
        }
      }
    }
  }
  if (p.stats != nullptr) {
#pragma unroll
    for (int i = 0; i < NI; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float a = ssum[i][r], b = ssq[i][r];
#pragma unroll
        for (int o = 1; o < 16; o <<= 1) { a += __shfl_xor(a, o, 64); b += __shfl_xor(b, o, 64); }
        const int c = n0 + wave_c0 + i * 16 + cq + r;
        if ((lane & 15) == 0 && c < p.Cout) {
          atomicAdd(p.stats + c, a);
          atomicAdd(p.stats + p.Cop + c, b);
        }
      }
  }
}

// y[m][c] = act(sum_ks ws[ks][m][c] + bias[c]) for the split-K path (regular convolutions only: out pixel == m)
template <typename T>
__global__ void conv_splitk_finish_kernel(const float* __restrict__ ws, T* __restrict__ y, const float* __restrict__ bias,
                                          long long M, int Cout, int Cop, int ksplit, int act, float slope) {
  const int GR = Cop >> 3;
  const long long total = M * GR;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int g = (int)(i % GR);
    float v[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int ks = 0; ks < ksplit; ++ks) {
      float t[8];
      load8(ws + ((size_t)ks * M * GR + i) * 8, t);
#pragma unroll
      for (int k = 0; k < 8; ++k) v[k] += t[k];
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int c = g * 8 + k;
      v[k] = (c < Cout) ? act_apply(v[k] + (bias != nullptr ? bias[c] : 0.f), act, slope) : 0.f;
    }
    store8(y + i * 8, v);
  }
}

template <int TILE_C, int TILE_P>
int pick_ksplit(const ConvP& p, long long M, int bk) {
  if (p.transposed || p.stats != nullptr) return 1;
  const long long blocks = ((M + TILE_P - 1) / TILE_P) * ((p.Cout + TILE_C - 1) / TILE_C);
  const int nsteps = (p.Kw + bk - 1) / bk;
  if (blocks >= 128 || nsteps < 32) return 1;
  long long ks = 512 / blocks;
  if (ks > nsteps / 8) ks = nsteps / 8;
  if (ks > 64) ks = 64;
  return ks < 2 ? 1 : (int)ks;
}

template <typename T, int WAVES_C, int WAVES_P, int NI, int NJ>
int launch_cfg(const ConvP& p, long long maxM, int ncls, hipStream_t st, size_t ws_bytes, size_t* ws_query) {
  constexpr int TILE_C = WAVES_C * NI * 16;
  constexpr int TILE_P = WAVES_P * NJ * 16;
  const long long mb = (maxM + TILE_P - 1) / TILE_P;
  if (mb <= 0) return VFD_OK;
  if (mb > 0x7fffffffLL) { vfd_set_error("conv: too many pixel blocks"); return VFD_EINVAL; }
  ConvP q = p;
  q.ksplit = pick_ksplit<TILE_C, TILE_P>(p, maxM, 4 * Elem<T>::VEC);
  const size_t need = q.ksplit > 1 ? (size_t)q.ksplit * (size_t)maxM * p.Cop * sizeof(float) : 0;
  if (ws_query != nullptr) { *ws_query = need; return VFD_OK; }
  if (q.ksplit > 1 && (p.ws == nullptr || ws_bytes < need)) q.ksplit = 1;   // no workspace: plain path
  dim3 grid((unsigned)mb, (unsigned)((p.Cout + TILE_C - 1) / TILE_C), (unsigned)(ncls * q.ksplit));
  if (grid.y > 65535u || grid.z > 65535u) { vfd_set_error("conv: grid too large"); return VFD_EINVAL; }
  hipLaunchKernelGGL((conv_igemm_kernel<T, WAVES_C, WAVES_P, NI, NJ>), grid, dim3(256), 0, st, q);
  VFD_CHECK_LAUNCH("conv_igemm");
  if (q.ksplit > 1) {
    const long long total = maxM * (p.Cop >> 3);
    long long nb = (total + 255) / 256;
    if (nb > 4096) nb = 4096;
    hipLaunchKernelGGL(conv_splitk_finish_kernel<T>, dim3((unsigned)nb), dim3(256), 0, st, q.ws, reinterpret_cast<T*>(p.y), p.bias,
                       maxM, p.Cout, p.Cop, q.ksplit, p.act, p.slope);
    VFD_CHECK_LAUNCH("conv_splitk_finish");
  }
  return VFD_OK;
}

template <typename T>
int launch(const ConvP& p, long long maxM, int ncls, hipStream_t st, size_t ws_bytes, size_t* ws_query) {
  if (p.Cout > 64) return launch_cfg<T, 2, 2, 4, 4>(p, maxM, ncls, st, ws_bytes, ws_query);   // 128 ch x 128 px
  if (p.Cout > 32) return launch_cfg<T, 1, 4, 4, 4>(p, maxM, ncls, st, ws_bytes, ws_query);   //  64 ch x 256 px
  if (p.Cout > 16) return launch_cfg<T, 1, 4, 2, 4>(p, maxM, ncls, st, ws_bytes, ws_query);   //  32 ch x 256 px
  return launch_cfg<T, 1, 4, 1, 4>(p, maxM, ncls, st, ws_bytes, ws_query);                    //  16 ch x 256 px
}

}  // namespace

int vfd_conv_check_desc(const vfd_conv_desc* d) {
  VFD_REQUIRE(d != nullptr, "conv: null descriptor");
  VFD_REQUIRE(d->dtype == VFD_F32 || d->dtype == VFD_BF16, "conv: bad dtype %d", d->dtype);
  VFD_REQUIRE(d->N > 0 && d->Di > 0 && d->Hi > 0 && d->Wi > 0 && d->Cin > 0, "conv: bad input dims");
  VFD_REQUIRE(d->Do > 0 && d->Ho > 0 && d->Wo > 0 && d->Cout > 0, "conv: bad output dims");
  VFD_REQUIRE(d->kd > 0 && d->kh > 0 && d->kw > 0 && d->sd > 0 && d->sh > 0 && d->sw > 0, "conv: bad filter/stride");
  VFD_REQUIRE(d->pd >= 0 && d->ph >= 0 && d->pw >= 0, "conv: negative padding");
  const int32_t k[3] = {d->kd, d->kh, d->kw}, s[3] = {d->sd, d->sh, d->sw}, pp[3] = {d->pd, d->ph, d->pw};
  const int32_t I[3] = {d->Di, d->Hi, d->Wi}, O[3] = {d->Do, d->Ho, d->Wo};
  for (int i = 0; i < 3; ++i) {
    if (!d->transposed) {
      VFD_REQUIRE(O[i] == (I[i] + 2 * pp[i] - k[i]) / s[i] + 1 && I[i] + 2 * pp[i] >= k[i],
                  "conv: output extent %d inconsistent with input %d k %d s %d p %d", O[i], I[i], k[i], s[i], pp[i]);
    } else {
      const int base = (I[i] - 1) * s[i] - 2 * pp[i] + k[i];
      // output_padding in [0, s) (a lone input voxel admits any stride, see DESIGN.md "1x1 inputs")
      VFD_REQUIRE(O[i] >= base && (O[i] - base < s[i] || I[i] == 1) && base > 0,
                  "convT: output extent %d inconsistent with input %d k %d s %d p %d", O[i], I[i], k[i], s[i], pp[i]);
    }
  }
  return VFD_OK;
}

static int conv_dispatch(const vfd_conv_desc* d_in, const void* x, const void* packed, const float* bias, void* y, float* stats,
                         void* ws, size_t ws_bytes, size_t* ws_query, void* stream) {
  VFD_REQUIRE(d_in != nullptr, "conv: null descriptor");
  vfd_conv_desc dn = *d_in;
  if (dn.transposed) {
    // A transposed convolution of a single input voxel along a dimension is stride-agnostic there
    // (out[o] = in[0] * w[o+p]).  Re-labelling its stride as k gives k one-tap classes instead of one class that
    // walks k taps of which k-1 fall outside the input (ganomaly Decoder initial ConvTranspose2d(nz, c, k, 1, 0)).
    if (dn.Di == 1) dn.sd = dn.kd;
    if (dn.Hi == 1) dn.sh = dn.kh;
    if (dn.Wi == 1) dn.sw = dn.kw;
  }
  const vfd_conv_desc* d = &dn;
  int rc = vfd_conv_check_desc(d);
  if (rc != VFD_OK) return rc;
  if (ws_query == nullptr) {
    VFD_REQUIRE(x && packed && y, "conv: null tensor pointer");
    VFD_REQUIRE((((uintptr_t)x | (uintptr_t)packed | (uintptr_t)y | (uintptr_t)ws) & 15) == 0, "conv: tensors must be 16-byte aligned");
  }
  ConvP p;
  p.x = x; p.w = packed; p.y = y; p.bias = bias; p.stats = stats;
  p.N = d->N; p.Di = d->Di; p.Hi = d->Hi; p.Wi = d->Wi; p.Cip = cpad(d->Cin);
  p.Do = d->Do; p.Ho = d->Ho; p.Wo = d->Wo; p.Cop = cpad(d->Cout); p.Cout = d->Cout;
  p.kd = d->kd; p.kh = d->kh; p.kw = d->kw; p.sd = d->sd; p.sh = d->sh; p.sw = d->sw;
  p.pd = d->pd; p.ph = d->ph; p.pw = d->pw;
  p.transposed = d->transposed;
  p.Kw = d->kd * d->kh * d->kw * p.Cip;
  p.act = d->act; p.slope = d->slope;
  p.ksplit = 1; p.ws = reinterpret_cast<float*>(ws);
  long long maxM;
  int ncls = 1;
  if (!d->transposed) {
    maxM = (long long)d->N * d->Do * d->Ho * d->Wo;
  } else {
    ncls = d->sd * d->sh * d->sw;
    const long long qd = (d->Do + d->sd - 1) / d->sd, qh = (d->Ho + d->sh - 1) / d->sh, qw = (d->Wo + d->sw - 1) / d->sw;
    maxM = (long long)d->N * qd * qh * qw;
  }
  hipStream_t st = as_stream(stream);
  return d->dtype == VFD_BF16 ? launch<bf16_t>(p, maxM, ncls, st, ws_bytes, ws_query) : launch<float>(p, maxM, ncls, st, ws_bytes, ws_query);
}

extern "C" int vfd_conv_workspace(const vfd_conv_desc* d, int want_stats, size_t* bytes) {
  VFD_REQUIRE(bytes != nullptr, "conv_workspace: null result pointer");
  float dummy;
  return conv_dispatch(d, nullptr, nullptr, nullptr, nullptr, want_stats ? &dummy : nullptr, nullptr, 0, bytes, nullptr);
}

extern "C" int vfd_conv_forward(const vfd_conv_desc* d, const void* x, const void* packed, const float* bias, void* y,
                                float* stats, void* ws, size_t ws_bytes, void* stream) {
  return conv_dispatch(d, x, packed, bias, y, stats, ws, ws_bytes, nullptr, stream);
}
