// Filter gradient of the convolution family as a split-K implicit GEMM on MFMA (gfx950).
//
//     dWp[r][t][c] = sum over pixels (n,q) of  S[n,q][r] * G[n, q*s - p + t][c]
//
// (S,G) = (dy, x) for a regular convolution and (x, dy) for a transposed one — in both cases S is indexed by
// the grid the stride multiplies and G is the gathered operand, so one kernel serves both.
// GEMM view: rows = channels of S, columns = (tap, channel of G) flattened, K = pixels.  Both operands are
// stored pixel-major (channels contiguous), i.e. K is the SLOW axis of both tiles: the bf16 path reads MFMA
// fragments with the gfx950 transposing LDS read (ds_read_b64_tr_b16), the f32 path with plain ds_read_b32.
// The pixel range is split over blockIdx.z; every split writes its own float32 slab (no atomics, bitwise
// reproducible); vfd_wgrad_reduce folds the slabs into the torch-layout gradient.
#include "common.hpp"

int vfd_conv_check_desc(const vfd_conv_desc* d);

namespace {

struct WgP {
  const void* S;
  const void* G;
  float* ws;
  int N;
  int Qd, Qh, Qw, Csp, Cs;      // grid and channels of S
  int Gd, Gh, Gw, Cgp;          // grid and padded channels of G
  int kd, kh, kw, sd, sh, sw, pd, ph, pw;
  int ncols;                    // kd*kh*kw*Cgp
  long long M;                  // N*Qd*Qh*Qw
  long long chunkM;             // pixels per split (multiple of the K step)
};

template <typename T> struct WgTraits;
template <> struct WgTraits<bf16_t> {
  static constexpr int KP = 32;       // pixels per K step
  static constexpr int PITCH = 288;   // bytes per pixel row of a 128-channel tile (256 + 32 pad)
};
template <> struct WgTraits<float> {
  static constexpr int KP = 16;
  static constexpr int PITCH = 576;   // 512 + 64 pad
};

// fragment loads: tile is [KP pixel rows][128 channels], `c0` = first channel of the 16-wide fragment
__device__ __forceinline__ bf16x8 frag_tr_bf16(const char* tile, int c0, int lane) {
  // lane = 16g + 4q + pp supplies the address of pixel row (4g + q [+16]), channels c0 + 4pp..+3;
  // it receives channel c0 + (lane & 15) of the 4 rows of its group -> MFMA k = 8g + j  (j<4: first read)
  const int g = lane >> 4, q = (lane >> 2) & 3, pp = lane & 3;
  const char* a0 = tile + (4 * g + q) * WgTraits<bf16_t>::PITCH + (c0 + 4 * pp) * 2;
  const char* a1 = a0 + 16 * WgTraits<bf16_t>::PITCH;
  typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
  const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(a0));
  const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(a1));
  typedef __attribute__((ext_vector_type(8))) short s16x8;
  const s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  return __builtin_bit_cast(bf16x8, v);
}

template <typename T> struct WgMma;
template <> struct WgMma<bf16_t> {
  __device__ static __forceinline__ void step(const char* st, const char* gt, int r0, int c0, int lane,
                                              f32x4 (&acc)[4][4]) {
    bf16x8 a[4], b[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) a[i] = frag_tr_bf16(st, r0 + i * 16, lane);
#pragma unroll
    for (int j = 0; j < 4; ++j) b[j] = frag_tr_bf16(gt, c0 + j * 16, lane);
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
  }
};
template <> struct WgMma<float> {
  __device__ static __forceinline__ void step(const char* st, const char* gt, int r0, int c0, int lane,
                                              f32x4 (&acc)[4][4]) {
    const int r = lane & 15, kq = lane >> 4;
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      float a[4], b[4];
      const int px = kk * 4 + kq;
#pragma unroll
      for (int i = 0; i < 4; ++i) a[i] = *reinterpret_cast<const float*>(st + px * WgTraits<float>::PITCH + (r0 + i * 16 + r) * 4);
#pragma unroll
      for (int j = 0; j < 4; ++j) b[j] = *reinterpret_cast<const float*>(gt + px * WgTraits<float>::PITCH + (c0 + j * 16 + r) * 4);
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i], b[j], acc[i][j], 0, 0, 0);
    }
  }
};

template <typename T>
__global__ __launch_bounds__(256) void conv_wgrad_kernel(const WgP p) {
  constexpr int VEC = Elem<T>::VEC;
  constexpr int KP = WgTraits<T>::KP;
  constexpr int PITCH = WgTraits<T>::PITCH;
  constexpr int CPR = 128 / VEC;         // 16-byte chunks per tile row (16 bf16 / 32 f32)
  constexpr int RPT = KP * CPR / 256;    // rows per thread (2)
  constexpr int RSTEP = 256 / CPR;       // pixel-row stride between a thread's rows
  constexpr int TILE_BYTES = KP * PITCH;

  __shared__ __attribute__((aligned(16))) char smem[2 * 2 * TILE_BYTES];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr0 = (wave & 1) * 64, wc0 = (wave >> 1) * 64;
  const int r0 = blockIdx.y * 128;    // first S channel of this tile
  const int nb0 = blockIdx.x * 128;   // first flattened column of this tile
  const long long mbeg = (long long)blockIdx.z * p.chunkM;
  long long mend = mbeg + p.chunkM;
  if (mend > p.M) mend = p.M;
  const int nsteps = (mbeg < mend) ? (int)((mend - mbeg + KP - 1) / KP) : 0;

  const int chunk = tid % CPR;
  const int prow = tid / CPR;
  // S column of this thread
  const int sc = r0 + chunk * VEC;
  const bool sc_ok = sc < p.Csp;
  // G column of this thread -> (tap, channel)
  const int col = nb0 + chunk * VEC;
  const bool col_ok = col < p.ncols;
  int gt_d = 0, gt_h = 0, gt_w = 0, gch = 0;
  if (col_ok) {
    int t = col / p.Cgp;
    gch = col - t * p.Cgp;
    gt_w = t % p.kw; t /= p.kw;
    gt_h = t % p.kh; gt_d = t / p.kh;
  }
  const int off_d = gt_d - p.pd, off_h = gt_h - p.ph, off_w = gt_w - p.pw;

  const T* __restrict__ Sg = reinterpret_cast<const T*>(p.S);
  const T* __restrict__ Gg = reinterpret_cast<const T*>(p.G);

  uint4 sreg[RPT], greg[RPT];
  auto load_global = [&](int s) {
#pragma unroll
    for (int i = 0; i < RPT; ++i) {
      const long long m = mbeg + (long long)s * KP + prow + RSTEP * i;
      uint4 sv = make_uint4(0, 0, 0, 0), gv = make_uint4(0, 0, 0, 0);
      if (m < mend) {
        if (sc_ok) sv = *reinterpret_cast<const uint4*>(Sg + (size_t)m * p.Csp + sc);
        if (col_ok) {
          unsigned q = (unsigned)m;
          const int qw = q % (unsigned)p.Qw; q /= (unsigned)p.Qw;
          const int qh = q % (unsigned)p.Qh; q /= (unsigned)p.Qh;
          const int qd = q % (unsigned)p.Qd; q /= (unsigned)p.Qd;
          const int gd = qd * p.sd + off_d, gh = qh * p.sh + off_h, gw = qw * p.sw + off_w;
          if ((unsigned)gd < (unsigned)p.Gd && (unsigned)gh < (unsigned)p.Gh && (unsigned)gw < (unsigned)p.Gw) {
            const size_t pix = ((size_t)((int)q * p.Gd + gd) * p.Gh + gh) * p.Gw + gw;
            gv = *reinterpret_cast<const uint4*>(Gg + pix * p.Cgp + gch);
          }
        }
      }
      sreg[i] = sv; greg[i] = gv;
    }
  };
  auto store_lds = [&](int buf) {
    char* st = smem + buf * 2 * TILE_BYTES;
    char* gt = st + TILE_BYTES;
#pragma unroll
    for (int i = 0; i < RPT; ++i) {
      const int row = prow + RSTEP * i;
      *reinterpret_cast<uint4*>(st + row * PITCH + chunk * 16) = sreg[i];
      *reinterpret_cast<uint4*>(gt + row * PITCH + chunk * 16) = greg[i];
    }
  };

  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  if (nsteps > 0) {
    load_global(0);
    store_lds(0);
    __syncthreads();
    for (int s = 0; s < nsteps; ++s) {
      const bool more = (s + 1) < nsteps;
      if (more) load_global(s + 1);
      const char* st = smem + (s & 1) * 2 * TILE_BYTES;
      WgMma<T>::step(st, st + TILE_BYTES, wr0, wc0, lane, acc);
      if (more) store_lds((s + 1) & 1);
      __syncthreads();
    }
  }

  // slab store: ws[z][r][col], D[row = r (lane>>4)*4+reg][col = lane&15]
  float* slab = p.ws + (size_t)blockIdx.z * p.Cs * p.ncols;
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int c = nb0 + wc0 + j * 16 + (lane & 15);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int rr = r0 + wr0 + i * 16 + (lane >> 4) * 4 + r;
        if (rr < p.Cs && c < p.ncols) slab[(size_t)rr * p.ncols + c] = acc[i][j][r];
      }
    }
}

// dw[a][b][t] = beta*dw + sum_z ws[z][a][t*Cbp + b].  64 elements x 4 split-lanes per workgroup: consecutive
// threads read consecutive slab columns (coalesced), every split-lane folds a quarter of the slabs, LDS sums the
// lanes in a fixed order (bitwise reproducible).
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* __restrict__ ws, float* __restrict__ dw, int A, int B,
                                                           int T, int Bp, int nsplit, float beta) {
  const long long total = (long long)A * T * B;
  const int el = threadIdx.x & 63, zl = threadIdx.x >> 6;
  const long long idx = (long long)blockIdx.x * 64 + el;
  const size_t ncols = (size_t)T * Bp;
  const size_t slab = (size_t)A * ncols;
  float s = 0.f;
  int a = 0, b = 0, t = 0;
  if (idx < total) {
    b = (int)(idx % B);
    const long long at = idx / B;
    t = (int)(at % T);
    a = (int)(at / T);
    const size_t off = (size_t)a * ncols + (size_t)t * Bp + b;
    for (int z = zl; z < nsplit; z += 4) s += ws[z * slab + off];
  }
  __shared__ float sh[4][64];
  sh[zl][el] = s;
  __syncthreads();
  if (zl == 0 && idx < total) {
    s = ((sh[0][el] + sh[1][el]) + sh[2][el]) + sh[3][el];
    float* dst = dw + ((size_t)a * B + b) * T + t;
    *dst = (beta != 0.f) ? beta * (*dst) + s : s;
  }
}

struct WgGeom {
  WgP p;
  int A, B, T, nsplit;
  size_t bytes;
};

int make_geom(const vfd_conv_desc* d, WgGeom& g) {
  int rc = vfd_conv_check_desc(d);
  if (rc != VFD_OK) return rc;
  WgP& p = g.p;
  p.N = d->N;
  if (!d->transposed) {  // S = dy, G = x
    p.Qd = d->Do; p.Qh = d->Ho; p.Qw = d->Wo; p.Cs = d->Cout;
    p.Gd = d->Di; p.Gh = d->Hi; p.Gw = d->Wi; g.B = d->Cin;
  } else {               // S = x, G = dy
    p.Qd = d->Di; p.Qh = d->Hi; p.Qw = d->Wi; p.Cs = d->Cin;
    p.Gd = d->Do; p.Gh = d->Ho; p.Gw = d->Wo; g.B = d->Cout;
  }
  p.Csp = cpad(p.Cs);
  p.Cgp = cpad(g.B);
  p.kd = d->kd; p.kh = d->kh; p.kw = d->kw; p.sd = d->sd; p.sh = d->sh; p.sw = d->sw;
  p.pd = d->pd; p.ph = d->ph; p.pw = d->pw;
  g.A = p.Cs;
  g.T = d->kd * d->kh * d->kw;
  p.ncols = g.T * p.Cgp;
  p.M = (long long)d->N * p.Qd * p.Qh * p.Qw;
  VFD_REQUIRE(p.M < 0x7fffffffLL, "wgrad: pixel count %lld exceeds 2^31", p.M);
  const int KP = d->dtype == VFD_BF16 ? 32 : 16;
  const long long tiles = (long long)((p.Cs + 127) / 128) * ((p.ncols + 127) / 128);
  long long nsplit = 2048 / tiles;
  const long long maxsplit = (p.M + 4 * KP - 1) / (4 * KP);
  if (nsplit > maxsplit) nsplit = maxsplit;
  if (nsplit > 256) nsplit = 256;
  if (nsplit < 1) nsplit = 1;
  const size_t slab = (size_t)p.Cs * p.ncols * sizeof(float);
  while (nsplit > 1 && slab * (size_t)nsplit > ((size_t)512 << 20)) nsplit /= 2;
  long long chunk = (p.M + nsplit - 1) / nsplit;
  chunk = (chunk + KP - 1) / KP * KP;
  nsplit = (p.M + chunk - 1) / chunk;
  p.chunkM = chunk;
  g.nsplit = (int)nsplit;
  g.bytes = slab * (size_t)nsplit;
  return VFD_OK;
}

}  // namespace

extern "C" int vfd_wgrad_workspace(const vfd_conv_desc* d, int32_t* nsplit, size_t* bytes) {
  WgGeom g;
  int rc = make_geom(d, g);
  if (rc != VFD_OK) return rc;
  if (nsplit) *nsplit = g.nsplit;
  if (bytes) *bytes = g.bytes;
  return VFD_OK;
}

extern "C" int vfd_conv_wgrad(const vfd_conv_desc* d, const void* x, const void* dy, void* ws, size_t ws_bytes,
                              void* stream) {
  WgGeom g;
  int rc = make_geom(d, g);
  if (rc != VFD_OK) return rc;
  VFD_REQUIRE(x && dy && ws, "wgrad: null pointer");
  if (ws_bytes < g.bytes) { vfd_set_error("wgrad: workspace %zu < %zu bytes", ws_bytes, g.bytes); return VFD_ENOSPACE; }
  g.p.S = d->transposed ? x : dy;
  g.p.G = d->transposed ? dy : x;
  g.p.ws = reinterpret_cast<float*>(ws);
  dim3 grid((g.p.ncols + 127) / 128, (g.p.Cs + 127) / 128, g.nsplit);
  VFD_REQUIRE(grid.y <= 65535u && grid.z <= 65535u, "wgrad: grid too large");
  if (d->dtype == VFD_BF16)
    hipLaunchKernelGGL(conv_wgrad_kernel<bf16_t>, grid, dim3(256), 0, as_stream(stream), g.p);
  else
    hipLaunchKernelGGL(conv_wgrad_kernel<float>, grid, dim3(256), 0, as_stream(stream), g.p);
  VFD_CHECK_LAUNCH("conv_wgrad");
  return VFD_OK;
}

extern "C" int vfd_wgrad_reduce(const vfd_conv_desc* d, const void* ws, float* dw, float beta, void* stream) {
  WgGeom g;
  int rc = make_geom(d, g);
  if (rc != VFD_OK) return rc;
  VFD_REQUIRE(ws && dw, "wgrad_reduce: null pointer");
  const long long total = (long long)g.A * g.T * g.B;
  const long long blocks = (total + 63) / 64;
  hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)blocks), dim3(256), 0, as_stream(stream),
                     reinterpret_cast<const float*>(ws), dw, g.A, g.B, g.T, g.p.Cgp, g.nsplit, beta);
  VFD_CHECK_LAUNCH("wgrad_reduce");
  return VFD_OK;
}
