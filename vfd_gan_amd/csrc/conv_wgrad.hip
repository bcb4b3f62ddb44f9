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
#include <stdlib.h>

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
  FastDiv fQw, fQh, fQd;        // pixel index -> (n, qd, qh, qw) without integer division
  int ntx, nty, nsplit, xcd_order;   // column tiles, row tiles, pixel splits; workgroup order (see kernel)
};

// Tile geometry: [KP pixel rows][128 channels], rows of ROWB bytes = CPR 16-byte chunks, no padding (the tiles are
// written by LDS-DMA, which is lane-linear); bank conflicts of the K-strided fragment reads are removed by XOR-ing
// the chunk index with a function of the pixel row, applied on the DMA's SOURCE side.
template <typename T> struct WgTraits;
template <> struct WgTraits<bf16_t> {
  static constexpr int KP = 32;       // pixels per K step
  static constexpr int SWZ_PERIOD = 8;   // swz(row) depends on row & 7
  static constexpr int ROWB = 256;
  // ds_read_b64_tr_b16: a half-wave touches 8 pixel rows x 32 B.  Rows of >= 256 B all start on bank 0: chunk ^ 2*(row&7)
  // spreads the 8 rows over all 64 banks.  128-byte rows alternate between the two halves of the 256-byte bank row, so
  // the 4 rows of one parity must take 4 different 32-byte slots of their half: chunk ^ 2*((row>>1)&3).
  template <int ROWB>
  __device__ static __forceinline__ int swz(int row) { return ROWB >= 256 ? 2 * (row & 7) : 2 * ((row >> 1) & 3); }
};
template <> struct WgTraits<float> {
  static constexpr int KP = 16;
  static constexpr int SWZ_PERIOD = 2;
  static constexpr int ROWB = 512;
  // ds_read_b32: a half-wave touches 2 pixel rows x 64 B; chunk ^ 4*(row&1) puts them on different 64-byte slots
  template <int ROWB>
  __device__ static __forceinline__ int swz(int row) { return 4 * (row & 1); }
};

// fragment loads: `c0` = first channel of the 16-wide fragment
template <int ROWB>
__device__ __forceinline__ bf16x8 frag_tr_bf16(const char* tile, int c0, int lane) {
  // lane = 16g + 4q + pp supplies the address of pixel row (4g + q [+16]), channels c0 + 4pp..+3;
  // it receives channel c0 + (lane & 15) of the 4 rows of its group -> MFMA k = 8g + j  (j<4: first read)
  const int g = lane >> 4, q = (lane >> 2) & 3, pp = lane & 3;
  const int row = 4 * g + q;
  const int ch = c0 + 4 * pp;
  const int off = row * ROWB + ((((ch >> 3) ^ WgTraits<bf16_t>::swz<ROWB>(row)) << 4) | ((ch & 4) << 1));
  typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
  const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(tile + off));
  const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(tile + off + 16 * ROWB));   // row + 16: same row & 7
  typedef __attribute__((ext_vector_type(8))) short s16x8;
  const s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  return __builtin_bit_cast(bf16x8, v);
}

template <typename T> struct WgMma;
template <> struct WgMma<bf16_t> {
  template <int SROWB, int GROWB>
  __device__ static __forceinline__ void step(const char* st, const char* gt, int r0, int c0, int lane,
                                              f32x4 (&acc)[4][4]) {
    bf16x8 a[4], b[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) a[i] = frag_tr_bf16<SROWB>(st, r0 + i * 16, lane);
#pragma unroll
    for (int j = 0; j < 4; ++j) b[j] = frag_tr_bf16<GROWB>(gt, c0 + j * 16, lane);
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
  }
};
template <> struct WgMma<float> {
  template <int ROWB>
  __device__ static __forceinline__ int off(int px, int c) {
    return px * ROWB + ((((c >> 2) ^ WgTraits<float>::swz<ROWB>(px)) << 4) | ((c & 3) << 2));
  }
  template <int SROWB, int GROWB>
  __device__ static __forceinline__ void step(const char* st, const char* gt, int r0, int c0, int lane,
                                              f32x4 (&acc)[4][4]) {
    const int r = lane & 15, kq = lane >> 4;
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      float a[4], b[4];
      const int px = kk * 4 + kq;
#pragma unroll
      for (int i = 0; i < 4; ++i) a[i] = *reinterpret_cast<const float*>(st + off<SROWB>(px, r0 + i * 16 + r));
#pragma unroll
      for (int j = 0; j < 4; ++j) b[j] = *reinterpret_cast<const float*>(gt + off<GROWB>(px, c0 + j * 16 + r));
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i], b[j], acc[i][j], 0, 0, 0);
    }
  }
};

__device__ uint4 g_wg_zero_page[4];   // source of every masked 16-byte chunk (see conv_igemm.hip)

// WR x WC waves of 64 x 64 outputs each: tile = 64*WR channels of S x 64*WC flattened columns of G.
// 2 x 2 (4 waves, 3 workgroups per CU) for narrow layers, 1 x 4 (64 x 256) when S has <= 64 channels; 4 x 2 (8 waves, 2 per CU) where S has >= 256 channels and
// 2 x 4 where it has 65..128: 24 KiB of DMA per 32-pixel step instead of 16 KiB for twice the MFMA work.
// 4 x 4 (16 waves, ONE workgroup per CU, 256 x 256: round 3): 32 KiB of DMA per 32-pixel step for 1024 MFMA cycles per SIMD =
// 31 B/clk/CU, against 47 B/clk/CU for 4 x 2 — the same step that took conv_igemm's 256-channel tile off the DMA ceiling.
// (launch bound: the 64 x 256 tile's 4 waves need ~140 registers, i.e. 2 workgroups' worth per SIMD, not 3)
template <typename T, int STAGES, int WR, int WC>
__global__ __launch_bounds__(64 * WR * WC, WR * WC == 4 ? (WR == 1 ? 2 : 3) : 4) void conv_wgrad_kernel(const WgP p) {
  constexpr int VEC = Elem<T>::VEC;
  constexpr int KP = WgTraits<T>::KP;
  constexpr int NW = WR * WC;                 // waves
  constexpr int TR = 64 * WR, TC = 64 * WC;   // tile rows (S channels), tile columns
  constexpr int SROWB = TR * (int)sizeof(T), GROWB = TC * (int)sizeof(T);   // bytes per pixel row of the two tiles
  constexpr int S_CPR = SROWB / 16, G_CPR = GROWB / 16;   // 16-byte chunks per tile row
  constexpr int S_RPI = 64 / S_CPR, G_RPI = 64 / G_CPR;   // pixel rows per 1-KiB DMA wave-instruction
  constexpr int S_NT = KP / S_RPI / NW, G_NT = KP / G_RPI / NW;   // DMA instructions per wave, tile and stage
  constexpr int S_BYTES = KP * SROWB, G_BYTES = KP * GROWB;
  constexpr int STAGE_BYTES = S_BYTES + G_BYTES;
  static_assert(S_NT >= 1 && G_NT >= 1 && S_NT * S_RPI * NW == KP && G_NT * G_RPI * NW == KP, "tile shape");
  // the source-side swizzle of a lane must not depend on the instruction index i (row = RPI * (wave + NW * i) + lrow)
  static_assert((S_RPI * NW) % WgTraits<T>::SWZ_PERIOD == 0 && (G_RPI * NW) % WgTraits<T>::SWZ_PERIOD == 0, "swizzle period");

  __shared__ __attribute__((aligned(16))) char smem[STAGES * STAGE_BYTES];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr0 = (wave % WR) * 64, wc0 = (wave / WR) * 64;
  // Workgroups go round-robin to the 8 XCDs (id % 8).  All tiles of one pixel split read the same S and G pixels, so
  // a split's tiles get ids congruent mod 8 and adjacent in dispatch order: one L2 fetches that pixel range once.
  int split, tile;
  const int ntiles = p.ntx * p.nty;
  if (p.xcd_order) {
    const int slot = blockIdx.x >> 3;
    const int sq = slot / ntiles;
    tile = slot - sq * ntiles;
    split = sq * 8 + (blockIdx.x & 7);
    if (split >= p.nsplit) return;
  } else {
    split = blockIdx.x / ntiles;
    tile = blockIdx.x - split * ntiles;
  }
  const int ty_ = tile / p.ntx, tx_ = tile - ty_ * p.ntx;
  const int r0 = ty_ * TR;     // first S channel of this tile
  const int nb0 = tx_ * TC;    // first flattened column of this tile
  const long long mbeg = (long long)split * p.chunkM;
  long long mend = mbeg + p.chunkM;
  if (mend > p.M) mend = p.M;
  const int nsteps = (mbeg < mend) ? (int)((mend - mbeg + KP - 1) / KP) : 0;

  // DMA bookkeeping: lane -> (pixel row within a group, physical slot); logical chunk = slot ^ swz(row)
  const int s_lrow = lane / S_CPR, g_lrow = lane / G_CPR;
  const int s_chunk = (lane % S_CPR) ^ WgTraits<T>::template swz<SROWB>(S_RPI * wave + s_lrow);
  const int g_chunk = (lane % G_CPR) ^ WgTraits<T>::template swz<GROWB>(G_RPI * wave + g_lrow);
  const int sc = r0 + s_chunk * VEC;
  const bool sc_ok = sc < p.Csp;
  const int col = nb0 + g_chunk * VEC;
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
  const char* zero = reinterpret_cast<const char*>(g_wg_zero_page);

  const uint32_t smem_base = __builtin_amdgcn_readfirstlane(lds_addr(smem));
  auto issue_stage = [&](int stage, int s) {
    const uint32_t st = smem_base + stage * STAGE_BYTES;     // wave-uniform LDS byte addresses
    const uint32_t gt = st + S_BYTES;
#pragma unroll
    for (int i = 0; i < S_NT; ++i) {
      const int g = wave + NW * i;
      const long long m = mbeg + (long long)s * KP + g * S_RPI + s_lrow;
      const char* ssrc = (m < mend && sc_ok) ? reinterpret_cast<const char*>(Sg + (size_t)m * p.Csp + sc) : zero;
      dma16_to_lds(ssrc, st + g * 1024);
    }
#pragma unroll
    for (int i = 0; i < G_NT; ++i) {
      const int g = wave + NW * i;
      const long long m = mbeg + (long long)s * KP + g * G_RPI + g_lrow;
      const char* gsrc = zero;
      if (m < mend && col_ok) {
        uint32_t q = (uint32_t)m, qw, qh, qd;
        fdivmod(q, p.fQw, q, qw);
        fdivmod(q, p.fQh, q, qh);
        fdivmod(q, p.fQd, q, qd);
        const int gd = (int)qd * p.sd + off_d, gh = (int)qh * p.sh + off_h, gw = (int)qw * p.sw + off_w;
        if ((unsigned)gd < (unsigned)p.Gd && (unsigned)gh < (unsigned)p.Gh && (unsigned)gw < (unsigned)p.Gw) {
          const size_t pix = ((size_t)((int)q * p.Gd + gd) * p.Gh + gh) * p.Gw + gw;
          gsrc = reinterpret_cast<const char*>(Gg + pix * p.Cgp + gch);
        }
      }
      dma16_to_lds(gsrc, gt + g * 1024);
    }
  };

  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  if (nsteps > 0) {
#pragma unroll
    for (int s = 0; s < STAGES - 1; ++s) issue_stage(s, s);     // steps past the end read the zero page
    int stage = 0;
    for (int s = 0; s < nsteps; ++s) {
      {
        constexpr int N = (S_NT + G_NT) * (STAGES - 2);
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");     // hand-written (asm): see tools/isa_audit.py
      }
      // WAR: the DMAs issued below overwrite the stage read in iteration s-1; `s_barrier` waits for no counter, so every
      // LDS read of that iteration must have RETURNED before this wave arrives here.  The compiler's own lgkmcnt waits
      // in front of the consuming MFMAs already guarantee it in every instantiation (tools/isa_audit.py); the explicit
      // wait makes it independent of how the MFMAs are scheduled.
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      int nstage = stage + STAGES - 1;
      if (nstage >= STAGES) nstage -= STAGES;
      issue_stage(nstage, s + STAGES - 1);
      const char* st = smem + stage * STAGE_BYTES;
      WgMma<T>::template step<SROWB, GROWB>(st, st + S_BYTES, wr0, wc0, lane, acc);
      if (++stage == STAGES) stage = 0;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }

  // slab store: ws[z][r][col], D[row = r (lane>>4)*4+reg][col = lane&15]
  float* slab = p.ws + (size_t)split * p.Cs * p.ncols;
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

// dw[a][b][t] = beta*dw + sum_z ws[z][a][t*Cbp + b].  A thread owns 4 consecutive slab columns (one float4 per slab,
// 1 KiB per wave-load; Cbp is a multiple of 8, so the four share a tap); 64 quads x 4 split-lanes per workgroup: every
// split-lane folds a quarter of the slabs with 4 loads in flight, LDS sums the lanes in a fixed order (bitwise
// reproducible), and the sums go to the torch layout.
// Workgroups past `wblocks` (bias_rep != null) fold the layer's BIAS gradient instead: the column sums of the output
// gradient that the consuming BatchNorm's apply pass left in replica rows bias_rep[VFD_STATS_REPLICAS][Cbp] are added to
// the parameter's gradient db[Cb] (vfd_bn_backward_apply_sums; a separate fold launch would cost as much as it does).
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* __restrict__ ws, float* __restrict__ dw, int A, int B,
                                                           int T, int Bp, int nsplit, float beta, long long wblocks,
                                                           const void* __restrict__ bias_rep, float* __restrict__ db, int Cb, int rep_stride,
                                                           int rep_f64) {
  if ((long long)blockIdx.x >= wblocks) {
    const int c = (int)(blockIdx.x - wblocks) * 256 + threadIdx.x;
    if (c < Cb) {
      if (rep_f64) {      // rows of a conv epilogue's statistics buffer (doubles, conv_epilogue.hpp)
        double v = 0.0;
#pragma unroll
        for (int r = 0; r < VFD_STATS_REPLICAS; ++r) v += reinterpret_cast<const double*>(bias_rep)[(size_t)r * rep_stride + c];
        db[c] += (float)v;
      } else {
        float v = 0.f;
#pragma unroll
        for (int r = 0; r < VFD_STATS_REPLICAS; ++r) v += reinterpret_cast<const float*>(bias_rep)[(size_t)r * rep_stride + c];
        db[c] += v;
      }
    }
    return;
  }
  const int ncols = T * Bp;
  const int qpr = ncols >> 2;                      // column quads per filter row
  const long long nquads = (long long)A * qpr;
  const int el = threadIdx.x & 63, zl = threadIdx.x >> 6;
  const long long quad = (long long)blockIdx.x * 64 + el;
  const size_t slab = (size_t)A * ncols;
  float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
  int a = 0, col = 0;
  if (quad < nquads) {
    a = (int)(quad / qpr);
    col = (int)(quad - (long long)a * qpr) << 2;
    const float* src = ws + (size_t)a * ncols + col;
    int z = zl;
    for (; z + 12 < nsplit; z += 16) {
      const float4 v0 = *reinterpret_cast<const float4*>(src + (size_t)z * slab);
      const float4 v1 = *reinterpret_cast<const float4*>(src + (size_t)(z + 4) * slab);
      const float4 v2 = *reinterpret_cast<const float4*>(src + (size_t)(z + 8) * slab);
      const float4 v3 = *reinterpret_cast<const float4*>(src + (size_t)(z + 12) * slab);
      s.x += v0.x; s.y += v0.y; s.z += v0.z; s.w += v0.w;
      s.x += v1.x; s.y += v1.y; s.z += v1.z; s.w += v1.w;
      s.x += v2.x; s.y += v2.y; s.z += v2.z; s.w += v2.w;
      s.x += v3.x; s.y += v3.y; s.z += v3.z; s.w += v3.w;
    }
    for (; z < nsplit; z += 4) {
      const float4 v = *reinterpret_cast<const float4*>(src + (size_t)z * slab);
      s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
  }
  __shared__ float4 sh[4][64];
  sh[zl][el] = s;
  __syncthreads();
  if (zl == 0 && quad < nquads) {
    const float4 p0 = sh[0][el], p1 = sh[1][el], p2 = sh[2][el], p3 = sh[3][el];
    const float r[4] = {((p0.x + p1.x) + p2.x) + p3.x, ((p0.y + p1.y) + p2.y) + p3.y, ((p0.z + p1.z) + p2.z) + p3.z,
                        ((p0.w + p1.w) + p2.w) + p3.w};
    const int t = col / Bp, b0 = col - t * Bp;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      if (b0 + k < B) {
        float* dst = dw + ((size_t)a * B + b0 + k) * T + t;
        *dst = (beta != 0.f) ? beta * (*dst) + r[k] : r[k];
      }
    }
  }
}

struct WgGeom {
  bool halo;      // conv_wgrad_halo.hip takes this layer (nsplit / bytes are then its slab geometry)
  WgP p;
  int A, B, T, nsplit;
  int tr, tc;     // tile: S channels x flattened columns
  size_t bytes;
};

int make_geom(const vfd_conv_desc* d, WgGeom& g) {
  int rc = vfd_conv_check_desc(d);
  if (rc != VFD_OK) return rc;
  g.halo = false;
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
  p.fQw = make_fastdiv((uint32_t)p.Qw); p.fQh = make_fastdiv((uint32_t)p.Qh); p.fQd = make_fastdiv((uint32_t)p.Qd);
  VFD_REQUIRE(p.M < 0x7fffffffLL, "wgrad: pixel count %lld exceeds 2^31", p.M);
  const int KP = d->dtype == VFD_BF16 ? 32 : 16;
  static const int force_tile = getenv("VFD_WGRAD_TILE") ? atoi(getenv("VFD_WGRAD_TILE")) : 0;   // tuning: 1 = 128x128, 2 = 256x128, 3 = 128x256
  // 4 = 64x256 (4 waves): S with <= 64 channels would leave half of a 128-row tile empty; 5 = 256x256 (16 waves, bf16)
  int shape = force_tile ? force_tile : (p.Cs >= 256 ? ((d->dtype == VFD_BF16 && p.ncols >= 256) ? 5 : 2) : (p.ncols >= 256 ? (p.Cs > 64 ? 3 : 4) : 1));
  if (shape == 5 && d->dtype != VFD_BF16) shape = 2;
  const bool wide = shape == 2 || shape == 3;
  g.tr = (shape == 2 || shape == 5) ? 256 : (shape == 4 ? 64 : 128);
  g.tc = (shape == 3 || shape == 4 || shape == 5) ? 256 : 128;
  const long long tiles = (long long)((p.Cs + g.tr - 1) / g.tr) * ((p.ncols + g.tc - 1) / g.tc);
  long long nsplit = (shape == 5 ? 256 : (wide ? 512 : 768)) / tiles;   // one round of 1 (16 waves), 2 (8 waves) or 3 (4 waves) resident workgroups per CU
  const long long maxsplit = (p.M + 4 * KP - 1) / (4 * KP);
  if (nsplit > maxsplit) nsplit = maxsplit;
  if (nsplit > 1024) nsplit = 1024;
  if (nsplit < 1) nsplit = 1;
  const size_t slab = (size_t)p.Cs * p.ncols * sizeof(float);
  while (nsplit > 1 && slab * (size_t)nsplit > ((size_t)512 << 20)) nsplit /= 2;
  long long chunk = (p.M + nsplit - 1) / nsplit;
  chunk = (chunk + KP - 1) / KP * KP;
  nsplit = (p.M + chunk - 1) / chunk;
  p.chunkM = chunk;
  g.nsplit = (int)nsplit;
  g.bytes = slab * (size_t)nsplit;
  {
    int hs = 0;
    size_t hb = 0;
    if (vfd_wgrad_halo_geom(d, &hs, &hb, nullptr) > 0) {
      g.halo = true;
      g.nsplit = hs;
      g.bytes = hb;
    }
  }
  return VFD_OK;
}

}  // namespace

extern "C" int vfd_wgrad_kernel_name(const vfd_conv_desc* d, char* buf, size_t n) {
  VFD_REQUIRE(buf != nullptr && n > 0, "wgrad_kernel_name: bad arguments");
  WgGeom g;
  int rc = make_geom(d, g);
  if (rc != VFD_OK) return rc;
  const char* t = d->dtype == VFD_BF16 ? "bf16" : "f32";
  if (g.halo) snprintf(buf, n, "conv_wgrad_halo<%s>", t);
  else snprintf(buf, n, "conv_wgrad<%s,%dx%d>", t, g.tr, g.tc);
  return VFD_OK;
}

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
  if (g.halo) {
    const int h = vfd_wgrad_halo_launch(d, x, dy, ws, as_stream(stream));
    if (h < 0) { vfd_set_error("conv_wgrad_halo: launch failed"); return VFD_ELAUNCH; }
    if (h > 0) return VFD_OK;
    vfd_set_error("conv_wgrad_halo: geometry changed between the workspace query and the launch");
    return VFD_EINVAL;
  }
  g.p.S = d->transposed ? x : dy;
  g.p.G = d->transposed ? dy : x;
  g.p.ws = reinterpret_cast<float*>(ws);
  static const bool no_xcd = getenv("VFD_NO_XCD_ORDER") != nullptr;
  g.p.ntx = (g.p.ncols + g.tc - 1) / g.tc;
  g.p.nty = (g.p.Cs + g.tr - 1) / g.tr;
  g.p.nsplit = g.nsplit;
  g.p.xcd_order = (!no_xcd && g.nsplit >= 8) ? 1 : 0;
  const long long nwg = (long long)g.p.ntx * g.p.nty * (g.p.xcd_order ? (g.nsplit + 7) / 8 * 8 : g.nsplit);
  VFD_REQUIRE(nwg < 0x7fffffffLL, "wgrad: grid too large");
  dim3 grid((unsigned)nwg, 1, 1);
  hipStream_t st = as_stream(stream);
  if (d->dtype == VFD_BF16) {
    // (a 4-deep ring on the 256 x 256 tile measured within 1 % of the 3-deep one: round 3, same box)
    if (g.tr == 256 && g.tc == 256) hipLaunchKernelGGL((conv_wgrad_kernel<bf16_t, 3, 4, 4>), grid, dim3(1024), 0, st, g.p);
    else if (g.tr == 256) hipLaunchKernelGGL((conv_wgrad_kernel<bf16_t, 3, 4, 2>), grid, dim3(512), 0, st, g.p);
    else if (g.tr == 64) hipLaunchKernelGGL((conv_wgrad_kernel<bf16_t, 3, 1, 4>), grid, dim3(256), 0, st, g.p);
    else if (g.tc == 256) hipLaunchKernelGGL((conv_wgrad_kernel<bf16_t, 3, 2, 4>), grid, dim3(512), 0, st, g.p);
    else hipLaunchKernelGGL((conv_wgrad_kernel<bf16_t, 3, 2, 2>), grid, dim3(256), 0, st, g.p);
  } else {
    if (g.tr == 256) hipLaunchKernelGGL((conv_wgrad_kernel<float, 3, 4, 2>), grid, dim3(512), 0, st, g.p);
    else if (g.tr == 64) hipLaunchKernelGGL((conv_wgrad_kernel<float, 3, 1, 4>), grid, dim3(256), 0, st, g.p);
    else if (g.tc == 256) hipLaunchKernelGGL((conv_wgrad_kernel<float, 3, 2, 4>), grid, dim3(512), 0, st, g.p);
    else hipLaunchKernelGGL((conv_wgrad_kernel<float, 3, 2, 2>), grid, dim3(256), 0, st, g.p);
  }
  VFD_CHECK_LAUNCH("conv_wgrad");
  return VFD_OK;
}

// The fold for filters with >= 64 padded gather channels: one workgroup per (filter row a, 64-channel chunk).  The slabs hold
// dWp[a][t][c] (c fastest) and torch's layout is dw[a][c][t] (t fastest): wgrad_reduce_kernel above reads runs of c and writes
// each value T floats apart — scattered 4-byte stores, measured at 0.9 TB/s on ganomaly's 512 x 256 x 4 x 4 filter (82 us for
// 67 MB of slabs).  Here the [T][64] block is summed slab by slab in 16-byte loads, turned in LDS and written as ONE
// contiguous run of 64 * T floats.  Sum order: slabs 0, 1, 2, ... per element (fixed; the kernel above folds four interleaved
// groups — last-bit differences between the two, both deterministic).
constexpr int RED_CB = 64;
__global__ __launch_bounds__(256) void wgrad_reduce_turn_kernel(const float* __restrict__ ws, float* __restrict__ dw, int A, int B,
                                                                 int T, int Bp, int nsplit, float beta, long long wblocks,
                                                                 const void* __restrict__ bias_rep, float* __restrict__ db, int Cb,
                                                                 int rep_stride, int rep_f64) {
  if ((long long)blockIdx.x >= wblocks) {
    const int c = (int)(blockIdx.x - wblocks) * 256 + threadIdx.x;
    if (c < Cb) {
      if (rep_f64) {
        double v = 0.0;
#pragma unroll
        for (int r = 0; r < VFD_STATS_REPLICAS; ++r) v += reinterpret_cast<const double*>(bias_rep)[(size_t)r * rep_stride + c];
        db[c] += (float)v;
      } else {
        float v = 0.f;
#pragma unroll
        for (int r = 0; r < VFD_STATS_REPLICAS; ++r) v += reinterpret_cast<const float*>(bias_rep)[(size_t)r * rep_stride + c];
        db[c] += v;
      }
    }
    return;
  }
  extern __shared__ float turn[];      // [T][RED_CB + 1]
  const int nchunk = Bp / RED_CB;
  const int a = (int)(blockIdx.x / nchunk), cb = (int)(blockIdx.x - (long long)a * nchunk);
  const int ncols = T * Bp;
  const size_t slab = (size_t)A * ncols;
  const float* src = ws + (size_t)a * ncols + cb * RED_CB;
  const int nq = T * (RED_CB / 4);
  for (int idx = threadIdx.x; idx < nq; idx += 256) {
    const int t = idx / (RED_CB / 4), q = idx - t * (RED_CB / 4);
    const float* p = src + (size_t)t * Bp + q * 4;
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    int z = 0;
    for (; z + 8 <= nsplit; z += 8) {
      float4 v[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) v[k] = *reinterpret_cast<const float4*>(p + (size_t)(z + k) * slab);
#pragma unroll
      for (int k = 0; k < 8; ++k) { s.x += v[k].x; s.y += v[k].y; s.z += v[k].z; s.w += v[k].w; }
    }
    for (; z < nsplit; ++z) {
      const float4 v = *reinterpret_cast<const float4*>(p + (size_t)z * slab);
      s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
    float* o = turn + t * (RED_CB + 1) + q * 4;
    o[0] = s.x; o[1] = s.y; o[2] = s.z; o[3] = s.w;
  }
  __syncthreads();
  const int nvalid = min(RED_CB, B - cb * RED_CB);
  if (nvalid <= 0) return;
  float* dst = dw + ((size_t)a * B + (size_t)cb * RED_CB) * T;
  const int nout = nvalid * T;
  for (int o = threadIdx.x; o < nout; o += 256) {
    const int c = o / T, t = o - c * T;
    const float v = turn[t * (RED_CB + 1) + c];
    dst[o] = (beta != 0.f) ? beta * dst[o] + v : v;
  }
}

static int wgrad_reduce_launch(const vfd_conv_desc* d, const void* ws, float* dw, float beta, const void* bias_rep, float* db,
                               int rep_stride, int rep_f64, void* stream) {
  WgGeom g;
  int rc = make_geom(d, g);
  if (rc != VFD_OK) return rc;
  VFD_REQUIRE(ws && dw, "wgrad_reduce: null pointer");
  const long long nquads = (long long)g.A * ((g.T * g.p.Cgp) >> 2);
  const long long blocks = (nquads + 63) / 64;
  const int extra = bias_rep != nullptr ? (d->Cout + 255) / 256 : 0;
  const long long tblocks = (long long)g.A * (g.p.Cgp / RED_CB);
  if (g.p.Cgp % RED_CB == 0 && tblocks >= 256 && g.T <= 128) {      // (fewer workgroups than CUs: the kernel below splits the slabs four ways)
    hipLaunchKernelGGL(wgrad_reduce_turn_kernel, dim3((unsigned)(tblocks + extra)), dim3(256), (size_t)g.T * (RED_CB + 1) * sizeof(float),
                       as_stream(stream), reinterpret_cast<const float*>(ws), dw, g.A, g.B, g.T, g.p.Cgp, g.nsplit, beta, tblocks, bias_rep,
                       db, d->Cout, rep_stride, rep_f64);
    VFD_CHECK_LAUNCH("wgrad_reduce_turn");
    return VFD_OK;
  }
  hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)(blocks + extra)), dim3(256), 0, as_stream(stream),
                     reinterpret_cast<const float*>(ws), dw, g.A, g.B, g.T, g.p.Cgp, g.nsplit, beta, blocks, bias_rep, db, d->Cout,
                     rep_stride, rep_f64);
  VFD_CHECK_LAUNCH("wgrad_reduce");
  return VFD_OK;
}

extern "C" int vfd_wgrad_reduce(const vfd_conv_desc* d, const void* ws, float* dw, float beta, void* stream) {
  return wgrad_reduce_launch(d, ws, dw, beta, nullptr, nullptr, 0, 0, stream);
}

extern "C" int vfd_wgrad_reduce_bias(const vfd_conv_desc* d, const void* ws, float* dw, float beta, const void* bias_rep,
                                     int rep_stride, int rep_f64, float* db, void* stream) {
  VFD_REQUIRE(d && bias_rep && db, "wgrad_reduce_bias: null pointer");
  VFD_REQUIRE(rep_stride >= cpad(d->Cout), "wgrad_reduce_bias: replica rows of %d floats are shorter than CPAD(Cout)", rep_stride);
  return wgrad_reduce_launch(d, ws, dw, beta, bias_rep, db, rep_stride, rep_f64 != 0, stream);
}
