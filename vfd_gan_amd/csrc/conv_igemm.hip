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
#include "conv_epilogue.hpp"
#include <stdlib.h>
#include <type_traits>

namespace {

constexpr int DIM_TAB = 8;   // per-dimension output classes with a host-built description (stride <= 8; else built on the device)

struct ConvP {
  const void* x;
  const void* w;
  void* y;
  const float* bias;
  double* stats;  // [VFD_STATS_REPLICAS][2][Cop] doubles or null (conv_epilogue.hpp)
  int N, Di, Hi, Wi, Cip;
  int Do, Ho, Wo, Cop, Cout;
  int kd, kh, kw, sd, sh, sw, pd, ph, pw;
  int transposed;
  int Kw;  // packed filter row length = kd*kh*kw*Cip
  int act;
  float slope;
  int ksplit;   // >1: K range split over blockIdx.z, raw f32 partial tiles go to ws[ksplit][M][Cop]
  float* ws;
  int variant;  // tuning: pipeline variant override (0 = default), env VFD_IGEMM_VARIANT
  int ny, ncls, mbp, xcd_order;   // channel tiles, output classes, pixel tiles rounded up to 8, workgroup order (see kernel)
  int tile_sub;          // FLEX kernels: 16-pixel sub-tiles per pixel tile (<= WAVES_P * NJ), see conv_igemm_kernel
  int dim_tab;           // 1: dims[][] below is valid
  DimClass dims[3][DIM_TAB];   // [d,h,w][output class r]: built on the host (make_dim_host), so that a workgroup's setup
                               // is scalar loads instead of ~15 integer divisions (one of them 64-bit) per dimension
  MulP mul;     // mul.src non-null: gradient hand-over in the epilogue (common.hpp)
  const float* qscale_x;   // fp8 operands: device scalars the tensors were multiplied by when quantised (x_q = x * scale);
  const float* qscale_w;   // the epilogue multiplies the accumulator by 1 / (scale_x * scale_w)
};


// device-side construction of the magic number (a handful of scalar instructions per workgroup, classes differ per block)
__device__ __forceinline__ FastDiv dev_fastdiv(uint32_t d) {
  FastDiv f;
  if (d == 0) d = 1;
  const uint32_t l = (d > 1) ? 32 - __builtin_clz(d - 1) : 0;   // ceil(log2 d)
  const uint32_t p = 31 + l;
  f.magic = (uint32_t)(((1ull << p) + d - 1) / d);
  f.shift = p;
  f.d = d;
  return f;
}

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
  d.fq = dev_fastdiv((uint32_t)d.Q);
  return d;
}

// LDS tile: rows of 64*KSUB bytes (4*KSUB chunks of 16 B); the chunk index is XOR-swizzled by the row so that the
// ds_read_b128 fragment reads (16 rows x 4 chunks per wave-instruction) are bank-conflict free
// (checked against the gfx950 lane-group / 64-bank model of MI355X_MICROARCH.md for both row lengths).
template <int KSUB>
__device__ __forceinline__ int row_swizzle(int row) {
  if constexpr (KSUB == 1) {
    const int g = (row >> 2) & 3;
    return (((g ^ (g >> 1)) & 1) << 1) | (g >> 1);
  } else {
    return (row >> 1) & 7;
  }
}
template <int KSUB>
__device__ __forceinline__ int lds_off(int row, int chunk) {
  return row * (64 * KSUB) + ((chunk ^ row_swizzle<KSUB>(row)) << 4);
}

template <typename T> struct Mma;
template <> struct Mma<bf16_t> {
  // one 64-byte K sub-step = 32 bf16 = one v_mfma_f32_16x16x32_bf16 per 16x16 tile
  // `last` (wave-uniform; FLEX tiles): whether the wave's last pixel sub-tile (j = NJ - 1) is in use.  Only that sub-tile's
  // fragment read and MFMAs are predicated: ONE code path, no second copy of the loop body (the 16-wave tile sits at its
  // 128-register cap: a duplicated body spilled 153 registers, tools/isa_audit.py).
  template <int NI, int NJ, int KSUB, bool FLEX = false>
  __device__ static __forceinline__ void step(const char* wt, const char* pt, int wrow0, int prow0, int lane,
                                              f32x4 (&acc)[NI][NJ], bool last = true) {
    const int r = lane & 15, ch = lane >> 4;
    constexpr int NJS = FLEX ? NJ - 1 : NJ;      // sub-tiles that are always in use
#pragma unroll
    for (int ks = 0; ks < KSUB; ++ks) {
      bf16x8 a[NI], b[NJS];
#pragma unroll
      for (int i = 0; i < NI; ++i) a[i] = *reinterpret_cast<const bf16x8*>(wt + lds_off<KSUB>(wrow0 + i * 16 + r, ch + 4 * ks));
#pragma unroll
      for (int j = 0; j < NJS; ++j) b[j] = *reinterpret_cast<const bf16x8*>(pt + lds_off<KSUB>(prow0 + j * 16 + r, ch + 4 * ks));
#pragma unroll
      for (int i = 0; i < NI; ++i)
#pragma unroll
        for (int j = 0; j < NJS; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
      if constexpr (FLEX) {
        if (last) {
          const bf16x8 bl = *reinterpret_cast<const bf16x8*>(pt + lds_off<KSUB>(prow0 + (NJ - 1) * 16 + r, ch + 4 * ks));
#pragma unroll
          for (int i = 0; i < NI; ++i) acc[i][NJ - 1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], bl, acc[i][NJ - 1], 0, 0, 0);
        }
      }
    }
  }
};
template <> struct Mma<fp8_t> {
  // one 128-byte row = 128 e4m3 = ONE v_mfma_f32_16x16x128_f8f6f4 per 16x16 tile (unscaled form: both block scales are the
  // constant 0, which the compiler folds away): twice the cycles of the bf16 16x16x32 at four times the K, i.e. twice the
  // FLOPs per clock, for the same LDS / DMA bytes per row as bf16.  Lane l holds K bytes [32 (l>>4), +32) of row l & 15
  // (two 16-byte chunks: ck_tile's kABKPerLane = 32 / kABKLane = 4 map; checked with exact integer data, tests/test_fp8.py).
  template <int NI, int NJ, int KSUB>
  __device__ static __forceinline__ void step(const char* wt, const char* pt, int wrow0, int prow0, int lane,
                                              f32x4 (&acc)[NI][NJ]) {
    static_assert(KSUB == 2, "fp8 tiles use 128-byte rows");
    typedef __attribute__((ext_vector_type(8))) int i32x8;
    const int r = lane & 15, q = lane >> 4;
    // 8 registers per fragment: with all NI + NJ fragments resident the 16-wave tile (64 accumulator registers, 128 in
    // all) spills; the A fragments stay, the B fragments pass through one register set (the 4 queued MFMAs of a B fragment
    // run 256 cycles: the next fragment's LDS read hides behind them)
    i32x8 a[NI], b;
    auto ldfrag = [&](const char* base, int row) __attribute__((always_inline)) {
      const uint4 lo = *reinterpret_cast<const uint4*>(base + lds_off<KSUB>(row, 2 * q));
      const uint4 hi = *reinterpret_cast<const uint4*>(base + lds_off<KSUB>(row, 2 * q + 1));
      return i32x8{(int)lo.x, (int)lo.y, (int)lo.z, (int)lo.w, (int)hi.x, (int)hi.y, (int)hi.z, (int)hi.w};
    };
#pragma unroll
    for (int i = 0; i < NI; ++i) a[i] = ldfrag(wt, wrow0 + i * 16 + r);
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      b = ldfrag(pt, prow0 + j * 16 + r);
#pragma unroll
      for (int i = 0; i < NI; ++i) acc[i][j] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a[i], b, acc[i][j], 0, 0, 0, 0, 0, 0);
    }
  }
};
template <> struct Mma<float> {
  // one 64-byte K sub-step = 16 f32 = four v_mfma_f32_16x16x4_f32 per 16x16 tile (exact f32 FMA chain)
  template <int NI, int NJ, int KSUB>
  __device__ static __forceinline__ void step(const char* wt, const char* pt, int wrow0, int prow0, int lane,
                                              f32x4 (&acc)[NI][NJ]) {
    const int r = lane & 15, kq = lane >> 4;
#pragma unroll
    for (int kk = 0; kk < 4 * KSUB; ++kk) {
      float a[NI], b[NJ];
#pragma unroll
      for (int i = 0; i < NI; ++i) a[i] = *reinterpret_cast<const float*>(wt + lds_off<KSUB>(wrow0 + i * 16 + r, kk) + kq * 4);
#pragma unroll
      for (int j = 0; j < NJ; ++j) b[j] = *reinterpret_cast<const float*>(pt + lds_off<KSUB>(prow0 + j * 16 + r, kk) + kq * 4);
#pragma unroll
      for (int i = 0; i < NI; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i], b[j], acc[i][j], 0, 0, 0);
    }
  }
};

// 64 zero bytes: the source of every out-of-bounds / padded 16-byte chunk of an LDS-DMA load (the DMA has no
// per-lane predicate that zero-fills, but its SOURCE address is per lane).
__device__ uint4 g_zero_page[4];

// Staging: global -> LDS directly (global_load_lds_dwordx4, 1 KiB = 16 tile rows per wave-instruction), STAGES-deep
// ring, ONE raw s_barrier per K-step, counted vmcnt so that STAGES-2 future steps stay in flight across the barrier.
// The LDS image is lane-linear per wave-instruction (row = lane/4, 16-byte slot = lane%4), so the bank swizzle of
// lds_off() is applied to the SOURCE: slot s of row r is fed logical chunk s ^ sw(r).
//
// FLEX (the 16-wave 256c x 256p tile): the pixel extent of a tile is a RUN-TIME number of 16-pixel sub-tiles,
// p.tile_sub in [WAVES_P * (NJ - 1), WAVES_P * NJ], dealt to the WAVES_P pixel-waves as evenly as possible (a wave column
// takes NJ or NJ - 1 of them; its unused sub-tile is never staged, never multiplied and is an out-of-range row to the
// epilogue).  The four waves that share a SIMD are the four pixel-wave columns of one channel-wave row, so the SIMD's MFMA
// work is tile_sub / (WAVES_P * NJ) of the full tile's.  Why: one workgroup per CU, and the pyramid layers have 196 / 392
// full tiles (25088 = 98 x 256 pixels per class) = 77 % of 256 CUs in the last round; 13 sub-tiles (208 pixels) give
// 242 / 484 tiles, i.e. the same number of rounds at 13/16 of the work per round (host: pick_tile_sub).
template <typename T, int WAVES_C, int WAVES_P, int NI, int NJ, int STAGES, int KSUB, bool BN = false, bool FLEX = false>
// launch bound = waves per SIMD the tile is DESIGNED for (a bound the allocator cannot meet only constrains it for nothing and
// draws "failed to meet occupancy target"): 4-wave tiles with >= 32 accumulator registers hold 2 workgroups' worth of waves
// per SIMD (their LDS allows 2-3 workgroups per CU, their 130-170 registers 2-3); the 16-channel tile reaches 3
__global__ __launch_bounds__(64 * WAVES_C * WAVES_P, (WAVES_C * WAVES_P == 4 ? (NI * NJ <= 4 ? 3 : 2) : ((STAGES * KSUB <= 3 && NI * NJ <= 16) ? 4 : 2))) void conv_igemm_kernel(const ConvP p) {
  constexpr int TILE_C = WAVES_C * NI * 16;
  constexpr int TILE_P = WAVES_P * NJ * 16;
  constexpr int VEC = Elem<T>::VEC;
  constexpr int ROWB = 64 * KSUB;              // bytes of K per tile row and stage
  constexpr int CPR = 4 * KSUB;                // 16-byte chunks per row
  constexpr int RPI = 64 / CPR;                // tile rows per 1-KiB DMA wave-instruction
  constexpr int BK = CPR * VEC;                // K elements per stage
  constexpr int NW = TILE_C / RPI;             // wave-instructions per stage for the filter tile
  constexpr int NP = TILE_P / RPI;             // ... for the pixel tile
  constexpr int NWAVES = WAVES_C * WAVES_P;
  constexpr int NWT = (NW + NWAVES - 1) / NWAVES;   // per wave
  constexpr int NPT = (NP + NWAVES - 1) / NWAVES;
  constexpr int STAGE_BYTES = (TILE_C + TILE_P) * ROWB;
  constexpr int DUMP_OFF = STAGES * STAGE_BYTES;   // 1 KiB sink for the padding DMAs of waves without a real row group
  static_assert(NWAVES == 4 || NWAVES == 8 || NWAVES == 16, "4, 8 or 16 waves per workgroup");
  static_assert(STAGES >= 2 && STAGES <= 4, "ring depth");
  static_assert(!FLEX || (STAGES == 2 && std::is_same<T, bf16_t>::value && RPI <= 16 && NJ >= 2),
                "FLEX skips the DMAs of unused sub-tiles: only where the ring wait is vmcnt(0) (2 stages)");

  constexpr int DUMP_BYTES = (NW % NWAVES == 0 && NP % NWAVES == 0) ? 0 : 1024;
  __shared__ __attribute__((aligned(16))) char smem[STAGES * STAGE_BYTES + DUMP_BYTES];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wave_c0 = (wave % WAVES_C) * (NI * 16);
  const int wave_p0 = (wave / WAVES_C) * (NJ * 16);

  // ---- work item of this workgroup ---------------------------------------------------------------------
  // Workgroups are dealt round-robin to the 8 XCDs (id % 8), each with its own L2.  All work items that gather the same
  // pixel tile - its channel tiles and, for a transposed convolution, its stride^3 output classes, which read the same
  // input pixels through different taps - are therefore given ids congruent mod 8 and adjacent in dispatch order, so
  // the re-reads hit that XCD's L2 instead of going back to HBM:  id = ((tile / 8) * group + j) * 8 + tile % 8.
  const int group = p.ncls * p.ny;
  int ptile, jitem;
  if (p.xcd_order) {
    const int slot = blockIdx.x >> 3;
    const int tq = slot / group;
    jitem = slot - tq * group;
    ptile = tq * 8 + (blockIdx.x & 7);
  } else {
    jitem = blockIdx.x / p.mbp;
    ptile = blockIdx.x - jitem * p.mbp;
  }
  int cls = jitem / p.ny;
  const int ytile = jitem - cls * p.ny;
  const int cls_id = cls;
  const int ksl = blockIdx.z;
  const int rw = p.transposed ? cls % p.sw : 0;
  if (p.transposed) cls /= p.sw;
  const int rh = p.transposed ? cls % p.sh : 0;
  if (p.transposed) cls /= p.sh;
  const int rd = p.transposed ? cls : 0;
  const DimClass dd = p.dim_tab ? p.dims[0][rd] : make_dim(p.transposed, rd, p.kd, p.sd, p.pd, p.Do);
  const DimClass dh = p.dim_tab ? p.dims[1][rh] : make_dim(p.transposed, rh, p.kh, p.sh, p.ph, p.Ho);
  const DimClass dw = p.dim_tab ? p.dims[2][rw] : make_dim(p.transposed, rw, p.kw, p.sw, p.pw, p.Wo);
  const long long Mcls = (long long)p.N * dd.Q * dh.Q * dw.Q;
  // FLEX: pixel-wave column wp owns sub-tiles [wp*fq + min(wp, frem), +fq + (wp < frem)) of the tile's tsub
  const int tsub = FLEX ? p.tile_sub : WAVES_P * NJ;
  const int fq = tsub / WAVES_P, frem = tsub - fq * WAVES_P;
  const long long m0 = (long long)ptile * (tsub * 16);
  if (m0 >= Mcls) return;  // uniform per workgroup
  // tile row (the LDS / accumulator row: wave column, sub-tile, pixel) -> pixel index inside the class, or -1
  auto row_m = [&](int r) -> long long {
    if constexpr (!FLEX) {
      return m0 + r;
    } else {
      const int wp = r / (NJ * 16), j = (r >> 4) % NJ;
      if (j >= fq + (wp < frem ? 1 : 0)) return -1;
      return m0 + ((wp * fq + min(wp, frem) + j) << 4) + (r & 15);
    }
  };
  const int nj_w = FLEX ? fq + ((wave / WAVES_C) < frem ? 1 : 0) : NJ;     // sub-tiles of this wave (wave-uniform)
  const int n0 = ytile * TILE_C;
  const int ntaps = dd.nk * dh.nk * dw.nk;
  const int Kcls = ntaps * p.Cip;
  const int nsteps_all = (Kcls + BK - 1) / BK;
  const int steps_per = (nsteps_all + p.ksplit - 1) / p.ksplit;
  const int s_begin = ksl * steps_per;
  const int nsteps = max(0, min(nsteps_all - s_begin, steps_per));

  // ---- per-thread DMA bookkeeping --------------------------------------------------------------------------
  // lane -> (row within a 16-row group, physical 16-byte slot); logical K chunk = slot ^ swizzle(row)
  const int lrow = lane / CPR;
  // row = RPI * g + lrow with g = wave + NWAVES i: the swizzle of that row depends on the lane and on wave & 1 only
  const int chunk = (lane % CPR) ^ row_swizzle<KSUB>(RPI * (wave & 1) + lrow);
  // position of this thread's chunk inside the flattened K axis: (tap, channel offset kc)
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
  const T* __restrict__ xg = reinterpret_cast<const T*>(p.x);
  const T* __restrict__ wg = reinterpret_cast<const T*>(p.w);

  // filter rows of this thread: group index gw = wave + 4*i (valid while < NW)
  // offsets are kept in 16-byte units in 32 bits (Cip, Kw and kc are multiples of the 16-byte granule; the host checks
  // that both tensors stay below 2^32 granules = 64 GiB): half the registers of 64-bit element offsets
  constexpr uint32_t NONE = 0xffffffffu;
  uint32_t wrow_off[NWT];       // granule offset of (row, tap 0, channel 0), or NONE
#pragma unroll
  for (int i = 0; i < NWT; ++i) {
    const int g = wave + NWAVES * i;
    const int co = n0 + g * RPI + lrow;
    wrow_off[i] = (g < NW && co < p.Cout) ? (uint32_t)(((long long)co * p.Kw) / VEC) : NONE;
  }
  // pixel rows of this thread
  int pn[NPT], pid[NPT], pih[NPT], piw[NPT];
  bool pgroup[NPT];       // FLEX: this wave-instruction's rows belong to a sub-tile in use (wave-uniform)
#pragma unroll
  for (int i = 0; i < NPT; ++i) {
    const int g = wave + NWAVES * i;
    const long long m = row_m(g * RPI + lrow);
    pgroup[i] = !FLEX || row_m(g * RPI) >= 0;
    if (g < NP && m >= 0 && m < Mcls) {
      uint32_t q = (uint32_t)m, qw, qh, qd;     // M < 2^31 (checked on the host): multiply-shift divisions
      fdivmod(q, dw.fq, q, qw);
      fdivmod(q, dh.fq, q, qh);
      fdivmod(q, dd.fq, q, qd);
      pn[i] = (int)q;
      pid[i] = (int)qd * dd.a + dd.c0;
      pih[i] = (int)qh * dh.a + dh.c0;
      piw[i] = (int)qw * dw.a + dw.c0;
    } else {
      pn[i] = -1; pid[i] = 0; pih[i] = 0; piw[i] = 0;
    }
  }
  // per-tap state (recomputed only when this thread's chunk moves to another tap)
  uint32_t wtap_off = 0;        // tap index * Cip inside a filter row, in granules
  uint32_t ppix_off[NPT];       // granule offset of the gathered input pixel, or NONE (padding / out of range)
  bool kvalid = false;
  auto enter_tap = [&]() {
    kvalid = (td < dd.nk) && ntaps > 0;
    const int tapidx = ((dd.k0 + td * dd.ks) * p.kh + (dh.k0 + th * dh.ks)) * p.kw + (dw.k0 + tw * dw.ks);
    wtap_off = (uint32_t)(tapidx * (p.Cip / VEC));
    const int od = td * dd.cs, oh = th * dh.cs, ow = tw * dw.cs;
#pragma unroll
    for (int i = 0; i < NPT; ++i) {
      const int id = pid[i] + od, ih = pih[i] + oh, iw = piw[i] + ow;
      const bool ok = kvalid && pn[i] >= 0 && (unsigned)id < (unsigned)p.Di && (unsigned)ih < (unsigned)p.Hi &&
                      (unsigned)iw < (unsigned)p.Wi;
      ppix_off[i] = ok ? (uint32_t)((((pn[i] * p.Di + id) * p.Hi + ih) * p.Wi + iw)) * (uint32_t)(p.Cip / VEC) : NONE;
    }
  };
  enter_tap();
  const char* zero = reinterpret_cast<const char*>(g_zero_page);
  const uint32_t smem_base = __builtin_amdgcn_readfirstlane(lds_addr(smem));
  auto issue_stage = [&](int stage) {
    const uint32_t wt = smem_base + stage * STAGE_BYTES;      // wave-uniform LDS byte addresses
    const uint32_t pt = wt + TILE_C * ROWB;
#pragma unroll
    for (int i = 0; i < NWT; ++i) {
      const int g = wave + NWAVES * i;   // wave-uniform
      const char* src = (kvalid && wrow_off[i] != NONE)
                            ? reinterpret_cast<const char*>(wg) + ((size_t)(wrow_off[i] + wtap_off) << 4) + kc * (int)sizeof(T) : zero;
      const uint32_t dst = (NW % NWAVES == 0 || g < NW) ? wt + g * 1024 : smem_base + DUMP_OFF;
      dma16_to_lds(src, dst);      // inline asm: see common.hpp
    }
#pragma unroll
    for (int i = 0; i < NPT; ++i) {
      const int g = wave + NWAVES * i;
      if (FLEX && !pgroup[i]) continue;      // rows of a sub-tile this tile does not use: nothing reads them
      const char* src = (ppix_off[i] != NONE) ? reinterpret_cast<const char*>(xg) + ((size_t)ppix_off[i] << 4) + kc * (int)sizeof(T) : zero;
      const uint32_t dst = (NP % NWAVES == 0 || g < NP) ? pt + g * 1024 : smem_base + DUMP_OFF;
      dma16_to_lds(src, dst);
    }
  };
  auto advance_k = [&]() {
    kc += BK;
    if (kc >= p.Cip) {
      do {
        kc -= p.Cip;
        if (++tw >= dw.nk) {
          tw = 0;
          if (++th >= dh.nk) { th = 0; ++td; }
        }
      } while (kc >= p.Cip);
      enter_tap();
    }
  };

  f32x4 acc[NI][NJ];
#pragma unroll
  for (int i = 0; i < NI; ++i)
#pragma unroll
    for (int j = 0; j < NJ; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  if (nsteps > 0) {
    // prologue: steps 0 .. STAGES-2 in flight (positions past the end of K read the zero page)
#pragma unroll
    for (int s = 0; s < STAGES - 1; ++s) {
      issue_stage(s);
      advance_k();
    }
    int stage = 0;
    for (int s = 0; s < nsteps; ++s) {
      // this wave's DMAs of step s have landed once at most (STAGES-2) younger steps' DMAs are outstanding
      {
        constexpr int N = (NWT + NPT) * (STAGES - 2);
        static_assert(N < 64, "vmcnt field");
        // hand-written wait (inline asm: tools/isa_audit.py tells it from a compiler-inserted one by the asm markers)
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
      }
      // every wave's step-s DMAs landed AND every wave finished reading the stage the next issue overwrites: the
      // barrier itself waits for no counter, so this wave's LDS reads of the previous step are retired explicitly
      // (conv_wgrad.hip has the same wait; tools/isa_audit.py checks the emitted stream)
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      int nstage = stage + STAGES - 1;
      if (nstage >= STAGES) nstage -= STAGES;
      issue_stage(nstage);          // step s + STAGES - 1 (zero page beyond the end: keeps the vmcnt accounting uniform)
      advance_k();
      const char* wt = smem + stage * STAGE_BYTES;
      const char* pt = wt + TILE_C * ROWB;
      if constexpr (FLEX) Mma<T>::template step<NI, NJ, KSUB, true>(wt, pt, wave_c0, wave_p0, lane, acc, nj_w == NJ);
      else Mma<T>::template step<NI, NJ, KSUB>(wt, pt, wave_c0, wave_p0, lane, acc);
      if (++stage == STAGES) stage = 0;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // drain the trailing zero-page DMAs before LDS is released
  }

  const int cq = (lane >> 4) * 4;
  if (!std::is_same<T, fp8_t>::value && p.ksplit > 1) {      // (fp8 tiles never split K: keeps `lane` from living across their main loop)
    // split-K: raw partial sums, [ksplit][M][Cop] float32; bias / activation / store happen in conv_splitk_finish
    float* slab = p.ws + (size_t)ksl * (size_t)Mcls * p.Cop;
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      const long long m = row_m(wave_p0 + j * 16 + (lane & 15));
#pragma unroll
      for (int i = 0; i < NI; ++i) {
        const int c = n0 + wave_c0 + i * 16 + cq;
        if (m >= 0 && m < Mcls && c < p.Cop)
          *reinterpret_cast<float4*>(slab + (size_t)m * p.Cop + c) = make_float4(acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]);
      }
    }
    return;
  }

  // ---- epilogue: bias, statistics, activation, channels-last store (conv_epilogue.hpp) ----------------------
  EpiP e;
  e.y = p.y; e.bias = p.bias; e.stats = p.stats; e.Cop = p.Cop; e.Cout = p.Cout; e.act = p.act; e.slope = p.slope;
  e.mul = p.mul;
  e.oscale = 1.f;
  if constexpr (std::is_same<T, fp8_t>::value) e.oscale = 1.f / (p.qscale_x[0] * p.qscale_w[0]);
  auto out_offset = [&](int r) -> long long {       // tile row -> element offset of the output pixel, or -1
    const long long m = row_m(r);
    if (m < 0 || m >= Mcls) return -1;
    uint32_t q = (uint32_t)m, qw, qh, qd;
    fdivmod(q, dw.fq, q, qw);
    fdivmod(q, dh.fq, q, qh);
    fdivmod(q, dd.fq, q, qd);
    const int n = (int)q;
    return ((((long long)(n * p.Do + (int)qd * dd.so + dd.r) * p.Ho + (int)qh * dh.so + dh.r) * p.Wo + (int)qw * dw.so + dw.r)) * p.Cop;
  };
  conv_epilogue<typename OutOf<T>::type, WAVES_C, WAVES_P, NI, NJ, STAGES * STAGE_BYTES + DUMP_BYTES, BN, std::is_same<T, fp8_t>::value>(smem, acc, e, n0, ptile + cls_id, out_offset,
                                                                                [&](int r) { const long long m = row_m(r); return m >= 0 && m < Mcls; });
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
  // regular convolutions, and transposed ones with unit stride (one output class, out pixel == m)
  if ((p.transposed && p.sd * p.sh * p.sw != 1) || p.stats != nullptr) return 1;
  const long long blocks = ((M + TILE_P - 1) / TILE_P) * ((p.Cout + TILE_C - 1) / TILE_C);
  const int nsteps = (p.Kw + bk - 1) / bk;
  if (blocks >= 128 || nsteps < 32) return 1;
  long long ks = 512 / blocks;
  if (ks > nsteps / 8) ks = nsteps / 8;
  if (ks > 32) ks = 32;      // (swept 8..256 on ganomaly's 7 x 7 -> 1 x 1 layers, round 3: 58 / 39 / 32.5 / 36 / 44 / 44 us at 8 / 16 / 32 / 64 / 128 / 256)
  return ks < 2 ? 1 : (int)ks;
}

// FLEX tiles (one workgroup per CU): the number of 16-pixel sub-tiles per pixel tile, in [lo, hi], that minimises
// rounds x (work per round): rounds = ceil(tiles / CUs) — workgroups of one launch are equal, so a launch takes whole rounds —
// and a tile costs its sub-tiles plus a fixed part (prologue, epilogue set-up: ~2 sub-tiles' worth).
int g_tile_sub_forced = -1;     // -1: not yet read from the environment (VFD_IGEMM_TILE_SUB); see vfd_conv_set_tile_sub
static int pick_tile_sub(long long maxM, int ny, int ncls, int lo, int hi) {
  if (g_tile_sub_forced < 0) g_tile_sub_forced = getenv("VFD_IGEMM_TILE_SUB") ? atoi(getenv("VFD_IGEMM_TILE_SUB")) : 0;
  const int forced = g_tile_sub_forced;
  if (forced >= lo && forced <= hi) return forced;
  static int cus = 0;
  if (cus == 0) {
    int dev = 0, n = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
    cus = n;
  }
  int best = hi;
  long long best_cost = -1;
  for (int n = hi; n >= lo; --n) {
    const long long tiles = ((maxM + 16LL * n - 1) / (16LL * n)) * ny * ncls;
    const long long cost = ((tiles + cus - 1) / cus) * (n + 2);
    if (best_cost < 0 || cost < best_cost) { best_cost = cost; best = n; }
  }
  return best;
}

template <typename T, int WAVES_C, int WAVES_P, int NI, int NJ, int STAGES, int KSUB, bool BN = false, bool FLEX = false>
int launch_cfg(const ConvP& p, long long maxM, int ncls, hipStream_t st, size_t ws_bytes, size_t* ws_query) {
  constexpr int TILE_C = WAVES_C * NI * 16;
  constexpr int TILE_P_FULL = WAVES_P * NJ * 16;
  const int tile_sub = FLEX ? pick_tile_sub(maxM, (p.Cout + TILE_C - 1) / TILE_C, ncls, WAVES_P * (NJ - 1), WAVES_P * NJ) : WAVES_P * NJ;
  const int TILE_P = tile_sub * 16;
  const long long mb = (maxM + TILE_P - 1) / TILE_P;
  if (mb <= 0) return VFD_OK;
  if (maxM >= 0x7fffffffLL) { vfd_set_error("conv: %lld output pixels per class exceed 2^31", maxM); return VFD_EINVAL; }
  {
    // the kernel addresses both operands in 32-bit counts of 16-byte granules (and input pixels in 31 bits)
    const long long in_px = (long long)p.N * p.Di * p.Hi * p.Wi;
    const long long x_gran = in_px * p.Cip / Elem<T>::VEC, w_gran = (long long)p.Cout * p.Kw / Elem<T>::VEC;
    if (in_px >= 0x7fffffffLL || x_gran >= 0xffffffffLL || w_gran >= 0xffffffffLL) {
      vfd_set_error("conv: operand too large for 32-bit granule offsets (%lld input pixels)", in_px);
      return VFD_EINVAL;
    }
  }
  ConvP q = p;
  q.ksplit = (p.mul.src != nullptr || std::is_same<T, fp8_t>::value) ? 1 : pick_ksplit<TILE_C, TILE_P_FULL>(p, maxM, 4 * KSUB * Elem<T>::VEC);
  q.tile_sub = tile_sub;
  if constexpr (FLEX) {
    // (few-pixel layers that split K keep the full tile: the slab layout is indexed by the full tile)
    if (q.ksplit > 1 && tile_sub != WAVES_P * NJ) return launch_cfg<T, WAVES_C, WAVES_P, NI, NJ, STAGES, KSUB, BN, false>(p, maxM, ncls, st, ws_bytes, ws_query);
  }
  const size_t need = q.ksplit > 1 ? (size_t)q.ksplit * (size_t)maxM * p.Cop * sizeof(float) : 0;
  if (ws_query != nullptr) { *ws_query = need; return VFD_OK; }
  if (q.ksplit > 1 && (p.ws == nullptr || ws_bytes < need)) q.ksplit = 1;   // no workspace: plain path
  static const bool no_xcd = getenv("VFD_NO_XCD_ORDER") != nullptr;
  q.ny = (p.Cout + TILE_C - 1) / TILE_C;
  q.ncls = ncls;
  q.xcd_order = (no_xcd || mb < 16) ? 0 : 1;     // few pixel tiles: the padding to a multiple of 8 would only add idle workgroups
  q.mbp = q.xcd_order ? (int)((mb + 7) / 8 * 8) : (int)mb;
  const long long nwg = (long long)q.mbp * q.ny * ncls;
  if (nwg >= 0x7fffffffLL || q.ksplit > 65535) { vfd_set_error("conv: grid too large"); return VFD_EINVAL; }
  dim3 grid((unsigned)nwg, 1, (unsigned)q.ksplit);
  hipLaunchKernelGGL((conv_igemm_kernel<T, WAVES_C, WAVES_P, NI, NJ, STAGES, KSUB, BN, FLEX>), grid, dim3(64 * WAVES_C * WAVES_P), 0, st, q);
  VFD_CHECK_LAUNCH("conv_igemm");
  if (q.ksplit > 1) {
    const long long total = maxM * (p.Cop >> 3);
    long long nb = (total + 255) / 256;
    if (nb > 4096) nb = 4096;
    hipLaunchKernelGGL(conv_splitk_finish_kernel<typename OutOf<T>::type>, dim3((unsigned)nb), dim3(256), 0, st, q.ws, reinterpret_cast<typename OutOf<T>::type*>(p.y), p.bias,
                       maxM, p.Cout, p.Cop, q.ksplit, p.act, p.slope);
    VFD_CHECK_LAUNCH("conv_splitk_finish");
  }
  return VFD_OK;
}

// e4m3 operands: 128-byte rows only (one K = 128 MFMA per row): the 16-wave 256 x 256 tile, and 128 x 128 (4 waves) below
// 129 channels
int launch_fp8(const ConvP& p, long long maxM, int ncls, hipStream_t st, size_t ws_bytes, size_t* ws_query) {
  if (p.mul.src != nullptr) { vfd_set_error("conv: no gradient hand-over with fp8 operands"); return VFD_EINVAL; }
  if (p.Cout > 128) return launch_cfg<fp8_t, 4, 4, 4, 4, 2, 2>(p, maxM, ncls, st, ws_bytes, ws_query);
  return launch_cfg<fp8_t, 2, 2, 4, 4, 2, 2>(p, maxM, ncls, st, ws_bytes, ws_query);
}

template <typename T>
int launch(const ConvP& p, long long maxM, int ncls, hipStream_t st, size_t ws_bytes, size_t* ws_query) {
  if constexpr (std::is_same<T, bf16_t>::value) {
    if (p.mul.bn_mean != nullptr) {      // BatchNorm hand-over: the default tile of each channel range, in its BN variant
      if (p.Cout > 128) return launch_cfg<T, 4, 4, 4, 4, 2, 2, true, true>(p, maxM, ncls, st, ws_bytes, ws_query);
      if (p.Cout > 64) return launch_cfg<T, 2, 4, 4, 4, 3, 1, true>(p, maxM, ncls, st, ws_bytes, ws_query);
      if (p.Cout > 32) return launch_cfg<T, 1, 4, 4, 4, 3, 1, true>(p, maxM, ncls, st, ws_bytes, ws_query);
      vfd_set_error("conv: the BatchNorm hand-over needs more than 32 output channels");
      return VFD_EINVAL;
    }
  } else {
    if (p.mul.bn_mean != nullptr) { vfd_set_error("conv: the BatchNorm hand-over is bf16 only"); return VFD_EINVAL; }
  }
  const int variant = p.variant;   // tuning override (env VFD_IGEMM_VARIANT); 0 = heuristic below
  if (p.Cout > 64) {
    // measured on the ganomaly pyramid (tools/layer_bench.py, bf16): 256c x 128p (8 waves) 750-825 TFLOP/s for
    // Cout >= 256; 128c x 256p (8 waves) 540-740 for Cout = 128; the 4-wave 128 x 128 tile 520-690.
    if (variant == 1) return launch_cfg<T, 2, 2, 4, 4, 3, 1>(p, maxM, ncls, st, ws_bytes, ws_query);   // 128c x 128p, 4 waves
    if (variant == 2) return launch_cfg<T, 2, 2, 4, 4, 2, 2>(p, maxM, ncls, st, ws_bytes, ws_query);   // ... 128-byte rows
    if (variant == 5 || ((variant == 0 || variant == 17) && p.Cout <= 128))
      return launch_cfg<T, 2, 4, 4, 4, 3, 1>(p, maxM, ncls, st, ws_bytes, ws_query);                   // 128c x 256p, 8 waves
    if (variant == 4 || variant == 17) return launch_cfg<T, 4, 2, 4, 4, 3, 1>(p, maxM, ncls, st, ws_bytes, ws_query);   // 256c x 128p, 8 waves
    // 256c x 256p, 16 waves (one workgroup per CU), 128-byte rows, 2 stages: the only tile whose DMA bytes per MFMA
    // cycle (31 B/clk/CU at full MFMA rate) fit under the 54 B/clk/CU the global->LDS path delivers with 128-byte rows
    if constexpr (std::is_same<T, bf16_t>::value) return launch_cfg<T, 4, 4, 4, 4, 2, 2, false, true>(p, maxM, ncls, st, ws_bytes, ws_query);
    else return launch_cfg<T, 4, 4, 4, 4, 2, 2>(p, maxM, ncls, st, ws_bytes, ws_query);
  }
  if (p.Cout > 32) return launch_cfg<T, 1, 4, 4, 4, 3, 1>(p, maxM, ncls, st, ws_bytes, ws_query);   //  64 ch x 256 px
  if (p.Cout > 16) return launch_cfg<T, 1, 4, 2, 4, 3, 1>(p, maxM, ncls, st, ws_bytes, ws_query);   //  32 ch x 256 px
  return launch_cfg<T, 1, 4, 1, 4, 3, 1>(p, maxM, ncls, st, ws_bytes, ws_query);                    //  16 ch x 256 px
}

}  // namespace

int vfd_conv_check_desc(const vfd_conv_desc* d) {
  VFD_REQUIRE(d != nullptr, "conv: null descriptor");
  VFD_REQUIRE(d->dtype == VFD_F32 || d->dtype == VFD_BF16 || d->dtype == VFD_FP8, "conv: bad dtype %d", d->dtype);
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

static int conv_dispatch(const vfd_conv_desc* d_in, const void* x, const void* packed, const float* bias, void* y, double* stats,
                         size_t stats_bytes, void* ws, size_t ws_bytes, size_t* ws_query, void* stream,
                         const MulP& mul = no_mul(), const float* qscale_x = nullptr, const float* qscale_w = nullptr) {
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
    VFD_REQUIRE(stats == nullptr || stats_bytes >= (size_t)VFD_STATS_REPLICAS * 2 * cpad(d->Cout) * sizeof(double),
                "conv: statistics buffer holds %zu bytes, needs VFD_STATS_REPLICAS*2*CPAD(Cout) doubles", stats_bytes);
    VFD_REQUIRE(((uintptr_t)stats & 7) == 0, "conv: the statistics buffer must be 8-byte aligned");
    VFD_REQUIRE((((uintptr_t)x | (uintptr_t)packed | (uintptr_t)y | (uintptr_t)ws) & 15) == 0, "conv: tensors must be 16-byte aligned");
  }
  const bool fp8 = d->dtype == VFD_FP8;      // e4m3 operands, bf16 output: conv_igemm's 128-byte-row tiles only
  if (fp8 && ws_query == nullptr) VFD_REQUIRE(qscale_x != nullptr && qscale_w != nullptr, "conv: fp8 operands need their scales (vfd_conv_forward_fp8)");
  if (!fp8) {
    // thin-channel pyramid ends have their own kernels (conv_small.hip); they need no workspace
    const int h = mul.src != nullptr ? 0 : vfd_conv_small_try(d, x, packed, bias, y, stats, ws_query != nullptr, as_stream(stream));
    if (h < 0) return VFD_ELAUNCH;
    if (h > 0) {
      if (ws_query != nullptr) *ws_query = 0;
      return VFD_OK;
    }
  }
  if (!fp8) {
    const int h = vfd_conv_halo_try(d, x, packed, bias, y, stats, mul, ws_query != nullptr, as_stream(stream));
    if (h < 0) return VFD_ELAUNCH;
    if (h > 0) {
      if (ws_query != nullptr) *ws_query = 0;
      return VFD_OK;
    }
  }
  ConvP p;
  p.x = x; p.w = packed; p.y = y; p.bias = bias; p.stats = stats;
  p.N = d->N; p.Di = d->Di; p.Hi = d->Hi; p.Wi = d->Wi; p.Cip = fp8 ? cpad16(d->Cin) : cpad(d->Cin);
  p.qscale_x = qscale_x; p.qscale_w = qscale_w;
  p.Do = d->Do; p.Ho = d->Ho; p.Wo = d->Wo; p.Cop = cpad(d->Cout); p.Cout = d->Cout;
  p.kd = d->kd; p.kh = d->kh; p.kw = d->kw; p.sd = d->sd; p.sh = d->sh; p.sw = d->sw;
  p.pd = d->pd; p.ph = d->ph; p.pw = d->pw;
  p.transposed = d->transposed;
  p.Kw = d->kd * d->kh * d->kw * p.Cip;
  p.act = d->act; p.slope = d->slope;
  p.ksplit = 1; p.ws = reinterpret_cast<float*>(ws);
  p.mul = mul;
  {
    const int kk[3] = {p.kd, p.kh, p.kw}, ss[3] = {p.sd, p.sh, p.sw}, pp[3] = {p.pd, p.ph, p.pw}, oo[3] = {p.Do, p.Ho, p.Wo};
    p.dim_tab = 1;
    for (int i = 0; i < 3; ++i) {
      const int nr = p.transposed ? ss[i] : 1;
      if (nr > DIM_TAB) { p.dim_tab = 0; break; }
      for (int r = 0; r < nr; ++r) p.dims[i][r] = make_dim_host(p.transposed, r, kk[i], ss[i], pp[i], oo[i]);
    }
  }
  {
    static const int v = getenv("VFD_IGEMM_VARIANT") ? atoi(getenv("VFD_IGEMM_VARIANT")) : 0;
    p.variant = v;
  }
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
  if (fp8) return launch_fp8(p, maxM, ncls, st, ws_bytes, ws_query);
  return d->dtype == VFD_BF16 ? launch<bf16_t>(p, maxM, ncls, st, ws_bytes, ws_query) : launch<float>(p, maxM, ncls, st, ws_bytes, ws_query);
}

extern "C" int vfd_conv_set_tile_sub(int sub) {
  const int prev = g_tile_sub_forced < 0 ? (getenv("VFD_IGEMM_TILE_SUB") ? atoi(getenv("VFD_IGEMM_TILE_SUB")) : 0) : g_tile_sub_forced;
  g_tile_sub_forced = sub < 0 ? 0 : sub;
  return prev;
}

extern "C" int vfd_conv_kernel_name(const vfd_conv_desc* d_in, int want_stats, char* buf, size_t n) {
  VFD_REQUIRE(d_in != nullptr && buf != nullptr && n > 0, "conv_kernel_name: bad arguments");
  vfd_conv_desc dn = *d_in;
  if (dn.transposed) {
    if (dn.Di == 1) dn.sd = dn.kd;
    if (dn.Hi == 1) dn.sh = dn.kh;
    if (dn.Wi == 1) dn.sw = dn.kw;
  }
  int rc = vfd_conv_check_desc(&dn);
  if (rc != VFD_OK) return rc;
  const char* t = dn.dtype == VFD_BF16 ? "bf16" : dn.dtype == VFD_FP8 ? "fp8" : "f32";
  if (dn.dtype == VFD_FP8) {
    snprintf(buf, n, "conv_igemm<fp8,%s>", dn.Cout > 128 ? "256c_x_256p" : "128c_x_128p");
    return VFD_OK;
  }
  double dummy_stats;
  if (vfd_conv_small_try(&dn, nullptr, nullptr, nullptr, nullptr, want_stats != 0 ? &dummy_stats : nullptr, true, nullptr) > 0) {
    const bool pointwise = dn.kh == 1 && dn.kw == 1;      // conv_small.hip: (kd,1,1) filters over 16..64 channels run on conv_cin8
    snprintf(buf, n, "%s<%s>", (cpad(dn.Cin) == 8 || pointwise) ? "conv_cin8" : "convt_thin", t);
    return VFD_OK;
  }
  const int halo = vfd_conv_halo_try(&dn, nullptr, nullptr, nullptr, nullptr, nullptr, no_mul(), true, nullptr);
  if (halo == 2) {
    snprintf(buf, n, "conv_halo_rows<%s,64c_x_256p>", t);
    return VFD_OK;
  }
  if (halo > 0) {
    snprintf(buf, n, "conv_halo<%s,%s>", t, dn.Cout > 32 ? "64c_x_256p" : (dn.Cout > 16 || (dn.Di == 1 && dn.Do == 1) || getenv("VFD_HALO_NO_16C") != nullptr) ? "32c_x_256p" : "16c_x_256p");
    return VFD_OK;
  }
  const int c = dn.Cout;   // launch<T>() below
  static const int forced = getenv("VFD_IGEMM_VARIANT") ? atoi(getenv("VFD_IGEMM_VARIANT")) : 0;
  const char* tile = c > 128 ? ((forced == 4 || forced == 17) ? "256c_x_128p" : "256c_x_256p") : c > 64 ? "128c_x_256p" : c > 32 ? "64c_x_256p" : c > 16 ? "32c_x_256p" : "16c_x_256p";
  snprintf(buf, n, "conv_igemm<%s,%s>", t, tile);
  return VFD_OK;
}

extern "C" int vfd_conv_workspace(const vfd_conv_desc* d, int want_stats, size_t* bytes) {
  VFD_REQUIRE(bytes != nullptr, "conv_workspace: null result pointer");
  double dummy;
  return conv_dispatch(d, nullptr, nullptr, nullptr, nullptr, want_stats ? &dummy : nullptr, 0, nullptr, 0, bytes, nullptr);
}

extern "C" int vfd_conv_forward(const vfd_conv_desc* d, const void* x, const void* packed, const float* bias, void* y,
                                double* stats, size_t stats_bytes, void* ws, size_t ws_bytes, void* stream) {
  return conv_dispatch(d, x, packed, bias, y, stats, stats_bytes, ws, ws_bytes, nullptr, stream);
}

// e4m3 operands (x: channels-last with CPAD16 channels, packed: vfd_pack_filter_fp8), bf16 output y = act(acc / (sx*sw) +
// bias), optional BatchNorm statistics; scale_x / scale_w are DEVICE scalars (what the tensors were multiplied by when
// quantised), read by the kernel, so that a captured step follows scales that change from step to step.
extern "C" int vfd_conv_forward_fp8(const vfd_conv_desc* d, const void* x, const float* scale_x, const void* packed,
                                    const float* scale_w, const float* bias, void* y, double* stats, size_t stats_bytes, void* stream) {
  VFD_REQUIRE(d != nullptr && d->dtype == VFD_FP8, "conv_forward_fp8: descriptor dtype must be VFD_FP8");
  VFD_REQUIRE(scale_x != nullptr && scale_w != nullptr, "conv_forward_fp8: null scale pointer");
  return conv_dispatch(d, x, packed, bias, y, stats, stats_bytes, nullptr, 0, nullptr, stream, no_mul(), scale_x, scale_w);
}

extern "C" int vfd_conv_forward_mul(const vfd_conv_desc* d, const void* x, const void* packed, const float* bias, void* y,
                                    const void* mul_src, int mul_act, float mul_slope, void* stream) {
  VFD_REQUIRE(mul_src != nullptr && ((uintptr_t)mul_src & 15) == 0, "conv_forward_mul: mul_src must be a 16-byte aligned tensor of y's shape");
  VFD_REQUIRE(mul_act >= VFD_ACT_NONE && mul_act <= VFD_ACT_TANH, "conv_forward_mul: bad activation %d", mul_act);
  MulP m = no_mul();
  m.src = mul_src; m.act = mul_act; m.slope = mul_slope;
  return conv_dispatch(d, x, packed, bias, y, nullptr, 0, nullptr, 0, nullptr, stream, m);
}

// the BatchNorm hand-over needs the row-store epilogue (conv_epilogue.hpp, VIA_LDS): bf16 tiles of >= 64 channels, which
// is what every dispatch path (conv_halo, conv_igemm) picks for more than 32 output channels
// ... but it is OFF by default (threshold "never"; vfd_conv_set_bn_handover_min_channels / VFD_BN_HANDOVER_MIN_C turn it on
// from a channel count).  Measured on ganomaly, 512 frames: the 256-channel tile's epilogue grows by what the separate
// reduce pass cost (30 us: a tie, two launches fewer), the 128- and 64-channel tiles grow by 78 / 150 us against a 60 /
// 100 us reduce pass — all workgroups reach their epilogues together, so the extra read is an HBM burst nothing hides — and
// since the filter gradients moved to a side stream the separate reduce pass overlaps with them: step 11.28 ms without
// the hand-over, 11.33 with it from 129 channels on.
static int g_bn_handover_min_c = -1;
static const int BN_HANDOVER_NEVER = 1 << 30;
extern "C" int vfd_conv_set_bn_handover_min_channels(int c) {
  const int prev = g_bn_handover_min_c < 0 ? BN_HANDOVER_NEVER : g_bn_handover_min_c;
  g_bn_handover_min_c = c < 33 ? 33 : c;
  return prev;
}
extern "C" int vfd_conv_bn_backward_supported(const vfd_conv_desc* d) {
  if (g_bn_handover_min_c < 0) g_bn_handover_min_c = getenv("VFD_BN_HANDOVER_MIN_C") ? atoi(getenv("VFD_BN_HANDOVER_MIN_C")) : BN_HANDOVER_NEVER;
  if (g_bn_handover_min_c < 33) g_bn_handover_min_c = 33;
  return d != nullptr && d->dtype == VFD_BF16 && d->Cout >= g_bn_handover_min_c;
}

extern "C" int vfd_conv_forward_bn_backward(const vfd_conv_desc* d, const void* x, const void* packed, void* y, const void* bn_x,
                                            const float* mean, const float* rstd, const float* gamma, const float* beta, int act,
                                            float slope, float* sums, size_t sums_bytes, void* stream) {
  VFD_REQUIRE(d != nullptr && d->dtype == VFD_BF16 && d->Cout > 32, "conv_forward_bn_backward: needs bf16 and more than 32 output channels");
  VFD_REQUIRE(bn_x != nullptr && ((uintptr_t)bn_x & 15) == 0 && mean && rstd && sums, "conv_forward_bn_backward: bad arguments");
  VFD_REQUIRE(sums_bytes >= (size_t)VFD_STATS_REPLICAS * 2 * cpad(d->Cout) * sizeof(float),
              "conv_forward_bn_backward: sums buffer holds %zu bytes, needs VFD_STATS_REPLICAS*2*CPAD(Cout) floats", sums_bytes);
  VFD_REQUIRE(act >= VFD_ACT_NONE && act <= VFD_ACT_TANH, "conv_forward_bn_backward: bad activation %d", act);
  VFD_REQUIRE(d->act == VFD_ACT_NONE, "conv_forward_bn_backward: the data-gradient convolution itself has no activation");
  MulP m = no_mul();
  m.src = bn_x; m.act = act; m.slope = slope;
  m.bn_mean = mean; m.bn_rstd = rstd; m.bn_gamma = gamma; m.bn_beta = beta; m.bn_sums = sums;
  return conv_dispatch(d, x, packed, nullptr, y, nullptr, 0, nullptr, 0, nullptr, stream, m);
}

// Tuning aid (not part of the public ABI): resident workgroups per CU the runtime grants the main kernels.
extern "C" int vfd_debug_occupancy(int* out, int n) {
  int k = 0, v = 0;
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, 0) != hipSuccess) return -1;
  if (k < n) out[k++] = (int)prop.maxSharedMemoryPerMultiProcessor;
  if (k < n) out[k++] = (int)prop.sharedMemPerBlock;
  if (k < n) out[k++] = prop.multiProcessorCount;
#define OCC(...)                                                                                       \
  if (k < n) {                                                                                         \
    v = -1;                                                                                            \
    (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&v, (const void*)(__VA_ARGS__), 256, 0);        \
    out[k++] = v;                                                                                      \
  }
  OCC(conv_igemm_kernel<bf16_t, 2, 2, 4, 4, 3, 1>)
  OCC(conv_igemm_kernel<bf16_t, 2, 2, 4, 4, 2, 2>)
  OCC(conv_igemm_kernel<bf16_t, 1, 4, 4, 4, 3, 1>)
  OCC(conv_igemm_kernel<bf16_t, 1, 4, 1, 4, 3, 1>)
#undef OCC
  return k;
}
