// Halo-tiled convolution on MFMA for layers whose taps walk the INPUT with unit stride (gfx950, bf16):
// stride-1 Conv3d / Conv2d, stride-1 transposed convolutions, and every output class of a strided transposed
// convolution (= the data gradient of a strided convolution), with at most 3 taps per dimension and class.
// These are the 32..64-channel 3-D layers of anogan's NetD / NetG (reference models/anogan.py:62-72, 84-93), the
// (1,3,3) / (3,1,1) factors of the (2+1)D blocks (models/spatiotempconv.py:45-55) and the 64-channel ends of the
// ganomaly pyramid (models/ganomaly.py:100-118).
//
// Why a second MFMA kernel: conv_igemm gathers the [pixel][K-slice] operand tap by tap from global memory, i.e. every
// input pixel crosses the global->LDS path once per tap (27x for a 3x3x3 filter).  With <= 64 output channels per tile
// the MFMA work per staged byte is too small for that path (80 B/clk/CU wanted, 28..54 delivered): those
// layers ran at 360..620 TFLOP/s.  Here a workgroup stages the input ONCE per 32-channel chunk as a halo block
// (TD+kd-1) x (TH+kh-1) x (16+kw-1) pixels and reads the B fragments of every tap from it at shifted rows: the
// im2col happens in the LDS addressing.  Only the [TILE_C][32] filter slice of a tap streams through a 2-deep ring.
//
//   LDS      halo rows of 64 B (one pixel's 32-channel chunk) stored W-MAJOR: row = wx * S + (dz * HH + hy) with the pitch
//            S = HD*HH rounded up to 1 (mod 4), 16-byte slot c of a row at c ^ 2*((wx>>2)&1).  A B fragment (16
//            consecutive w at fixed (d,h)) then touches rows S apart: bank quarter (r & 3) advances by 1 per pixel, the
//            swizzle depends on wx only, and the fragment is conflict-free for ds_read_b128 at any tap shift (checked
//            against the lane groups of MI355X_MICROARCH.md §LDS).  Above all the read address is
//            [lane constant for the w shift] + [wave-uniform (d,h) shift]: ONE vector add per fragment and stage.  (The
//            first version kept h*w-major rows and recomputed a row-dependent swizzle per tap: ~60 VALU per 16 MFMAs and
//            wave; an ablation showed the kernel's time did not change with the MFMAs removed — it was issue-bound.)
//            filter stage [taps of one (kd,kh)][TILE_C rows][64 B], rows 16-aligned, slot c of row r at c ^ 2*((r>>2)&1).
//   waves    4 per workgroup, each 64 pixels (4 groups of 16 consecutive w) x TILE_C channels, 2 workgroups per CU
//            (67.6 KB LDS): while one loads its next halo block the other computes.
//   sync     one raw s_barrier per (kd,kh) filter stage = up to 3 taps = 48 MFMAs per wave; `s_waitcnt vmcnt(0)` (every
//            DMA issued one stage earlier) — no counted waits: a halo block takes a varying number of DMA instructions.
#include "common.hpp"
#include "conv_epilogue.hpp"
#include <stdlib.h>
#include <type_traits>

namespace {

struct HaloP {
  const void* x;
  const void* w;
  EpiP e;
  int N, Di, Hi, Wi, Cip;
  int Do, Ho, Wo;
  int kh, kw;            // full filter extents (tap index arithmetic)
  int Kw;                // packed filter row length in elements = kd*kh*kw*Cip
  int transposed, sh, sw;   // class decomposition: cls = (rd*sh + rh)*sw + rw (transposed only)
  int ncls, ny;          // output classes, channel tiles
  int ntd, nth, ntw;     // pixel tiles per dimension (class 0, the largest)
  int per_xcd;           // workgroups per XCD group (grid = 8 * per_xcd)
  long long nwork;       // ncls * ny * N * ntd * nth * ntw
  FastDiv fgroup, fntw, fnth, fntd;
  DimClass dims[3][4];   // [d,h,w][class], stride <= 4
  int vP;                // conv_halo_rows_kernel: period of the virtual row axis (class rows per frame + 1 shared zero row)
  FastDiv fvP;
};

__device__ uint4 g_halo_zero_page[4];

constexpr int halo_pitch(int hdhh) { return hdhh + ((1 - hdhh) % 4 + 4) % 4; }    // >= hdhh, == 1 (mod 4)

template <int TILE_C, int TD, int TH>
struct HaloCfg {
  static constexpr int NI = TILE_C / 16;
  static constexpr int NJ = 4;
  static constexpr int HD_MAX = TD > 1 ? TD + 2 : 1;          // frames (TD == 1) carry no depth halo
  static constexpr int S_MAX = halo_pitch(HD_MAX * (TH + 2));
  static constexpr int ROWS_MAX = 18 * S_MAX;
  static constexpr int HALO_BYTES = (ROWS_MAX + 15) / 16 * 1024;
  static constexpr int FTAP = TILE_C * 64;                    // one tap's filter slice
  static constexpr int FSTAGE = 3 * FTAP;                     // the (up to 3) w-taps of one (kd, kh)
  static constexpr int EPI_BYTES = 256 * TILE_C * 2 + epi_red_bytes(TILE_C, 4, false) + 256 * 8;      // output rows + statistics block (conv_epilogue.hpp) + row offsets
  static constexpr int NFS = 2;                               // filter ring depth
  static constexpr int LDS = (HALO_BYTES + NFS * FSTAGE > EPI_BYTES) ? HALO_BYTES + NFS * FSTAGE : EPI_BYTES;
  static constexpr int NHW = ((ROWS_MAX + 15) / 16 + 3) / 4;  // halo DMA instructions per wave (upper bound)
  static constexpr int NFW = (3 * NI + 3) / 4;                // filter DMA instructions per wave and stage (upper bound)
};

// DBG bit 64 = the BatchNorm hand-over epilogue (a production variant, conv_epilogue.hpp BN).  The other DBG bits (tuning
// builds only, env VFD_HALO_DBG; results are WRONG): 1 = no LDS-DMA inside the loop, 2 = no barrier inside the
// loop, 4 = no MFMA, 8 = no main loop (prologue + epilogue only), 16 = no fragment reads
template <int TILE_C, int TD, int TH, int DBG = 0>
__global__ __launch_bounds__(256, 2) void conv_halo_kernel(const HaloP p) {
  using C = HaloCfg<TILE_C, TD, TH>;
  constexpr int NI = C::NI, NJ = C::NJ;
  static_assert(TD * TH == 16, "a tile is 16 groups of 16 pixels");
  __shared__ __attribute__((aligned(16))) char smem[C::LDS];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

  // ---- work item: (sample, pixel tile, class, channel tile).  Workgroups go round-robin to the 8 XCDs (id % 8): each
  // XCD takes a CONTIGUOUS range of work items, so tiles that share halo planes (and a tile's classes / channel tiles,
  // which read the same pixels) meet in one L2.
  const long long item = (long long)(blockIdx.x & 7) * p.per_xcd + (blockIdx.x >> 3);
  if (item >= p.nwork) return;
  uint32_t t = (uint32_t)item, jitem, itw, ith, itd;
  fdivmod(t, p.fgroup, t, jitem);
  fdivmod(t, p.fntw, t, itw);
  fdivmod(t, p.fnth, t, ith);
  fdivmod(t, p.fntd, t, itd);
  const int n = (int)t;
  int cls = (int)jitem / p.ny;
  const int ytile = (int)jitem - cls * p.ny;
  const int cls_id = cls;
  const int rw = p.transposed ? cls % p.sw : 0;
  if (p.transposed) cls /= p.sw;
  const int rh = p.transposed ? cls % p.sh : 0;
  if (p.transposed) cls /= p.sh;
  const int rd = p.transposed ? cls : 0;
  const DimClass dd = p.dims[0][rd], dh = p.dims[1][rh], dw = p.dims[2][rw];
  const int q0d = (int)itd * TD, q0h = (int)ith * TH, q0w = (int)itw * 16;
  if (q0d >= dd.Q || q0h >= dh.Q || q0w >= dw.Q) return;      // smaller classes have fewer tiles (uniform per workgroup)
  const int n0 = ytile * TILE_C;
  const int nchunks = (p.Cip + 31) >> 5;

  // halo block of this tile: input coordinates [o, o + H) per dimension
  const int HD = TD + dd.nk - 1, HH = TH + dh.nk - 1, HW = 16 + dw.nk - 1;
  const int od0 = q0d + dd.c0 - (dd.cs < 0 ? dd.nk - 1 : 0);
  const int oh0 = q0h + dh.c0 - (dh.cs < 0 ? dh.nk - 1 : 0);
  const int ow0 = q0w + dw.c0 - (dw.cs < 0 ? dw.nk - 1 : 0);
  const int HDH = HD * HH;
  const int S = HDH + ((1 - HDH) % 4 + 4) % 4;          // row pitch of one w column (1 mod 4)
  const int rows = HW * S;
  const int ninst = (rows + 15) >> 4;
  const int nstages = dd.nk * dh.nk * nchunks;           // one filter stage per (chunk, kd, kh)

  constexpr uint32_t NONE = 0xffffffffu;
  const int slot = lane & 3;
  const uint32_t smem_base = __builtin_amdgcn_readfirstlane(lds_addr(smem));
  const uint32_t halo_base = smem_base, filt_base = smem_base + C::HALO_BYTES;
  const char* zero = reinterpret_cast<const char*>(g_halo_zero_page);
  const char* xg = reinterpret_cast<const char*>(p.x);
  const char* wg = reinterpret_cast<const char*>(p.w);

  // ---- per-lane halo sources, once per tile: granule offset of the pixel's channel 0 (or NONE) and the 16-byte piece of
  // the 32-channel chunk this lane fetches (source-side swizzle: physical slot s of a row of column wx holds logical
  // chunk s ^ 2*((wx>>2)&1))
  uint32_t hoff[C::NHW];
  int hlc[C::NHW];
  {
    const float inv_s = 1.0f / (float)S, inv_hh = 1.0f / (float)HH;
    const int gpp = p.Cip >> 3;   // granules per pixel
#pragma unroll
    for (int k = 0; k < C::NHW; ++k) {
      const int r = (wave + 4 * k) * 16 + (lane >> 2);
      const int wx = (int)(((float)r + 0.5f) * inv_s);
      const int rem = r - wx * S;
      const int dz = (int)(((float)rem + 0.5f) * inv_hh);
      const int hy = rem - dz * HH;
      const int id = od0 + dz, ih = oh0 + hy, iw = ow0 + wx;
      uint32_t off = NONE;
      if (r < rows && rem < HDH && (unsigned)id < (unsigned)p.Di && (unsigned)ih < (unsigned)p.Hi && (unsigned)iw < (unsigned)p.Wi)
        off = (uint32_t)(((n * p.Di + id) * p.Hi + ih) * p.Wi + iw) * (uint32_t)gpp;
      hoff[k] = off;
      hlc[k] = slot ^ (((wx >> 2) & 1) << 1);
    }
  }
  auto issue_halo = [&](int chunk) __attribute__((always_inline)) {
#pragma unroll
    for (int k = 0; k < C::NHW; ++k) {
      const int inst = wave + 4 * k;       // wave-uniform
      if (inst < ninst) {
        const int ch = chunk * 32 + hlc[k] * 8;
        const char* src = (ch < p.Cip && hoff[k] != NONE) ? xg + ((size_t)hoff[k] << 4) + ch * 2 : zero;
        dma16_to_lds(src, halo_base + inst * 1024);
      }
    }
  };
  // ---- filter stage of (kd tap td, kh tap th, chunk): slices of the w-taps tw = 0 .. nkw-1, [tw][TILE_C rows][64 B];
  // DMA instruction `inst` of the stage = 16 rows: tap inst / NI, row block inst % NI; wave w issues inst = w, w+4, ...
  uint32_t wrow[C::NFW];         // granule offset of (row, tap 0, channel 0) for this lane's row of instruction k, or NONE
  int wtap[C::NFW];
#pragma unroll
  for (int k = 0; k < C::NFW; ++k) {
    const int inst = wave + 4 * k;
    const int co = n0 + (inst % NI) * 16 + (lane >> 2);
    wtap[k] = inst / NI;
    wrow[k] = (co < p.e.Cout) ? (uint32_t)(((long long)co * p.Kw) >> 3) : NONE;
  }
  const int flc = slot ^ (((lane >> 4) & 1) << 1);       // rows of a 16-row instruction: bit 2 of the row = bit 4 of the lane
  auto issue_filter = [&](int fs, int td, int th, int chunk) __attribute__((always_inline)) {
    const int tap0 = ((dd.k0 + td * dd.ks) * p.kh + (dh.k0 + th * dh.ks)) * p.kw + dw.k0;
    const int ch = chunk * 32 + flc * 8;
#pragma unroll
    for (int k = 0; k < C::NFW; ++k) {
      const int inst = wave + 4 * k;
      if (inst < dw.nk * NI) {
        const int tapidx = tap0 + wtap[k] * dw.ks;
        const char* src = (wrow[k] != NONE && ch < p.Cip) ? wg + ((size_t)(wrow[k] + (uint32_t)(tapidx * (p.Cip >> 3))) << 4) + ch * 2 : zero;
        dma16_to_lds(src, filt_base + fs * C::FSTAGE + inst * 1024);
      }
    }
  };

  // ---- fragment addressing
  const int l15 = lane & 15, c16 = (lane >> 4) << 4;
  // A (filter stage): row = 16 i + l15, rows start at multiples of 16: swizzle bit = bit 2 of l15
  const int a_off = (l15 << 6) + (c16 ^ ((l15 & 4) << 3));
  // B (halo): pixel group j of this wave at w shift sw: row = (sw + l15) * S + (dl * HH + hl); the (kd,kh) shift of a stage
  // adds a wave-uniform number of rows
  int baddr[NJ][3];
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    const int g = wave * 4 + j;
    const int base = (g / TH) * HH + (g % TH);
#pragma unroll
    for (int sw = 0; sw < 3; ++sw) {
      const int wx = sw + l15;
      baddr[j][sw] = ((wx * S + base) << 6) + (c16 ^ (((wx >> 2) & 1) << 5));
    }
  }

  f32x4 acc[NI][NJ];
#pragma unroll
  for (int i = 0; i < NI; ++i)
#pragma unroll
    for (int j = 0; j < NJ; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  // ---- main loop: one filter stage = the w-taps of one (chunk, kd, kh) per iteration, 2-deep filter ring.
  // Measured alternatives (tools/layer_bench.py --set anogan, 64->64 3-D layer forward / data gradient, TFLOP/s):
  //   one tap per stage, h*w-major halo, per-tap swizzle arithmetic, 3 workgroups per CU      950 / 987
  //   + next tap's fragments prefetched into a second register set                            804 / 825 (event-timed)
  //   this version (w-major halo, 3-tap stages, 2 workgroups per CU)                          978 / 996
  //   + 3-deep filter ring with a counted vmcnt (a slice issued two stages ahead)             941 / 959, and the 2-D
  //     transposed pyramid layer 597 -> 443 (its LDS grows from 48 to 60 KB: 3 -> 2 workgroups per CU)
  // An ablation (compile-time DBG variants) shows what is left: with the MFMAs removed the layer takes 1093 us, with the
  // DMAs removed 1360, prologue + epilogue alone 390, against 1450..1690 for the whole: the phases of a workgroup
  // (DMA wait -> barrier -> fragment reads -> MFMAs) run back to back and two workgroups per CU overlap them only partly.
  auto next_pos = [&](int& a_td, int& a_th, int& a_chunk) __attribute__((always_inline)) {
    if (++a_th == dh.nk) { a_th = 0; if (++a_td == dd.nk) { a_td = 0; ++a_chunk; } }
  };
  int td = 0, th = 0, chunk = 0;            // stage s
  int td1 = 0, th1 = 0, chunk1 = 0;          // stage s+1 (the one to issue)
  issue_halo(0);
  issue_filter(0, 0, 0, 0);
  next_pos(td1, th1, chunk1);
  int fs = 0;
  for (int s = 0; s < ((DBG & 8) ? 0 : nstages); ++s) {
    const bool new_chunk = (td | th) == 0 && chunk > 0;
    // this stage's filter slices (issued one stage ago) and, in stage 0, the first halo block have landed; every wave is
    // done with the LDS reads of the previous stage (the barrier waits for no counter: retire them explicitly)
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    if (!(DBG & 2)) __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if (new_chunk && !(DBG & 1)) issue_halo(chunk);                 // the old block is dead: all waves passed the barrier above
    if (s + 1 < nstages && !(DBG & 1)) issue_filter(fs ^ 1, td1, th1, chunk1);
    next_pos(td1, th1, chunk1);
    if (new_chunk) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      if (!(DBG & 2)) __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
    }
    const int sd_ = dd.cs > 0 ? td : dd.nk - 1 - td, sh_ = dh.cs > 0 ? th : dh.nk - 1 - th;
    const int dh_off = (sd_ * HH + sh_) << 6;            // wave-uniform byte shift of this (kd, kh)
    const char* ft = smem + C::HALO_BYTES + fs * C::FSTAGE;
#pragma unroll
    for (int sw = 0; sw < 3; ++sw) {
      if (sw < dw.nk) {
        const int tw = dw.cs > 0 ? sw : dw.nk - 1 - sw;   // filter tap whose input shift is sw
        bf16x8 a[NI], b[NJ];
#pragma unroll
        for (int i = 0; i < NI; ++i) {
          if (DBG & 16) { asm volatile("" : "=v"(a[i]) : "v"(ft + tw * C::FTAP + a_off)); continue; }
          a[i] = *reinterpret_cast<const bf16x8*>(ft + tw * C::FTAP + i * 1024 + a_off);
        }
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
          if (DBG & 16) { asm volatile("" : "=v"(b[j]) : "v"(baddr[j][sw] + dh_off)); continue; }
          b[j] = *reinterpret_cast<const bf16x8*>(smem + baddr[j][sw] + dh_off);
        }
        if (!(DBG & 32)) __builtin_amdgcn_s_setprio(1);      // keeps the cluster contiguous: +2 % (DBG 32 = without)
#pragma unroll
        for (int i = 0; i < NI; ++i)
#pragma unroll
          for (int j = 0; j < NJ; ++j) {
            if (DBG & 4) { asm volatile("" :: "v"(a[i]), "v"(b[j])); continue; }
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
          }
        if (!(DBG & 32)) __builtin_amdgcn_s_setprio(0);
      }
    }
    fs ^= 1;
    next_pos(td, th, chunk);
  }
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");

  // ---- epilogue (conv_epilogue.hpp): tile row r = 16 g + wl, g = TH * dl + hl
  auto row_q = [&](int r, int& qd, int& qh, int& qw) __attribute__((always_inline)) {
    const int g = r >> 4;
    qd = q0d + g / TH; qh = q0h + g % TH; qw = q0w + (r & 15);
    return qd < dd.Q && qh < dh.Q && qw < dw.Q;
  };
  auto out_offset = [&](int r) __attribute__((always_inline)) -> long long {
    int qd, qh, qw;
    if (!row_q(r, qd, qh, qw)) return -1;
    return (((long long)(n * p.Do + qd * dd.so + dd.r) * p.Ho + qh * dh.so + dh.r) * p.Wo + qw * dw.so + dw.r) * p.e.Cop;
  };
  conv_epilogue<bf16_t, 1, 4, NI, NJ, C::LDS, (DBG & 64) != 0, false>(smem, acc, p.e, n0, (int)(item & 0x7fffffff) + cls_id, out_offset,
                                             [&](int r) { int a_, b_, c_; return row_q(r, a_, b_, c_); });
}

// ---- frames under classes of at most 2 x 2 taps (the k4 s2 p1 transposed form: ganomaly's 64-channel pyramid ends) ----------
// The <.,1,16> tile above cuts every frame into 16 x 16-pixel tiles of its own: 28 class rows = 16 + 12, i.e. a quarter of
// the row tiles is half empty, and its 67 KB of LDS / 143 VGPRs hold 2 workgroups per CU whose phases (halo fetch, filter
// stages, output rows) run one after the other.  An ablation on the 28 x 28-per-class layer (round 3; -DVFD_HALO_TUNING
// builds) shows the phases ADD UP: fetch 26 + loop DMAs 56 + MFMAs 48 + barriers 26 + output rows 42 = 198 of 206 us.
// This variant changes three things:
//   * the row axis is VIRTUAL: frame n's class rows q = 0 .. Q-1 sit at v = n (Q + 1) + q, and v = n (Q + 1) + Q is a zero row
//     shared by frame n (its row Q) and frame n + 1 (its row -1).  Tiles are 16 consecutive virtual rows, whatever Q is: no
//     partial row tile per frame; 1 / (Q + 1) of the rows are the zero rows (3.4 % at Q = 28).  Needs every class to read rows
//     -1 .. Q of a Q-row input at most, which the host checks.
//   * 35 KB of LDS (2-tap filter stages, 17-row halo) and <= 128 VGPRs (halo source addresses recomputed per chunk, the
//     B-fragment swizzle kept as flip bits): 4 workgroups per CU, so that one workgroup's fetch and output phases meet
//     another's MFMAs.
//   * pixel groups are cut from the tile in RASTER order (group g = pixels 16 g .. 16 g + 15 of the 16 x WB block), which
//     for WB = 16 is the old row = group mapping and for other widths lets groups straddle rows (a B fragment is still
//     [per-lane constant] + [wave-uniform tap shift]).
// Measured (tools/layer_bench.py, dec.pyr 128->64 convT @28, forward, same box): <64,1,16> 197 us; WB = 28 (7 waves, every
// pixel of a 28-wide class row in use, 2 workgroups per CU, halo double-buffered with a counted vmcnt) 200 us — fewer MFMAs
// and a third of the DMA bytes per MFMA bought nothing, the phases still add up; WB = 16 with 4 workgroups per CU 180 us.
template <int TILE_C, int WB, int NHB_>
struct RowsCfg {
  static constexpr int NI = TILE_C / 16;
  static constexpr int NJ = 4;
  static constexpr int NW = WB / 4;                           // waves
  static constexpr int KMAX = 2;                              // taps per dimension and class (k4 s2 transposed: 2 x 2)
  static constexpr int S_MAX = halo_pitch(16 + KMAX - 1);
  static constexpr int ROWS_MAX = (WB + KMAX - 1) * S_MAX;
  static constexpr int NHI = (ROWS_MAX + 15) / 16;            // halo DMA instructions (upper bound)
  static constexpr int HALO_BYTES = NHI * 1024;
  static constexpr int NHB = NHB_;                            // halo buffers: 2 = chunk c + 1 is staged while chunk c is multiplied
  static constexpr int FTAP = TILE_C * 64;
  static constexpr int FSTAGE = KMAX * FTAP;
  static constexpr int NFS = 2;
  static constexpr int TILE_P = 16 * WB;
  static constexpr int EPI_BYTES = TILE_P * TILE_C * 2 + epi_red_bytes(TILE_C, NW, false) + TILE_P * 8;
  static constexpr int LDS = (NHB * HALO_BYTES + NFS * FSTAGE > EPI_BYTES) ? NHB * HALO_BYTES + NFS * FSTAGE : EPI_BYTES;
  static constexpr int NHW = (NHI + NW - 1) / NW;             // halo DMA instructions per wave
  static constexpr int NFW = (KMAX * NI + NW - 1) / NW;       // filter DMA instructions per wave and stage
};

// DBG (tuning builds only, -DVFD_HALO_TUNING + env VFD_HALO_DBG; results are WRONG): 1 = no LDS-DMA inside the loop, 4 = no MFMA,
// 8 = no main loop, 16 = no epilogue
template <int TILE_C, int WB, int NHB, int DBG = 0>
__global__ __launch_bounds__(WB / 4 * 64, 4) void conv_halo_rows_kernel(const HaloP p) {
  using C = RowsCfg<TILE_C, WB, NHB>;
  constexpr int NI = C::NI, NJ = C::NJ, NW = C::NW;
  static_assert(WB % 4 == 0 && C::NHW <= 5, "16 x WB pixels = WB groups of 16 = WB / 4 waves of 4 groups; the counted halo wait covers <= 5 instructions per wave");
  __shared__ __attribute__((aligned(16))) char smem[C::LDS];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const long long item = (long long)(blockIdx.x & 7) * p.per_xcd + (blockIdx.x >> 3);
  if (item >= p.nwork) return;
  uint32_t t = (uint32_t)item, jitem, itw;
  fdivmod(t, p.fgroup, t, jitem);
  fdivmod(t, p.fntw, t, itw);
  const int itv = (int)t;
  int cls = (int)jitem / p.ny;
  const int ytile = (int)jitem - cls * p.ny;
  const int rw = p.transposed ? cls % p.sw : 0;
  const int rh = p.transposed ? (cls / p.sw) % p.sh : 0;
  const DimClass dh = p.dims[1][rh], dw = p.dims[2][rw];
  const int P = p.vP;
  const int v0 = itv * 16, q0w = (int)itw * WB;
  if (v0 >= p.N * P || q0w >= dw.Q) return;
  const int n0 = ytile * TILE_C;
  const int nchunks = (p.Cip + 31) >> 5;

  const int HH = 16 + dh.nk - 1, HW = WB + dw.nk - 1;
  const int oh0 = v0 + dh.c0 - (dh.cs < 0 ? dh.nk - 1 : 0);       // VIRTUAL row of halo row 0
  const int ow0 = q0w + dw.c0 - (dw.cs < 0 ? dw.nk - 1 : 0);
  const int S = HH + ((1 - HH) % 4 + 4) % 4;
  const int rows = HW * S;
  const int ninst = (rows + 15) >> 4;
  const int nstages = dh.nk * nchunks;

  constexpr uint32_t NONE = 0xffffffffu;
  const int slot = lane & 3;
  const uint32_t smem_base = __builtin_amdgcn_readfirstlane(lds_addr(smem));
  const uint32_t halo_base = smem_base, filt_base = smem_base + C::NHB * C::HALO_BYTES;
  const char* zero = reinterpret_cast<const char*>(g_halo_zero_page);
  const char* xg = reinterpret_cast<const char*>(p.x);
  const char* wg = reinterpret_cast<const char*>(p.w);

  // per-lane halo sources are recomputed for every chunk (4 chunks per tile, ~15 VALU per DMA instruction) instead of being
  // kept: 7 waves x 2 workgroups per CU = 4 waves on some SIMDs, i.e. a 128-VGPR budget (the <64,1,16> tile above spends 143)
  const float inv_s = 1.0f / (float)S;
  const int gpp = p.Cip >> 3;
  auto issue_halo = [&](int chunk) __attribute__((always_inline)) {
#pragma unroll
    for (int k = 0; k < C::NHW; ++k) {
      const int inst = wave + NW * k;
      if (inst < ninst) {
        int r = inst * 16 + (lane >> 2);
        asm volatile("" : "+v"(r));      // (keeps the chunk-invariant address arithmetic below from being hoisted back into registers)
        const int wx = (int)(((float)r + 0.5f) * inv_s);
        const int hy = r - wx * S;
        const int u = oh0 + hy, iw = ow0 + wx;
        const int ch = chunk * 32 + (slot ^ (((wx >> 2) & 1) << 1)) * 8;
        const char* src = zero;
        if (r < rows && hy < HH && u >= 0 && (unsigned)iw < (unsigned)p.Wi && ch < p.Cip) {
          uint32_t nu, ih;
          fdivmod((uint32_t)u, p.fvP, nu, ih);
          if ((int)nu < p.N && (int)ih < p.Hi)
            src = xg + ((size_t)((uint32_t)(((int)nu * p.Hi + (int)ih) * p.Wi + iw) * (uint32_t)gpp) << 4) + ch * 2;
        }
        dma16_to_lds(src, halo_base + (NHB == 2 ? (chunk & 1) * C::HALO_BYTES : 0) + inst * 1024);
      }
    }
  };
  int nhalo_w = 0;                 // halo DMA instructions THIS wave issues per chunk (wave-uniform): the counted wait below
#pragma unroll
  for (int k = 0; k < C::NHW; ++k) nhalo_w += (wave + NW * k < ninst) ? 1 : 0;
  uint32_t wrow[C::NFW];
  int wtap[C::NFW];
#pragma unroll
  for (int k = 0; k < C::NFW; ++k) {
    const int inst = wave + NW * k;
    const int co = n0 + (inst % NI) * 16 + (lane >> 2);
    wtap[k] = inst / NI;
    wrow[k] = (co < p.e.Cout) ? (uint32_t)(((long long)co * p.Kw) >> 3) : NONE;
  }
  const int flc = slot ^ (((lane >> 4) & 1) << 1);
  auto issue_filter = [&](int fs, int th, int chunk) __attribute__((always_inline)) {
    const int tap0 = (dh.k0 + th * dh.ks) * p.kw + dw.k0;      // frames: one depth tap (index 0)
    const int ch = chunk * 32 + flc * 8;
#pragma unroll
    for (int k = 0; k < C::NFW; ++k) {
      const int inst = wave + NW * k;
      if (inst < dw.nk * NI) {
        const int tapidx = tap0 + wtap[k] * dw.ks;
        const char* src = (wrow[k] != NONE && ch < p.Cip) ? wg + ((size_t)(wrow[k] + (uint32_t)(tapidx * (p.Cip >> 3))) << 4) + ch * 2 : zero;
        dma16_to_lds(src, filt_base + fs * C::FSTAGE + inst * 1024);
      }
    }
  };

  const int l15 = lane & 15, c16 = (lane >> 4) << 4;
  const int a_off = (l15 << 6) + (c16 ^ ((l15 & 4) << 3));
  // B fragment of group j at w shift sw: ((pw + sw) * S + prow) * 64 + (c16 ^ 32 * bit 2 of (pw + sw)) = bbase[j] + sw * S * 64 +
  // 32 * (flip(sw) ? (bit 5 of bbase[j] ? -1 : +1) : 0), flip(sw) = bit 2 of (pw + sw) differs from bit 2 of pw: kept as one
  // bit per (j, sw) in `bflip` instead of 12 address registers
  int bbase[NJ];
  uint32_t bflip = 0;
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    const int px = (wave * 4 + j) * 16 + l15;       // raster index in the 16 x WB block
    const int prow = px / WB, pw = px - prow * WB;
    bbase[j] = ((pw * S + prow) << 6) + (c16 ^ (((pw >> 2) & 1) << 5));
#pragma unroll
    for (int sw = 1; sw < C::KMAX; ++sw)
      if ((((pw + sw) >> 2) & 1) != ((pw >> 2) & 1)) bflip |= 1u << (j * 2 + sw - 1);
  }
  auto baddr = [&](int j, int sw, int uni) __attribute__((always_inline)) -> int {      // uni = sw * S * 64 + the stage's (kh) shift
    int a = bbase[j] + uni;
    if (sw > 0) a ^= (int)((bflip >> (j * 2 + sw - 1)) & 1u) << 5;      // bit 5 only: c16 ^ 32, the row part is a multiple of 64
    return a;
  };

  f32x4 acc[NI][NJ];
#pragma unroll
  for (int i = 0; i < NI; ++i)
#pragma unroll
    for (int j = 0; j < NJ; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  int th = 0, chunk = 0, th1 = 0, chunk1 = 0;
  auto next_pos = [&](int& a_th, int& a_chunk) __attribute__((always_inline)) {
    if (++a_th == dh.nk) { a_th = 0; ++a_chunk; }
  };
  // The halo block of chunk c + 1 is issued in the FIRST stage of chunk c, after that stage's filter prefetch, into the other
  // halo buffer (free: its last readers, chunk c - 1, are behind the barrier).  DMAs land in issue order, so the next stage
  // top waits for its filter slices with vmcnt(<this wave's halo instructions>) and leaves the halo block in flight until the
  // first stage of chunk c + 1 (vmcnt(0)): the block has a whole chunk of MFMAs to arrive.  (Single buffer, issued and
  // awaited at the chunk boundary as in conv_halo_kernel: the 28 x 28-per-class layer 199 us; this form: see DESIGN.md.)
  issue_halo(0);
  issue_filter(0, 0, 0);
  next_pos(th1, chunk1);
  int fs = 0;
  bool halo_pending = false;
  for (int s = 0; s < ((DBG & 8) ? 0 : nstages); ++s) {
    if (halo_pending && th != 0) {
      switch (nhalo_w) {
        case 0: asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); break;
        case 1: asm volatile("s_waitcnt vmcnt(1) lgkmcnt(0)" ::: "memory"); break;
        case 2: asm volatile("s_waitcnt vmcnt(2) lgkmcnt(0)" ::: "memory"); break;
        case 3: asm volatile("s_waitcnt vmcnt(3) lgkmcnt(0)" ::: "memory"); break;
        case 4: asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory"); break;
        default: asm volatile("s_waitcnt vmcnt(5) lgkmcnt(0)" ::: "memory"); break;
      }
    } else {
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    // (DBG 32: the stage's DMAs are issued here, before its fragment reads and MFMAs; default: between the two w-taps)
    auto issue_next = [&]() __attribute__((always_inline)) {
      if (s + 1 < nstages && !(DBG & 1)) issue_filter(fs ^ 1, th1, chunk1);
      next_pos(th1, chunk1);
      halo_pending = false;
      if (NHB == 2 && th == 0 && chunk + 1 < nchunks && !(DBG & 1)) { issue_halo(chunk + 1); halo_pending = true; }
    };
    if (NHB == 1 && th == 0 && chunk > 0) {      // one halo buffer: the block is fetched at the chunk boundary (all readers are behind the barrier)
      if (!(DBG & 1)) issue_halo(chunk);
      issue_next();
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
    } else if ((DBG & 32) || NHB == 1) issue_next();
    const int sh_ = dh.cs > 0 ? th : dh.nk - 1 - th;
    const int dh_off = (sh_ << 6) + (NHB == 2 ? (chunk & 1) * C::HALO_BYTES : 0);
    const char* ft = smem + C::NHB * C::HALO_BYTES + fs * C::FSTAGE;
#pragma unroll
    for (int sw = 0; sw < C::KMAX; ++sw) {
      if (sw < dw.nk) {
        const int tw = dw.cs > 0 ? sw : dw.nk - 1 - sw;
        bf16x8 a[NI], b[NJ];
#pragma unroll
        for (int i = 0; i < NI; ++i) a[i] = *reinterpret_cast<const bf16x8*>(ft + tw * C::FTAP + i * 1024 + a_off);
#pragma unroll
        for (int j = 0; j < NJ; ++j) b[j] = *reinterpret_cast<const bf16x8*>(smem + baddr(j, sw, sw * (S << 6) + dh_off));
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int i = 0; i < NI; ++i)
#pragma unroll
          for (int j = 0; j < NJ; ++j) {
            if (DBG & 4) { asm volatile("" :: "v"(a[i]), "v"(b[j])); continue; }
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
          }
        __builtin_amdgcn_s_setprio(0);
      }
      if (sw == 0 && !(DBG & 32) && NHB == 2) issue_next();
    }
    fs ^= 1;
    next_pos(th, chunk);
  }
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");

  if (DBG & 16) return;
  // ---- epilogue: tile row r = raster index of the pixel in the 16 x WB block
  auto row_q = [&](int r, int& n, int& qh, int& qw) __attribute__((always_inline)) {
    const int prow = r / WB;
    qw = q0w + (r - prow * WB);
    uint32_t nu, q;
    fdivmod((uint32_t)(v0 + prow), p.fvP, nu, q);
    n = (int)nu; qh = (int)q;
    return n < p.N && qh < dh.Q && qw < dw.Q;
  };
  auto out_offset = [&](int r) __attribute__((always_inline)) -> long long {
    int n, qh, qw;
    if (!row_q(r, n, qh, qw)) return -1;
    return (((long long)n * p.Ho + qh * dh.so + dh.r) * p.Wo + qw * dw.so + dw.r) * p.e.Cop;
  };
  conv_epilogue<bf16_t, 1, NW, NI, NJ, C::LDS, false, false>(smem, acc, p.e, n0, (int)(item & 0x7fffffff), out_offset,
                                                            [&](int r) { int a_, b_, c_; return row_q(r, a_, b_, c_); });
}

template <int TILE_C, int TD, int TH>
int launch_halo(const HaloP& p, hipStream_t st) {
  const long long nwg = (long long)p.per_xcd * 8;
  if (nwg >= 0x7fffffffLL) return 0;
#ifdef VFD_HALO_TUNING
  static const int dbg = getenv("VFD_HALO_DBG") ? atoi(getenv("VFD_HALO_DBG")) : 0;
  if (TILE_C == 64 && TD == 4) {
    switch (dbg) {
      case 1: hipLaunchKernelGGL((conv_halo_kernel<64, 4, 4, 1>), dim3((unsigned)nwg), dim3(256), 0, st, p); return 1;
      case 2: hipLaunchKernelGGL((conv_halo_kernel<64, 4, 4, 2>), dim3((unsigned)nwg), dim3(256), 0, st, p); return 1;
      case 4: hipLaunchKernelGGL((conv_halo_kernel<64, 4, 4, 4>), dim3((unsigned)nwg), dim3(256), 0, st, p); return 1;
      case 5: hipLaunchKernelGGL((conv_halo_kernel<64, 4, 4, 5>), dim3((unsigned)nwg), dim3(256), 0, st, p); return 1;
      case 7: hipLaunchKernelGGL((conv_halo_kernel<64, 4, 4, 7>), dim3((unsigned)nwg), dim3(256), 0, st, p); return 1;
      case 8: hipLaunchKernelGGL((conv_halo_kernel<64, 4, 4, 8>), dim3((unsigned)nwg), dim3(256), 0, st, p); return 1;
      case 16: hipLaunchKernelGGL((conv_halo_kernel<64, 4, 4, 16>), dim3((unsigned)nwg), dim3(256), 0, st, p); return 1;
      case 20: hipLaunchKernelGGL((conv_halo_kernel<64, 4, 4, 20>), dim3((unsigned)nwg), dim3(256), 0, st, p); return 1;
      case 23: hipLaunchKernelGGL((conv_halo_kernel<64, 4, 4, 23>), dim3((unsigned)nwg), dim3(256), 0, st, p); return 1;
      case 32: hipLaunchKernelGGL((conv_halo_kernel<64, 4, 4, 32>), dim3((unsigned)nwg), dim3(256), 0, st, p); return 1;
      default: break;
    }
  }
#endif
  if constexpr (TILE_C == 64) {
    if (p.e.mul.bn_mean != nullptr) {      // BatchNorm hand-over: epilogue variant of its own (DBG bit 64, conv_epilogue.hpp)
      hipLaunchKernelGGL((conv_halo_kernel<TILE_C, TD, TH, 64>), dim3((unsigned)nwg), dim3(256), 0, st, p);
      return hipGetLastError() == hipSuccess ? 1 : -1;
    }
  }
  if (p.e.mul.bn_mean != nullptr) return 0;   // 32-channel tiles leave through the direct store: conv_igemm refuses in turn
  hipLaunchKernelGGL((conv_halo_kernel<TILE_C, TD, TH>), dim3((unsigned)nwg), dim3(256), 0, st, p);
  return hipGetLastError() == hipSuccess ? 1 : -1;
}

}  // namespace

static int g_halo_mode = -1;     // -1: not yet read from the environment (VFD_NO_HALO=1 -> mode 1)

extern "C" int vfd_conv_set_halo_mode(int mode) {
  const int prev = g_halo_mode < 0 ? (getenv("VFD_NO_HALO") != nullptr ? 1 : 0) : g_halo_mode;
  g_halo_mode = (mode >= 0 && mode <= 2) ? mode : 0;
  return prev;
}

// 1 = handled (or, with query, would be handled), 0 = not a halo shape, < 0 = launch error
int vfd_conv_halo_try(const vfd_conv_desc* d, const void* x, const void* packed, const float* bias, void* y, double* stats,
                      const MulP& mul, bool query, hipStream_t st) {
  if (g_halo_mode < 0) g_halo_mode = getenv("VFD_NO_HALO") != nullptr ? 1 : 0;
  if (g_halo_mode == 1 || d->dtype != VFD_BF16) return 0;
  const int k[3] = {d->kd, d->kh, d->kw}, s[3] = {d->sd, d->sh, d->sw}, pp[3] = {d->pd, d->ph, d->pw};
  const int O[3] = {d->Do, d->Ho, d->Wo};
  for (int i = 0; i < 3; ++i) {
    if (!d->transposed && s[i] != 1) return 0;                      // the taps must walk the input with unit stride
    if (d->transposed && (s[i] > 4 || (k[i] + s[i] - 1) / s[i] > 3)) return 0;
    if (!d->transposed && k[i] > 3) return 0;
  }
  static const int max_cout = getenv("VFD_HALO_MAX_COUT") ? atoi(getenv("VFD_HALO_MAX_COUT")) : 64;      // tuning: channel tiles of 64 beyond the first
  if (d->Cout > max_cout || d->Cin < 25) return 0;      // wider outputs: conv_igemm's 128/256-channel tiles; thin inputs: conv_small / igemm
  HaloP p;
  p.x = x; p.w = packed;
  p.e.y = y; p.e.bias = bias; p.e.stats = stats; p.e.Cop = cpad(d->Cout); p.e.Cout = d->Cout; p.e.act = d->act; p.e.slope = d->slope;
  p.e.mul = mul;
  p.N = d->N; p.Di = d->Di; p.Hi = d->Hi; p.Wi = d->Wi; p.Cip = cpad(d->Cin);
  p.Do = d->Do; p.Ho = d->Ho; p.Wo = d->Wo;
  p.kh = d->kh; p.kw = d->kw;
  p.Kw = d->kd * d->kh * d->kw * p.Cip;
  p.transposed = d->transposed; p.sh = d->sh; p.sw = d->sw;
  p.ncls = d->transposed ? d->sd * d->sh * d->sw : 1;
  int Q0[3];
  for (int i = 0; i < 3; ++i) {
    const int nr = d->transposed ? s[i] : 1;
    for (int r = 0; r < nr; ++r) p.dims[i][r] = make_dim_host(d->transposed, r, k[i], s[i], pp[i], O[i]);
    Q0[i] = p.dims[i][0].Q;
    for (int r = 0; r < nr; ++r)
      if (p.dims[i][r].nk < 1) return 0;           // a class without taps (k < s): leave it to the general kernel
  }
  const bool frames = Q0[0] == 1 && d->Di == 1;
  if (frames) {
    // the <.,1,16> tile is sized WITHOUT a depth halo (HaloCfg::HD_MAX == 1): a depth-1 input under a filter with depth taps
    // (Conv3d(k=3, pad=1) on (N,C,1,H,W), or a class of a transposed form with 2-3 depth taps) would stage HD = 3 planes past
    // HALO_BYTES into the filter ring.  Those shapes stay on conv_igemm.
    const int nr = d->transposed ? s[0] : 1;
    for (int r = 0; r < nr; ++r)
      if (p.dims[0][r].nk != 1) return 0;
  }
  if (!frames && Q0[0] < 3) return 0;              // 2-deep volumes would leave half of a 4-deep tile empty
  if (Q0[2] < 12 || Q0[1] < (frames ? 12 : 3)) return 0;
  const int TD = frames ? 1 : 4, TH = frames ? 16 : 4;
  static const bool no16 = getenv("VFD_HALO_NO_16C") != nullptr;      // A/B switch (layer benchmarks)
  const int tile_c = d->Cout > 32 ? 64 : (d->Cout > 16 || frames || no16) ? 32 : 16;      // 16: the 3-channel ends of the 3-D nets
  p.ny = (d->Cout + tile_c - 1) / tile_c;
  p.vP = 0;
  static const bool no_rows = getenv("VFD_HALO_NO_ROWS") != nullptr;      // A/B switch (layer benchmarks)
  if (frames && tile_c == 64 && mul.bn_mean == nullptr && !no_rows) {
    // conv_halo_rows_kernel<64, 28>: every class grid Q x (multiple of 28), every class reads input rows -1 .. Q of Hi == Q rows
    bool ok = true;
    const int nrh = d->transposed ? s[1] : 1, nrw = d->transposed ? s[2] : 1;
    for (int r = 0; r < nrh && ok; ++r) {
      const DimClass& c = p.dims[1][r];
      const int lo = c.c0 - (c.cs < 0 ? c.nk - 1 : 0), hi = c.c0 + (c.cs > 0 ? c.nk - 1 : 0);
      ok = c.Q == d->Hi && c.Q == Q0[1] && lo >= -1 && hi <= 1 && c.nk <= 2;
    }
    const int wb = 16;
    for (int r = 0; r < nrw && ok; ++r) ok = p.dims[2][r].Q == Q0[2] && p.dims[2][r].nk <= 2;
    const long long vrows = (long long)d->N * (Q0[1] + 1);
    if (ok && vrows < (1 << 24)) {
      p.vP = Q0[1] + 1;
      p.fvP = make_fastdiv((uint32_t)p.vP);
      p.ntd = 1; p.nth = (int)((vrows + 15) / 16); p.ntw = (Q0[2] + wb - 1) / wb;
    }
  }
  if (p.vP == 0) { p.ntd = (Q0[0] + TD - 1) / TD; p.nth = (Q0[1] + TH - 1) / TH; p.ntw = (Q0[2] + 15) / 16; }
  const long long tiles = p.vP != 0 ? (long long)p.nth * p.ntw : (long long)d->N * p.ntd * p.nth * p.ntw;
  p.nwork = tiles * p.ncls * p.ny;
  if (p.nwork < 512 && g_halo_mode != 2) return 0;     // few tiles: conv_igemm's split-K paths
  // 32-bit granule addressing (as conv_igemm)
  const long long in_px = (long long)d->N * d->Di * d->Hi * d->Wi;
  if (in_px >= 0x7fffffffLL || in_px * p.Cip / 8 >= 0xffffffffLL || (long long)d->Cout * p.Kw / 8 >= 0xffffffffLL) return 0;
  if (query) return p.vP != 0 ? 2 : 1;
  p.per_xcd = (int)((p.nwork + 7) / 8);
  p.fgroup = make_fastdiv((uint32_t)(p.ncls * p.ny));
  p.fntw = make_fastdiv((uint32_t)p.ntw); p.fnth = make_fastdiv((uint32_t)p.nth); p.fntd = make_fastdiv((uint32_t)p.ntd);
  if (p.nwork >= 0x7fffffffLL) return 0;
  if (p.vP == 0) {
    // host-side bound of what the kernel stages: every class's halo block must fit the tile's compile-time halo region
    const int hd_max = frames ? 1 : TD + 2, s_max = halo_pitch(hd_max * (TH + 2)), rows_max = 18 * s_max;
    for (int rd = 0; rd < (d->transposed ? s[0] : 1); ++rd)
      for (int rh = 0; rh < (d->transposed ? s[1] : 1); ++rh)
        for (int rw = 0; rw < (d->transposed ? s[2] : 1); ++rw) {
          const int HD = TD + p.dims[0][rd].nk - 1, HH = TH + p.dims[1][rh].nk - 1, HW = 16 + p.dims[2][rw].nk - 1;
          VFD_REQUIRE(HD <= hd_max && HW * halo_pitch(HD * HH) <= rows_max, "conv_halo: halo block %dx%dx%d exceeds the tile's LDS region", HD, HH, HW);
        }
  }
  if (p.vP != 0) {
    const long long nwg = (long long)p.per_xcd * 8;
    hipLaunchKernelGGL((conv_halo_rows_kernel<64, 16, 1>), dim3((unsigned)nwg), dim3(256), 0, st, p);
    return hipGetLastError() == hipSuccess ? 1 : -1;
  }
  if (frames) return tile_c == 64 ? launch_halo<64, 1, 16>(p, st) : launch_halo<32, 1, 16>(p, st);
  return tile_c == 64 ? launch_halo<64, 4, 4>(p, st) : tile_c == 32 ? launch_halo<32, 4, 4>(p, st) : launch_halo<16, 4, 4>(p, st);
}
