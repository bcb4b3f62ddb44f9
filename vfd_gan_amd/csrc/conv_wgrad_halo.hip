// Filter gradient of stride-1 convolutions with a 3 x 3 spatial footprint (kd = 1 or 3, kh = kw = 3, pad 1), bf16, as a
// halo-tiled split-K MFMA kernel (gfx950).  These are the 3-D layers of anogan's NetD / NetG (reference
// models/anogan.py:44-45,49-57,64-65,84-89,96-102) and the (1,3,3) spatial factors of the (2+1)D blocks
// (models/spatiotempconv.py:45-47): in the anogan step the filter gradient was the largest kernel group (21 of 65 ms)
// at 370..600 TFLOP/s, because conv_wgrad stages the gathered operand G once per TAP: 20 KB of LDS-DMA per 256 MFMA
// cycles (80 B/clk/CU wanted, 28..54 delivered).
//
//     dWp[r][t][c] = sum over pixels (n,q) of  S[n,q][r] * G[n, q - 1 + t][c]        (stride 1, pad 1; see conv_wgrad.hip)
//
// Work item = (64 channels of S) x (64 channels of G) x (the 9 in-plane taps of ONE depth tap kd) x (a range of pixel
// blocks).  A pixel block is 8 rows x 16 pixels of one (n, d) plane; per block the workgroup stages S [128 px][64 ch] and
// the G halo [10 x 18 px][64 ch] ONCE and reads all 9 taps' fragments from the halo at shifted rows: 40 KB of DMA per
// 2304 MFMA cycles (17 B/clk/CU).  8 waves = 2 (32 S-channels) x 4 (16 G-channels); a wave owns 32 x (9 taps x 16)
// outputs = 72 accumulator registers and runs 4 K-steps of 32 pixels (two 16-pixel rows) per block.  One workgroup per
// CU (its registers hold the 64 x 576 accumulator tile), 3-stage ring, ONE barrier per block (72 MFMAs per wave).
// Fragments come through ds_read_b64_tr_b16 as in conv_wgrad.hip; 128-byte rows, 32-byte slot p of row r at
// p ^ ((r>>1)&3): conflict-free for the 8 rows of a half-wave at ANY row alignment (taps shift the rows).
// Every split writes its own float32 slab region (no atomics); vfd_wgrad_reduce folds them (conv_wgrad.hip).
#include "common.hpp"
#include <stdlib.h>

namespace {

struct WhP {
  const void* S;
  const void* G;
  float* ws;
  int N, D, H, W;          // common spatial grid of S and G (stride 1, "same" padding)
  int Csp, Cs, Cgp;        // padded / logical channels of S, padded channels of G
  int kd, pd;              // depth taps, depth padding
  int nkc;                 // depth-tap classes: kd (GCH 64: one depth tap per work item) or ceil(kd/2) (GCH 32: two)
  int ncols;               // kd*9*Cgp
  int nrt, nct;            // 64-channel tiles of S and of G
  int nhb, nwb;            // pixel blocks per plane: ceil(H/8), ceil(W/16)
  long long nblocks;       // N*D*nhb*nwb
  long long per_split;     // pixel blocks per split
  int nsplit;
  FastDiv fwb, fhb, fd;
};

constexpr int S_ROWS = 128, G_ROWS = 192;                       // G: 10 x 18 = 180 halo rows, padded to 24 DMA instructions
constexpr int S_BYTES = S_ROWS * 128, G_BYTES = G_ROWS * 128;
constexpr int STAGE = S_BYTES + G_BYTES;                        // 40 KiB
constexpr int NSTAGE = 3;

__device__ uint4 g_wh_zero_page[4];

typedef __attribute__((address_space(3))) s16x4 lds_s16x4;

__device__ __forceinline__ bf16x8 tr_pair(uint32_t lo_addr, uint32_t hi_addr) {
  const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(uintptr_t)lo_addr);
  const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(uintptr_t)hi_addr);
  typedef __attribute__((ext_vector_type(8))) short s16x8;
  const s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  return __builtin_bit_cast(bf16x8, v);
}

// GCH = G channels per work item: 64 (128-byte halo rows, ONE depth tap per work item) or 32 (operands with <= 32 channels on the
// gathered side: 64-byte rows, TWO depth taps per work item = two halo planes, so that the accumulator tile stays
// 64 x (2 x 9 x 32) = 64 x 576 and the wave layout is unchanged: wave column wc = (plane = wc >> 1, 16-channel block = wc & 1))
template <int GCH>
__global__ __launch_bounds__(512, 2) void conv_wgrad_halo_kernel(const WhP p) {
  constexpr int GROW = GCH * 2;                 // bytes per halo row
  __shared__ __attribute__((aligned(16))) char smem[NSTAGE * STAGE];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave & 1, wc = wave >> 1;       // S-channel half (32), G-channel block (16)

  // ---- work item: (split, kd, column tile, row tile); splits of one pixel range are adjacent (one L2 reads the pixels once)
  int item = blockIdx.x;
  const int rt = item % p.nrt; item /= p.nrt;
  const int ct = item % p.nct; item /= p.nct;
  const int kdc = item % p.nkc; item /= p.nkc;
  const int kdi = GCH == 64 ? kdc : 2 * kdc;          // first depth tap of this work item
  const int split = item;
  const int r0 = rt * 64, c0 = ct * 64;
  const long long b_beg = (long long)split * p.per_split;
  long long b_end = b_beg + p.per_split;
  if (b_end > p.nblocks) b_end = p.nblocks;
  const int nb = b_beg < b_end ? (int)(b_end - b_beg) : 0;

  const uint32_t smem_base = __builtin_amdgcn_readfirstlane(lds_addr(smem));
  const char* zero = reinterpret_cast<const char*>(g_wh_zero_page);
  const char* Sg = reinterpret_cast<const char*>(p.S);
  const char* Gg = reinterpret_cast<const char*>(p.G);

  // ---- DMA: one instruction = 8 rows x 128 B; lane -> (row in the group, 16-byte piece); the piece's 32-byte slot is
  // swizzled on the SOURCE side: physical slot ps of row r holds logical slot ps ^ ((r>>1)&3), (r>>1)&3 == (lane>>4)&3
  const int drow = lane >> 3;
  const int lpiece = ((((lane & 7) >> 1) ^ ((lane >> 4) & 3)) << 1) | (lane & 1);     // logical 16-byte piece (8 channels)
  // 64-byte G rows: 16 rows per instruction, lane -> (row lane>>2, piece lane&3), 32-byte slot swizzled by (row>>2)&1 = (lane>>4)&1
  const int gpiece = GCH == 64 ? lpiece : (((((lane & 3) >> 1) ^ ((lane >> 4) & 1)) << 1) | (lane & 1));
  const int grow_in_inst = GCH == 64 ? drow : (lane >> 2);
  const bool s_ch_ok = r0 + lpiece * 8 < p.Csp, g_ch_ok = c0 + gpiece * 8 < p.Cgp;
  // position of the block to ISSUE next (2 blocks ahead of the one being computed)
  int in_, id_, ihb, iwb;
  {
    uint32_t t = (uint32_t)b_beg, qw, qh, qd;
    fdivmod(t, p.fwb, t, qw);
    fdivmod(t, p.fhb, t, qh);
    fdivmod(t, p.fd, t, qd);
    in_ = (int)t; id_ = (int)qd; ihb = (int)qh; iwb = (int)qw;
  }
  int issued = 0;
  auto issue_block = [&](int stage) __attribute__((always_inline)) {
    const uint32_t sb = smem_base + stage * STAGE, gb = sb + S_BYTES;
    const bool live = issued < nb;                    // blocks past the range read the zero page (uniform DMA count)
    const int h0 = ihb * 8, w0 = iwb * 16;
    const int gd = id_ + kdi - p.pd;                  // G plane of this depth tap
    // S: 16 instructions, 2 per wave: rows = pixels (h0 + row/16, w0 + row%16)
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const int inst = wave + 8 * k;
      const int row = inst * 8 + drow;
      const int h = h0 + (row >> 4), w = w0 + (row & 15);
      const bool ok = live && s_ch_ok && h < p.H && w < p.W;
      const size_t pix = ((size_t)(in_ * p.D + id_) * p.H + h) * p.W + w;
      const char* src = ok ? Sg + (pix * p.Csp + r0 + lpiece * 8) * 2 : zero;
      dma16_to_lds(src, sb + inst * 1024);
    }
    // G halo: 24 instructions, 3 per wave.  GCH 64: one plane of 192 rows (180 used) x 128 B; GCH 32: two planes (depth taps
    // kdi, kdi+1) of 192 rows x 64 B.  Rows = (hy, wx) of the 10 x 18 halo, pixel (h0-1+hy, w0-1+wx)
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const int inst = wave + 8 * k;
      const int plane = GCH == 64 ? 0 : inst / 12;
      const int row = (GCH == 64 ? inst * 8 : (inst - plane * 12) * 16) + grow_in_inst;
      const int hy = (row * 3641) >> 16;              // row / 18 for row < 192
      const int wx = row - hy * 18;
      const int h = h0 - 1 + hy, w = w0 - 1 + wx;
      const int gdp = gd + plane;
      const bool ok = live && kdi + plane < p.kd && (unsigned)gdp < (unsigned)p.D && g_ch_ok && row < 180 &&
                      (unsigned)h < (unsigned)p.H && (unsigned)w < (unsigned)p.W;
      const size_t pix = ((size_t)(in_ * p.D + gdp) * p.H + h) * p.W + w;
      const char* src = ok ? Gg + (pix * p.Cgp + c0 + gpiece * 8) * 2 : zero;
      dma16_to_lds(src, gb + inst * 1024);
    }
    ++issued;
    if (++iwb == p.nwb) { iwb = 0; if (++ihb == p.nhb) { ihb = 0; if (++id_ == p.D) { id_ = 0; ++in_; } } }
  };

  // ---- fragment addresses (byte offsets inside a stage) for K-step 0: lane = 16 g + 4 q + pp supplies the address of
  // tile row 4g + q, channels cb*16 + 4pp .. +3 (ds_read_b64_tr_b16); K-step ks adds 32 S rows / 2 halo lines
  const int rowl = ((lane >> 4) << 2) + ((lane >> 2) & 3), pp = lane & 3;
  auto tr_addr = [&](int row, int cb) __attribute__((always_inline)) { return row * 128 + ((cb ^ ((row >> 1) & 3)) << 5) + pp * 8; };
  uint32_t a_addr[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) a_addr[i] = tr_addr(rowl, wr * 2 + i);                     // + 2048: rows 16..31 (same swizzle)
  // G fragments: GCH 64: channel block wc of the single plane; GCH 32: channel block wc & 1 of plane wc >> 1 (64-byte rows:
  // 32-byte slot p of row r at p ^ ((r>>2)&1), conflict-free for the 8 rows of a half-wave at any alignment)
  auto tr_addr_g = [&](int row, int cb) __attribute__((always_inline)) {
    return GCH == 64 ? row * 128 + ((cb ^ ((row >> 1) & 3)) << 5) + pp * 8 : row * 64 + ((cb ^ ((row >> 2) & 1)) << 5) + pp * 8;
  };
  const int gplane_rows = GCH == 64 ? 0 : (wc >> 1) * 192;
  const int gcb = GCH == 64 ? wc : (wc & 1);
  uint32_t b_lo[9], b_hi[9];
#pragma unroll
  for (int t = 0; t < 9; ++t) {
    const int row = gplane_rows + (t / 3) * 18 + (t % 3) + rowl;
    b_lo[t] = S_BYTES + tr_addr_g(row, gcb);
    b_hi[t] = S_BYTES + tr_addr_g(row + 18, gcb);
  }

  f32x4 acc[2][9];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int t = 0; t < 9; ++t) acc[i][t] = f32x4{0.f, 0.f, 0.f, 0.f};

  if (nb > 0) {
    issue_block(0);
    issue_block(1);
    int stage = 0;
    for (int b = 0; b < nb; ++b) {
      // this wave's 5 DMAs of block b landed once only the 5 of block b+1 are outstanding; its LDS reads of block b-1 retired
      asm volatile("s_waitcnt vmcnt(5) lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      int nstage = stage + 2;
      if (nstage >= NSTAGE) nstage -= NSTAGE;
      issue_block(nstage);                       // block b+2 -> the stage block b-1 was read from
      const uint32_t sbase = smem_base + stage * STAGE;
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        // S: +32 rows per K-step (swizzle unchanged); G: +36 halo rows per K-step: 128-byte rows: (row>>1)&3 advances by 2 ->
        // slot ^ 2 = byte ^ 64; 64-byte rows: (row>>2)&1 flips -> slot ^ 1 = byte ^ 32
        bf16x8 a[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          const uint32_t lo = sbase + a_addr[i] + ks * 4096;
          a[i] = tr_pair(lo, lo + 2048);
        }
#pragma unroll
        for (int t = 0; t < 9; ++t) {
          const uint32_t x = (ks & 1) ? (GCH == 64 ? 64u : 32u) : 0u;
          const bf16x8 bf = tr_pair(sbase + (b_lo[t] ^ x) + ks * (36 * GROW), sbase + (b_hi[t] ^ x) + ks * (36 * GROW));
#pragma unroll
          for (int i = 0; i < 2; ++i) acc[i][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], bf, acc[i][t], 0, 0, 0);
        }
      }
      if (++stage == NSTAGE) stage = 0;
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  }

  // ---- slab store: ws[split][r][col], col = (kd*9 + t)*Cgp + c;  D[row = 4 (lane>>4) + reg][col = lane & 15]
  float* slab = p.ws + (size_t)split * p.Cs * p.ncols;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      const int c = c0 + gcb * 16 + (lane & 15);
      const int kdw = kdi + (GCH == 64 ? 0 : (wc >> 1));
      const int col = (kdw * 9 + t) * p.Cgp + c;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int rr = r0 + wr * 32 + i * 16 + (lane >> 4) * 4 + r;
        if (rr < p.Cs && c < p.Cgp && kdw < p.kd) slab[(size_t)rr * p.ncols + col] = acc[i][t][r];
      }
    }
}

}  // namespace

static int g_wgrad_halo_mode = -1;    // 0 default rules, 1 never, 2 whenever eligible (tests)

extern "C" int vfd_wgrad_set_halo_mode(int mode) {
  const int prev = g_wgrad_halo_mode < 0 ? (getenv("VFD_NO_WGRAD_HALO") != nullptr ? 1 : 0) : g_wgrad_halo_mode;
  g_wgrad_halo_mode = (mode >= 0 && mode <= 2) ? mode : 0;
  return prev;
}

// Geometry of the halo filter-gradient path for `d`: returns 1 and fills nsplit / bytes when the layer is eligible.
int vfd_wgrad_halo_geom(const vfd_conv_desc* d, int* nsplit, size_t* bytes, WhGeomOut* out) {
  if (g_wgrad_halo_mode < 0) g_wgrad_halo_mode = getenv("VFD_NO_WGRAD_HALO") != nullptr ? 1 : 0;
  if (g_wgrad_halo_mode == 1 || d->dtype != VFD_BF16) return 0;
  if (d->kh != 3 || d->kw != 3 || (d->kd != 1 && d->kd != 3)) return 0;
  if (d->sd != 1 || d->sh != 1 || d->sw != 1 || d->ph != 1 || d->pw != 1 || d->pd != (d->kd - 1) / 2) return 0;
  if (d->Do != d->Di || d->Ho != d->Hi || d->Wo != d->Wi) return 0;
  // S = dy, G = x (regular) or S = x, G = dy (transposed): same grid either way
  const int Cs = d->transposed ? d->Cin : d->Cout, Cg = d->transposed ? d->Cout : d->Cin;
  // gathered side with <= 32 channels: two depth taps per work item instead.  With ONE depth tap (the (1,3,3) factor of a
  // (2+1)D block, mygan's 32 -> 57 layers) the second plane of the item is masked off and half of the accumulator tile idles —
  // those layers are bound by the per-tap gather of conv_wgrad (352 us for 0.4 GB), not by MFMA time, so the halo staging
  // still pays.  Thinner S operands would leave most of the 64-row tile empty
  const bool g32 = Cg <= 32;
  if (Cs < 33 || Cg < 17) return 0;
  const int nrt = (Cs + 63) / 64, nct = g32 ? 1 : (cpad(Cg) + 63) / 64;
  const int nhb = (d->Hi + 7) / 8, nwb = (d->Wi + 15) / 16;
  const long long nblocks = (long long)d->N * d->Di * nhb * nwb;
  const int nkc = g32 ? (d->kd + 1) / 2 : d->kd;
  const int classes = nrt * nct * nkc;
  long long ns = 256 / classes;                     // one workgroup per CU
  if (ns < 1) ns = 1;
  const long long minb = g_wgrad_halo_mode == 2 ? 1 : 8;     // at least 8 blocks per split, or the ring's fill / drain dominates
  if (ns > nblocks / minb) ns = nblocks / minb;
  if (ns < 1 || (g_wgrad_halo_mode != 2 && ns * classes < 128)) return 0;
  if (nblocks >= 0x7fffffffLL) return 0;
  const long long per_split = (nblocks + ns - 1) / ns;
  ns = (nblocks + per_split - 1) / per_split;
  const int ncols = d->kd * 9 * cpad(Cg);
  if (nsplit) *nsplit = (int)ns;
  if (bytes) *bytes = (size_t)ns * Cs * ncols * sizeof(float);
  if (out) {
    out->Cs = Cs; out->Cg = Cg; out->nrt = nrt; out->nct = nct; out->nhb = nhb; out->nwb = nwb;
    out->nblocks = nblocks; out->per_split = per_split; out->nsplit = (int)ns; out->ncols = ncols; out->nkc = nkc; out->g32 = g32 ? 1 : 0;
  }
  return 1;
}

int vfd_wgrad_halo_launch(const vfd_conv_desc* d, const void* x, const void* dy, void* ws, hipStream_t st) {
  WhGeomOut g;
  if (!vfd_wgrad_halo_geom(d, nullptr, nullptr, &g)) return 0;
  WhP p;
  p.S = d->transposed ? x : dy;
  p.G = d->transposed ? dy : x;
  p.ws = reinterpret_cast<float*>(ws);
  p.N = d->N; p.D = d->Di; p.H = d->Hi; p.W = d->Wi;
  p.Cs = g.Cs; p.Csp = cpad(g.Cs); p.Cgp = cpad(g.Cg);
  p.kd = d->kd; p.pd = d->pd; p.nkc = g.nkc;
  p.ncols = g.ncols;
  p.nrt = g.nrt; p.nct = g.nct; p.nhb = g.nhb; p.nwb = g.nwb;
  p.nblocks = g.nblocks; p.per_split = g.per_split; p.nsplit = g.nsplit;
  p.fwb = make_fastdiv((uint32_t)g.nwb); p.fhb = make_fastdiv((uint32_t)g.nhb); p.fd = make_fastdiv((uint32_t)d->Di);
  const long long nwg = (long long)g.nsplit * g.nkc * g.nct * g.nrt;
  if (g.g32) hipLaunchKernelGGL(conv_wgrad_halo_kernel<32>, dim3((unsigned)nwg), dim3(512), 0, st, p);
  else hipLaunchKernelGGL(conv_wgrad_halo_kernel<64>, dim3((unsigned)nwg), dim3(512), 0, st, p);
  return hipGetLastError() == hipSuccess ? 1 : -1;
}
