// Thin-channel ends of the image pyramids (bf16): the first convolution of an encoder / discriminator
// (Conv(nc -> ndf, 4, 2, 1), nc <= 8; models/ganomaly.py:39-41, :155-160) and the last transposed convolution of a
// decoder (ConvTranspose(ngf -> nc, 4, 2, 1), nc <= 4; models/ganomaly.py:100-104), together with the data gradients
// that have the same two shapes.  In the implicit-GEMM kernel these layers run 4 (resp. 16) K-steps per workgroup and
// are bound by the per-workgroup fill/drain of the LDS ring (conv_igemm.hip), 5-10x off the HBM time of their tensors.
// Here neither operand is staged: the activation operand goes from global memory straight into MFMA B fragments
// (a pixel's 8-channel granule, or 8 consecutive channels of a 64..128-channel pixel, IS a fragment lane's 16 bytes)
// and the filter lives in LDS / registers for the whole kernel.
//
//   conv_cin8   regular conv, CPAD(Cin) == 8, 3..4 x 3..4 filter rows (x up to 4 depth taps): one K-step = the taps of
//               one filter row (4 taps x 8 channels = 32 = K of v_mfma_f32_16x16x32_bf16; a 3-tap row leaves the 4th
//               K group zero).  Persistent workgroups, filter in LDS in
//               fragment order, 64 channels x 32 pixels per wave.
//   convt_thin  ConvTranspose k4 s2 p1 with Cout <= 4, scatter form: per INPUT pixel the 16 taps x Cout products
//               D[tap, co] = W[:, co, tap] . x[pixel, :] are one small GEMM (16*Cout rows, K = Cin); the strip's D
//               goes to LDS and every output pixel then adds its 4 contributing (input pixel, tap) terms in a fixed
//               order (deterministic, unlike an atomic scatter).  The input is read once instead of once per tap.
#include "common.hpp"
#include <stdlib.h>

namespace {

// ============================================================================================================
// conv_cin8
// ============================================================================================================
struct Cin8P {
  const bf16_t* x;
  const bf16_t* w;      // packed [Cout][kd*kh*kw][8]
  bf16_t* y;
  const float* bias;
  double* stats;        // BatchNorm partial sums [VFD_STATS_REPLICAS][2][Cop] doubles (sum, sum of squares of conv + bias), or null
  int N, Di, Hi, Wi, Do, Ho, Wo, Cout, Cop;
  int kd, kh, kw, sd, sh, sw, pd, ph, pw;
  int act;
  float slope;
  int flip;             // stride-1 transposed convolution = regular convolution with the taps reversed and pad k-1-p;
                        // bit 0 / 1 / 2: reverse the depth / row / column taps (7 = all; 1 = depth only, see the granule remap)
  FastDiv fWo, fHo, fDo;
  long long M;          // output pixels
  int ntiles;           // tiles of CIN8_TILE pixels
};

constexpr int CIN8_NI = 4;                 // 64 output channels per wave
constexpr int CIN8_NJ = 2;                 // 32 pixels per wave
constexpr int CIN8_WAVES = 4;
constexpr int CIN8_TILE = CIN8_WAVES * CIN8_NJ * 16;   // 128 pixels per workgroup and tile

// ROWS = kd*kh when the whole filter's activation fragments fit in registers (<= 9 rows: every load of a tile is issued
// at once, and the NEXT tile's loads are issued before this tile's epilogue, so the load latency is paid behind the
// epilogue instead of once per depth tap), 0 = generic (row group by row group; any kd, kh <= 4)
// NIT = 16-channel blocks of the output a wave computes (1, 2 or 4): thin outputs (14, 21, 32 channels of the (2+1)D stems)
// run a quarter / half of the MFMAs, filter-fragment reads and epilogue arithmetic of the 64-channel form
template <int ROWS, int NIT>
__global__ __launch_bounds__(64 * CIN8_WAVES, ROWS == 0 ? 4 : 3) void conv_cin8_kernel(const Cin8P p) {
  constexpr int NI_ = NIT;
  extern __shared__ __attribute__((aligned(16))) char smem[];   // filter fragments [kd*4][NI][64 lanes][16 B]
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int g = lane >> 4;            // K group of the fragment = tap kx
  const int nks = p.kd * p.kh;        // K-step = one filter row (kz, ky): its <= 4 taps x 8 channels

  // ---- filter -> LDS, in fragment order: lane (m = lane&15, g) of fragment (ks, i) holds w[i*16+m][ks*4+g][0..7]
  for (int f = tid; f < nks * NI_ * 64; f += 64 * CIN8_WAVES) {
    const int l = f & 63, fi = (f >> 6) % NI_, ks = (f >> 6) / NI_;
    // K-step ks = (kz, ky), K group = kx padded to 4 taps: absent taps are zero filter rows
    const int co = fi * 16 + (l & 15), kz = ks / p.kh, ky = ks - kz * p.kh, kx = l >> 4;
    uint4 v = make_uint4(0, 0, 0, 0);
    if (co < p.Cout && kx < p.kw) {
      const int fz = (p.flip & 1) ? p.kd - 1 - kz : kz, fy = (p.flip & 2) ? p.kh - 1 - ky : ky, fx = (p.flip & 4) ? p.kw - 1 - kx : kx;
      v = *reinterpret_cast<const uint4*>(p.w + ((size_t)co * (p.kd * p.kh * p.kw) + (fz * p.kh + fy) * p.kw + fx) * 8);
    }
    *reinterpret_cast<uint4*>(smem + (size_t)f * 16) = v;
  }
  // bias (zero beyond Cout) behind the filter fragments and the per-wave transpose tiles
  float* bias_s = reinterpret_cast<float*>(smem + (size_t)nks * NI_ * 1024 + CIN8_WAVES * CIN8_NJ * 16 * 128);
  if (tid < 16 * NI_) bias_s[tid] = (p.bias != nullptr && tid < p.Cout) ? p.bias[tid] : 0.f;
  // BatchNorm partial sums of this (persistent) workgroup: accumulated in LDS over all its tiles, ONE global atomic per
  // channel and workgroup at the end
  double* red_s = reinterpret_cast<double*>(bias_s + 16 * NI_);       // [2][16 NI], in double (conv_epilogue.hpp: EpiP::stats)
  if (tid < 2 * 16 * NI_) red_s[tid] = 0.0;
  __syncthreads();

  const int cq = g * 4;

  // ---- loads: the 4 filter rows x NJ fragments of one depth tap are issued back to back.  Every address is in bounds
  // (masked lanes read the start of the block and are zeroed on use: no branch around a load, no early wait).  Four
  // waves per SIMD hide the latency; an explicit next-tile prefetch at three measured the same (83 vs 85 us, enc.init).
  long long base[CIN8_NJ];
  int id0[CIN8_NJ], ih0[CIN8_NJ];
  bool colok[CIN8_NJ], pxok[CIN8_NJ];
  auto coords = [&](int tile) {
#pragma unroll
    for (int j = 0; j < CIN8_NJ; ++j) {
      const long long m = (long long)tile * CIN8_TILE + wave * (CIN8_NJ * 16) + j * 16 + (lane & 15);
      uint32_t q = (uint32_t)(m < p.M ? m : 0), ow, oh, od;
      fdivmod(q, p.fWo, q, ow);
      fdivmod(q, p.fHo, q, oh);
      fdivmod(q, p.fDo, q, od);
      const int n = (int)q;
      id0[j] = (int)od * p.sd - p.pd;
      ih0[j] = (int)oh * p.sh - p.ph;
      const int iw = (int)ow * p.sw - p.pw + g;
      pxok[j] = m < p.M;
      colok[j] = m < p.M && (unsigned)iw < (unsigned)p.Wi && g < p.kw;
      base[j] = ((((long long)n * p.Di + id0[j]) * p.Hi + ih0[j]) * p.Wi + iw) * 8;
    }
  };
  constexpr int NB = ROWS == 0 ? 4 : ROWS;
  uint4 braw[NB][CIN8_NJ];
  unsigned bok = 0;
  // generic: the <= 4 filter rows of depth tap kz; ROWS > 0: every row (kz, ky) of the filter
  auto issue = [&](int kz) {
    bok = 0;
#pragma unroll
    for (int r = 0; r < NB; ++r) {
      const int rz = ROWS == 0 ? kz : r / p.kh, ky = ROWS == 0 ? r : r - rz * p.kh;
      if (ROWS != 0 || ky < p.kh) {        // wave-uniform: absent filter rows cost no loads
#pragma unroll
        for (int j = 0; j < CIN8_NJ; ++j) {
          const bool ok = colok[j] && (unsigned)(id0[j] + rz) < (unsigned)p.Di && (unsigned)(ih0[j] + ky) < (unsigned)p.Hi;
          const long long off = ok ? base[j] + ((long long)rz * p.Hi + ky) * p.Wi * 8 : 0;
          braw[r][j] = *reinterpret_cast<const uint4*>(p.x + off);
          bok |= ok ? 1u << (r * CIN8_NJ + j) : 0u;
        }
      }
    }
  };
  auto mma_rows = [&](f32x4 (&acc)[NI_][CIN8_NJ], int kz) {
#pragma unroll
    for (int r = 0; r < NB; ++r) {
      if (ROWS == 0 && r >= p.kh) continue;       // wave-uniform
      const int ks = ROWS == 0 ? kz * p.kh + r : r;
      bf16x8 a[NI_], b[CIN8_NJ];
#pragma unroll
      for (int j = 0; j < CIN8_NJ; ++j) {
        uint4 v = braw[r][j];
        if (!((bok >> (r * CIN8_NJ + j)) & 1u)) v = make_uint4(0, 0, 0, 0);
        b[j] = __builtin_bit_cast(bf16x8, v);
      }
#pragma unroll
      for (int i = 0; i < NI_; ++i)
        a[i] = *reinterpret_cast<const bf16x8*>(smem + ((size_t)(ks * NI_ + i) * 64 + lane) * 16);
#pragma unroll
      for (int i = 0; i < NI_; ++i)
#pragma unroll
        for (int j = 0; j < CIN8_NJ; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
    }
  };

  if (ROWS != 0 && (int)blockIdx.x < p.ntiles) {
    coords(blockIdx.x);
    issue(0);
  }
  for (int tile = blockIdx.x; tile < p.ntiles; tile += gridDim.x) {
    f32x4 acc[NI_][CIN8_NJ];
#pragma unroll
    for (int i = 0; i < NI_; ++i)
#pragma unroll
      for (int j = 0; j < CIN8_NJ; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    bool pxok_t[CIN8_NJ];
    if (ROWS == 0) {
      for (int kz = 0; kz < p.kd; ++kz) {
        if (kz == 0) coords(tile);
        issue(kz);
        mma_rows(acc, kz);
      }
#pragma unroll
      for (int j = 0; j < CIN8_NJ; ++j) pxok_t[j] = pxok[j];
    } else {
      mma_rows(acc, 0);
#pragma unroll
      for (int j = 0; j < CIN8_NJ; ++j) pxok_t[j] = pxok[j];
      if (tile + (int)gridDim.x < p.ntiles) {      // next tile's loads fly behind this tile's epilogue
        coords(tile + gridDim.x);
        issue(0);
      }
    }

    // ---- epilogue: bias + activation, then a transpose through this wave's 4 KiB of LDS so that every lane stores 16
    // bytes and a wave-instruction covers 8 whole pixels (1 KiB contiguous: out pixel == m for a regular convolution)
    // instead of 16 pixels x 32 bytes.  8-byte unit u of pixel row n sits at slot u ^ n: conflict-free for the
    // ds_write_b64 (16 pixels of one unit) and for the ds_read_b128 (8 chunks of one pixel).
    char* ot = smem + (size_t)nks * NI_ * 1024 + wave * (CIN8_NJ * 16 * 128);
    float v[NI_ * CIN8_NJ * 4];
#pragma unroll
    for (int i = 0; i < NI_; ++i) {
      const float4 b4 = *reinterpret_cast<const float4*>(bias_s + i * 16 + cq);
#pragma unroll
      for (int j = 0; j < CIN8_NJ; ++j) {
        float* o = v + (i * CIN8_NJ + j) * 4;
        o[0] = acc[i][j][0] + b4.x; o[1] = acc[i][j][1] + b4.y; o[2] = acc[i][j][2] + b4.z; o[3] = acc[i][j][3] + b4.w;
      }
    }
    if (p.stats != nullptr) {
      // sum / sum of squares of (conv + bias) over this wave's valid pixels: lanes of one K group hold the same 4 channels
      // of 16 different pixels -> shuffle over the 16 lanes, then one LDS atomic per channel and wave
#pragma unroll
      for (int i = 0; i < NI_; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float f1 = 0.f, f2 = 0.f;      // a lane's CIN8_NJ values in float32, the partials in double from here on
#pragma unroll
          for (int j = 0; j < CIN8_NJ; ++j) {
            const float t = pxok_t[j] ? v[(i * CIN8_NJ + j) * 4 + r] : 0.f;
            f1 += t; f2 += t * t;
          }
          f1 = row16_sum(f1);
          f2 = row16_sum(f2);
          const double a1 = (double)f1, a2 = (double)f2;
          if ((lane & 15) == 0) {
            atomicAdd(red_s + i * 16 + cq + r, a1);
            atomicAdd(red_s + 16 * NI_ + i * 16 + cq + r, a2);
          }
        }
    }
    act_apply_n<true>(v, p.act, p.slope);
    if (p.act == VFD_ACT_SIGMOID) {   // pad channels stay zero (filter rows and bias beyond Cout are zero; sigmoid(0) is not)
#pragma unroll
      for (int i = 0; i < NI_; ++i)
#pragma unroll
        for (int j = 0; j < CIN8_NJ; ++j)
#pragma unroll
          for (int r = 0; r < 4; ++r)
            if (i * 16 + cq + r >= p.Cout) v[(i * CIN8_NJ + j) * 4 + r] = 0.f;
    }
#pragma unroll
    for (int j = 0; j < CIN8_NJ; ++j) {
      const int n = lane & 15;
#pragma unroll
      for (int i = 0; i < NI_; ++i) {
        const float* o4 = v + (i * CIN8_NJ + j) * 4;
        uint2 o;
        o.x = pack2bf(o4[0], o4[1]);
        o.y = pack2bf(o4[2], o4[3]);
        *reinterpret_cast<uint2*>(ot + (j * 16 + n) * 128 + (((i * 4 + g) ^ n) << 3)) = o;
      }
    }
    const long long mw = (long long)tile * CIN8_TILE + wave * (CIN8_NJ * 16);
    int lane_o = lane;
    asm volatile("" : "+v"(lane_o));      // keeps the 8 read addresses below from being hoisted out of the tile loop (3 spills)
#pragma unroll
    for (int it = 0; it < CIN8_NJ * 2; ++it) {
      const int px = it * 8 + (lane_o >> 3), c = lane_o & 7, n = px & 15;
      uint4 v = *reinterpret_cast<const uint4*>(ot + px * 128 + ((c ^ (n >> 1)) << 4));
      if (n & 1) v = make_uint4(v.z, v.w, v.x, v.y);
      if (mw + px < p.M && c * 8 < p.Cop)
        *reinterpret_cast<uint4*>(p.y + (mw + px) * p.Cop + c * 8) = v;
    }
  }
  if (p.stats != nullptr) {
    __syncthreads();
    if (tid < 2 * 16 * NI_) {
      const int which = tid / (16 * NI_), c = tid - which * (16 * NI_);
      double* rep = p.stats + (blockIdx.x % VFD_STATS_REPLICAS) * 2 * p.Cop;      // 32-bit offsets: the buffer is a few KB
      if (c < p.Cout) atomicAdd(rep + which * p.Cop + c, red_s[tid]);
    }
  }
}

// ============================================================================================================
// convt_thin
// ============================================================================================================
struct ThinP {
  const bf16_t* x;
  const bf16_t* w;      // packed [Cout][16][Cip]
  bf16_t* y;
  const float* bias;
  int planes, Hi, Wi, Ho, Wo, Cip, Cout, Cop;
  int R, nstrips;       // input rows per strip (plus one halo row on either side)
  int act;
  float slope;
  FastDiv fWo, fWi;
};

// MT = Cout (row m = tap * Cout + co, 16 * Cout rows = MT MFMA row tiles), KS = CPAD(Cin) / 32
template <int MT, int KS>
__global__ __launch_bounds__(256) void convt_thin_kernel(const ThinP p) {
  // D[R+2 rows][Wi+2 columns][16*Cout + 1] float: input rows r0-1 .. r0+R, columns -1 .. Wi; whatever lies outside the
  // image is a zero guard, so the overlap-add below needs no bounds test
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* D = reinterpret_cast<float*>(smem);
  constexpr int COUT = MT;
  constexpr int MROWS = 16 * COUT;
  constexpr int STRIDE = MROWS + 1;      // odd: the pixel-strided accesses spread over all banks
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int g = lane >> 4;
  const int plane = blockIdx.x / p.nstrips;
  const int strip = blockIdx.x - plane * p.nstrips;
  const int r0 = strip * p.R;
  const int row_lo = max(r0 - 1, 0);
  const int row_hi = min(r0 + p.R + 1, p.Hi);
  const int npx = (row_hi - row_lo) * p.Wi;
  const int ntile = (npx + 15) >> 4;
  const int W2 = p.Wi + 2;

  // filter fragments, resident in registers
  bf16x8 a[MT][KS];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    const int m = mt * 16 + (lane & 15);
    const int tap = m / COUT, co = m - tap * COUT;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks)
      a[mt][ks] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(p.w + ((size_t)co * 16 + tap) * p.Cip + ks * 32 + g * 8));
  }

  // zero guards: the two guard columns of every row, and the rows above / below the image
  for (int i = tid; i < (p.R + 2) * 2 * MROWS; i += 256) {
    const int m = i % MROWS, e = i / MROWS;
    D[((e >> 1) * W2 + ((e & 1) ? p.Wi + 1 : 0)) * STRIDE + m] = 0.f;
  }
  if (r0 == 0)
    for (int i = tid; i < p.Wi * MROWS; i += 256) D[(1 + i / MROWS) * STRIDE + i % MROWS] = 0.f;
  for (int rr = row_hi - (r0 - 1); rr < p.R + 2; ++rr)
    for (int i = tid; i < p.Wi * MROWS; i += 256) D[(rr * W2 + 1 + i / MROWS) * STRIDE + i % MROWS] = 0.f;

  const bf16_t* xs = p.x + ((size_t)plane * p.Hi + row_lo) * p.Wi * p.Cip;
  const int slot0 = (row_lo - (r0 - 1)) * W2 + 1;
  // ---- D = W^T x over the strip's pixels, 16 pixels per MFMA column tile; a wave issues the loads of up to TPW of its
  // tiles back to back (one memory latency per pass, not one per tile)
  constexpr int TPW = KS <= 2 ? 6 : 3;
  for (int t0 = 0; t0 < ntile; t0 += 4 * TPW) {
    uint4 braw[TPW][KS];
    int px[TPW];
#pragma unroll
    for (int u = 0; u < TPW; ++u) {
      px[u] = (t0 + wave + 4 * u) * 16 + (lane & 15);
      const int pc = px[u] < npx ? px[u] : 0;
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) braw[u][ks] = *reinterpret_cast<const uint4*>(xs + (size_t)pc * p.Cip + ks * 32 + g * 8);
    }
#pragma unroll
    for (int u = 0; u < TPW; ++u) {
      f32x4 acc[MT];
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) acc[mt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        const bf16x8 b = __builtin_bit_cast(bf16x8, braw[u][ks]);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) acc[mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[mt][ks], b, acc[mt], 0, 0, 0);
      }
      if (px[u] < npx) {
        const uint32_t rl = fdiv((uint32_t)px[u], p.fWi);
        float* drow = D + (slot0 + (int)rl * 2 + px[u]) * STRIDE + g * 4;      // slot = slot0 + rl * W2 + (px - rl * Wi)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
          for (int r = 0; r < 4; ++r) drow[mt * 16 + r] = acc[mt][r];
      }
    }
  }
  __syncthreads();

  // ---- overlap-add: out[oy][ox][co] = sum over the 2 x 2 taps with ky = oy+1 (mod 2), kx = ox+1 (mod 2), in a fixed
  // order; 4 output pixels per thread and pass so that their LDS reads are all in flight together
  const int oy_lo = 2 * r0, oy_hi = min(2 * (r0 + p.R), p.Ho);
  const int nout = (oy_hi - oy_lo) * p.Wo;
  float bias[COUT];
#pragma unroll
  for (int c = 0; c < COUT; ++c) bias[c] = p.bias != nullptr ? p.bias[c] : 0.f;
  bf16_t* yrow = p.y + ((size_t)plane * p.Ho + oy_lo) * p.Wo * p.Cop;
  for (int i0 = 0; i0 < nout; i0 += 1024) {
    float s[4][COUT];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int idx = min(i0 + k * 256 + tid, nout - 1);
      uint32_t oyl, ox;
      fdivmod((uint32_t)idx, p.fWo, oyl, ox);
      const int ky0 = ((int)oyl + 1) & 1, kx0 = ((int)ox + 1) & 1;
      const int rrA = (((int)oyl + 1 - ky0) >> 1) + 1, ccA = (((int)ox + 1 - kx0) >> 1) + 1;
#pragma unroll
      for (int c = 0; c < COUT; ++c) s[k][c] = bias[c];
#pragma unroll
      for (int dy = 0; dy < 2; ++dy)
#pragma unroll
        for (int dx = 0; dx < 2; ++dx) {
          const float* d = D + ((rrA - dy) * W2 + ccA - dx) * STRIDE + ((ky0 + 2 * dy) * 4 + kx0 + 2 * dx) * COUT;
#pragma unroll
          for (int c = 0; c < COUT; ++c) s[k][c] += d[c];
        }
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      act_apply_n<true>(s[k], p.act, p.slope);
      float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int c = 0; c < COUT; ++c) v[c] = s[k][c];
      const int idx = i0 + k * 256 + tid;
      if (idx < nout) store8(yrow + (size_t)idx * p.Cop, v);
    }
  }
}

template <int MT, int KS>
int launch_thin(const ThinP& p, size_t lds, hipStream_t st) {
  static bool attr_set = false;
  if (!attr_set) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(convt_thin_kernel<MT, KS>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 1024) != hipSuccess)
      (void)hipGetLastError();
    attr_set = true;
  }
  hipLaunchKernelGGL((convt_thin_kernel<MT, KS>), dim3((unsigned)(p.planes * p.nstrips)), dim3(256), lds, st, p);
  VFD_CHECK_LAUNCH("convt_thin");
  return 1;
}

bool small_enabled() {
  static const bool off = getenv("VFD_NO_SMALL") != nullptr;
  return !off;
}

}  // namespace

// Returns 1 when the layer was handled (or, with `query`, would be), 0 when it is not one of the two shapes, < 0 on error.
static int small_try_impl(const vfd_conv_desc* d, const void* x, const void* packed, const float* bias, void* y, double* stats,
                          bool query, hipStream_t st, int force_flip);

int vfd_conv_small_try(const vfd_conv_desc* d, const void* x, const void* packed, const float* bias, void* y, double* stats,
                       bool query, hipStream_t st) {
  if (!small_enabled() || d->dtype != VFD_BF16) return 0;
  {
    // Convolutions with NO in-plane taps — (kd,1,1) filters: the pointwise (1x1x1) and the temporal (3,1,1) factors of the
    // (2+1)D blocks, forward and, at stride 1, their data gradients — over 16 / 24 / 32 / 48 / 64 input channels are conv_cin8
    // problems in disguise: a pixel's CPAD(Cin) channels are g = CPAD(Cin)/8 consecutive 8-channel granules, and with no taps
    // along H and W a plane of H x W pixels is one row of H*W*g "pixels" of an 8-channel image, convolved with a kd x 1 x g
    // filter at stride g (g = 6 / 8: a 2 x 3 / 2 x 4 patch of a [2 H W][3|4] image at stride (2, 3|4)).  The packed filter
    // [Cout][kd][CPAD(Cin)] IS [Cout][kd][g][8], and a wave's 16 output pixels read one contiguous run per depth tap.
    // (conv_igemm / conv_halo ran these at 1.5 TB/s: one or three 64-byte K-steps per tile.)  A transposed stride-1 form
    // reverses the DEPTH taps only (flip mask 1) — the granule "taps" are channels.
    const int cip = cpad(d->Cin), g = cip / 8;
    const bool noplane = d->kh == 1 && d->kw == 1 && d->sh == 1 && d->sw == 1 && d->ph == 0 && d->pw == 0 && d->Ho == d->Hi && d->Wo == d->Wi;
    const bool patch = cip == 48 || cip == 64;
    const bool tr_ok = d->transposed && d->sd == 1 && d->Do == d->Di + d->kd - 1 - 2 * d->pd && d->kd - 1 - d->pd >= 0;
    if (noplane && (!d->transposed || tr_ok) && d->kd <= 4 && ((cip > 8 && cip <= 32) || patch) && d->Cout <= 16 * CIN8_NI) {
      const long long plane = (long long)d->Hi * d->Wi;
      if (plane * (patch ? 2 : g) >= 0x7fffffffLL) return 0;
      vfd_conv_desc v = *d;
      v.Cin = 8;
      v.transposed = 0;
      if (d->transposed) v.pd = d->kd - 1 - d->pd;      // the regular form of the stride-1 transposed convolution
      if (!patch) {
        v.Hi = 1; v.Wi = (int)(plane * g); v.Ho = 1; v.Wo = (int)plane;
        v.kw = g; v.sw = g;
      } else {
        v.Hi = (int)(2 * plane); v.Wi = g / 2; v.Ho = (int)plane; v.Wo = 1;
        v.kh = 2; v.kw = g / 2; v.sh = 2; v.sw = g / 2;
      }
      return small_try_impl(&v, x, packed, bias, y, stats, query, st, d->transposed ? 1 : 0);
    }
  }
  return small_try_impl(d, x, packed, bias, y, stats, query, st, -1);
}

// force_flip < 0: the tap reversal follows from the descriptor (a stride-1 transposed convolution over <= 8 channels: all
// three axes); >= 0: the descriptor is already the regular form and force_flip is the tap-reversal mask
static int small_try_impl(const vfd_conv_desc* d, const void* x, const void* packed, const float* bias, void* y, double* stats,
                          bool query, hipStream_t st, int force_flip) {
  const int Cip = cpad(d->Cin), Cop = cpad(d->Cout);
  // a stride-1 transposed convolution without output padding is a regular one with reversed taps and pad k-1-p
  const bool flip = force_flip < 0 && d->transposed && d->sd == 1 && d->sh == 1 && d->sw == 1 && Cip == 8 &&
                    d->Do == d->Di + d->kd - 1 - 2 * d->pd && d->Ho == d->Hi + d->kh - 1 - 2 * d->ph && d->Wo == d->Wi + d->kw - 1 - 2 * d->pw &&
                    d->kd - 1 - d->pd >= 0 && d->kh - 1 - d->ph >= 0 && d->kw - 1 - d->pw >= 0;
  if (!d->transposed || flip) {
    // any filter of <= 4 x 4 x 4 taps over <= 8 input channels, <= 64 output channels (BatchNorm sums optional): the first
    // convolutions of the pyramids, the (1,3,3) / (3,1,1) / 1x1x1 factors of the (2+1)D stems, and the data gradients of
    // last layers with <= 8 output channels (they arrive here as regular convolutions only when their stride is 1 and
    // the caller's transposed flag says so; strided ones stay with conv_igemm)
    if (Cip != 8 || d->kh > 4 || d->kw > 4 || d->Cout > 16 * CIN8_NI || d->kd > 4) return 0;
    const long long M = (long long)d->N * d->Do * d->Ho * d->Wo;
    if (M >= 0x7fffffffLL / CIN8_TILE * CIN8_TILE || M <= 0) return 0;
    if (query) return 1;
    Cin8P p;
    p.x = reinterpret_cast<const bf16_t*>(x); p.w = reinterpret_cast<const bf16_t*>(packed);
    p.y = reinterpret_cast<bf16_t*>(y); p.bias = bias; p.stats = stats;
    p.N = d->N; p.Di = d->Di; p.Hi = d->Hi; p.Wi = d->Wi; p.Do = d->Do; p.Ho = d->Ho; p.Wo = d->Wo;
    p.Cout = d->Cout; p.Cop = Cop;
    p.kd = d->kd; p.kh = d->kh; p.kw = d->kw; p.sd = d->sd; p.sh = d->sh; p.sw = d->sw; p.pd = d->pd; p.ph = d->ph; p.pw = d->pw;
    p.flip = force_flip >= 0 ? force_flip : (flip ? 7 : 0);
    if (flip) { p.pd = d->kd - 1 - d->pd; p.ph = d->kh - 1 - d->ph; p.pw = d->kw - 1 - d->pw; }
    p.act = d->act; p.slope = d->slope;
    p.fWo = make_fastdiv((uint32_t)d->Wo); p.fHo = make_fastdiv((uint32_t)d->Ho); p.fDo = make_fastdiv((uint32_t)d->Do);
    p.M = M;
    p.ntiles = (int)((M + CIN8_TILE - 1) / CIN8_TILE);
    const int ni = d->Cout <= 16 ? 1 : (d->Cout <= 32 ? 2 : 4);
    const size_t lds = (size_t)d->kd * d->kh * ni * 1024 + CIN8_WAVES * CIN8_NJ * 16 * 128 + 16 * ni * sizeof(float) + 2 * 16 * ni * sizeof(double);      // + bias + double sums
    const int blocks = p.ntiles < 1024 ? p.ntiles : 1024;
    const int rows = d->kd * d->kh;
    const int rsel = (rows == 1 || rows == 3 || rows == 4 || rows == 9) ? rows : 0;
#define CIN8_LAUNCH(R_, N_) hipLaunchKernelGGL((conv_cin8_kernel<R_, N_>), dim3((unsigned)blocks), dim3(64 * CIN8_WAVES), lds, st, p)
#define CIN8_ROWS(N_)                                   \
    switch (rsel) {                                     \
      case 1: CIN8_LAUNCH(1, N_); break;                \
      case 3: CIN8_LAUNCH(3, N_); break;                \
      case 4: CIN8_LAUNCH(4, N_); break;                \
      case 9: CIN8_LAUNCH(9, N_); break;                \
      default: CIN8_LAUNCH(0, N_); break;               \
    }
    if (ni == 1) { CIN8_ROWS(1) } else if (ni == 2) { CIN8_ROWS(2) } else { CIN8_ROWS(4) }
#undef CIN8_ROWS
#undef CIN8_LAUNCH
    VFD_CHECK_LAUNCH("conv_cin8");
    return 1;
  }
  // ConvTranspose k4 s2 p1 (no output padding) on 2-D planes, thin output
  if (stats != nullptr) return 0;
  if (d->kd != 1 || d->sd != 1 || d->pd != 0 || d->Do != d->Di) return 0;
  if (d->kh != 4 || d->kw != 4 || d->sh != 2 || d->sw != 2 || d->ph != 1 || d->pw != 1) return 0;
  if (d->Ho != 2 * d->Hi || d->Wo != 2 * d->Wi || d->Cout > 4 || d->Cout == 2) return 0;
  if (Cip != 32 && Cip != 64 && Cip != 128) return 0;
  if ((long long)d->N * d->Di * d->Ho * d->Wo * Cop >= (1ll << 40)) return 0;
  const int stride = 16 * d->Cout + 1;
  static const int forced_r = getenv("VFD_CONVT_THIN_R") ? atoi(getenv("VFD_CONVT_THIN_R")) : 0;
  const size_t budget = 80 * 1000;         // two workgroups per CU
  int R = 0;
  long long best = -1;
  for (int r = 1; r <= d->Hi && r <= 16; ++r) {
    if ((size_t)(r + 2) * (d->Wi + 2) * stride * 4 > budget) break;
    const long long cost = (long long)((d->Hi + r - 1) / r) * (r + 2);
    if (best < 0 || cost < best) { best = cost; R = r; }
  }
  if (forced_r > 0 && (size_t)(forced_r + 2) * (d->Wi + 2) * stride * 4 <= 150 * 1000) R = forced_r < d->Hi ? forced_r : d->Hi;
  if (R == 0) return 0;
  const long long planes = (long long)d->N * d->Di;
  const int nstrips = (d->Hi + R - 1) / R;
  if (planes * nstrips >= 0x7fffffffLL) return 0;
  if (query) return 1;
  ThinP p;
  p.x = reinterpret_cast<const bf16_t*>(x); p.w = reinterpret_cast<const bf16_t*>(packed);
  p.y = reinterpret_cast<bf16_t*>(y); p.bias = bias;
  p.planes = (int)planes; p.Hi = d->Hi; p.Wi = d->Wi; p.Ho = d->Ho; p.Wo = d->Wo;
  p.Cip = Cip; p.Cout = d->Cout; p.Cop = Cop;
  p.R = R; p.nstrips = nstrips;
  p.act = d->act; p.slope = d->slope;
  p.fWo = make_fastdiv((uint32_t)d->Wo);
  p.fWi = make_fastdiv((uint32_t)d->Wi);
  const size_t lds = (size_t)(R + 2) * (d->Wi + 2) * stride * 4;
  const int mt = d->Cout, ks = Cip / 32;
#define THIN(MT_, KS_) if (mt == MT_ && ks == KS_) return launch_thin<MT_, KS_>(p, lds, st);
  THIN(1, 1) THIN(1, 2) THIN(1, 4) THIN(3, 1) THIN(3, 2) THIN(3, 4) THIN(4, 1) THIN(4, 2) THIN(4, 4)
#undef THIN
  return 0;
}
