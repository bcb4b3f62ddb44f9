// Epilogue shared by the MFMA convolution kernels (conv_igemm.hip, conv_halo.hip): bias, BatchNorm partial sums,
// activation, optional gradient hand-over (MulP: multiply by act', BatchNorm backward sums), channels-last store.
//
// Accumulator layout (both kernels): wave w owns channels [wave_c0, +NI*16) x tile rows (pixels) [wave_p0, +NJ*16) with
// wave_c0 = (w % WAVES_C) * NI*16, wave_p0 = (w / WAVES_C) * NJ*16; acc[i][j][r] = channel wave_c0 + 16 i + 4 (lane>>4) + r
// of tile row wave_p0 + 16 j + (lane & 15)   (v_mfma_f32_16x16x*: channels are the MFMA row dimension).
// `out_offset(tile_row)` -> element offset of that pixel's channel 0 in y, or -1 (row outside the tensor);
// `row_valid(tile_row)` == (out_offset(tile_row) >= 0), in a form that needs no address arithmetic.
// The staging LDS (`smem`, RING_BYTES, all DMAs drained by the caller) is reused.
#pragma once
#include "common.hpp"
#include <type_traits>

struct EpiP {
  void* y;
  const float* bias;
  double* stats;         // [VFD_STATS_REPLICAS][2][Cop] DOUBLES or null: per-channel sum / sum of squares of (conv + bias) for the
                         // BatchNorm that follows.  The variance is later formed as E[x^2] - mean^2, which in float32 sums loses
                         // |mean|/sigma squared digits (a bias-dominated layer on a sparse input - mygan's SDisc on the 0/1 mask -
                         // has |mean|/sigma ~ 30: measured 5e-3 on its loss): see the shifted sums below
  int Cop, Cout;
  int act;
  float slope;
  MulP mul;              // mul.src non-null: gradient hand-over (common.hpp), mul.src has y's shape
  float oscale;          // SCALED kernels (fp8 operands): the accumulator is multiplied by this first (1 / (scale_x * scale_w))
};

// LDS bytes of the statistics block behind the output image: shifted sums [2][TILE_C][WAVES_P] floats + shifts [TILE_C] + the
// valid-row counter (the BatchNorm hand-over: [2][TILE_C][WAVES_P] floats)
constexpr int epi_red_bytes(int tile_c, int waves_p, bool bn) {
  return bn ? 2 * tile_c * waves_p * 4 : (2 * waves_p + 1) * tile_c * 4 + 16;
}

template <typename T, int WAVES_C, int WAVES_P, int NI, int NJ, int RING_BYTES, bool BN, bool SCALED, typename OutOff, typename RowValid>
__device__ __forceinline__ void conv_epilogue(char* smem, f32x4 (&acc)[NI][NJ], const EpiP& p, int n0, int stats_replica,
                                              OutOff out_offset, RowValid row_valid) {
  constexpr int TILE_C = WAVES_C * NI * 16;
  constexpr int TILE_P = WAVES_P * NJ * 16;
  constexpr int NWAVES = WAVES_C * WAVES_P;
  // (fp8 tiles: the lane index is re-derived from the exec mask count instead of threadIdx.x, which would otherwise be the
  // one value that has to survive the main loop in a VGPR — spilled at the 128-register cap of the 16-wave tile)
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int lane = SCALED ? (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)) : (int)(threadIdx.x & 63);
  const int tid = SCALED ? wave * 64 + lane : (int)threadIdx.x;
  const int wave_c0 = (wave % WAVES_C) * (NI * 16);
  const int wave_p0 = (wave / WAVES_C) * (NJ * 16);
  const int cq = (lane >> 4) * 4;
  // (channel-tile outer loop: only 4 bias values and 8 statistic partials are live at a time; the activation is
  // dispatched ONCE around the loops: a per-value switch is replicated, branches included, in every unrolled copy)
  T* __restrict__ yg = reinterpret_cast<T*>(p.y);
  // BN (a kernel variant of its own: as a runtime branch its registers spill the plain kernels): the BatchNorm hand-over
  // of MulP.  It rides on the statistics machinery below with other summands: per channel, the sums of g = y * act'(z)
  // and of g * xh instead of y and y^2, added into mul.bn_sums.
  const bool want_stats = BN || p.stats != nullptr;
  // bf16 tiles of >= 64 channels leave through LDS: the MFMA layout gives a lane 4 channels (8 bytes) of one pixel, i.e.
  // 16 pixels x 32 bytes per store instruction; transposed through the (now idle) staging ring every lane stores 16
  // bytes and an instruction covers whole pixel rows of the tile (128..512 contiguous bytes each).
  typedef typename std::conditional<BN, float, double>::type S;      // accumulation type of the per-channel sums (BN hand-over: float)
  // statistics (ATOM): SHIFTED sums.  Per channel the workgroup picks a shift c (the value of its first tile row) and adds up
  // d = t - c and d^2 — small numbers whatever |mean| / sigma is — in float32: registers, 16-lane shuffles, then ONE SLOT per
  // channel and pixel-wave in LDS (no LDS float atomics: their arrival order would make the sums, and through BatchNorm and
  // Adam the whole run, differ from launch to launch; a first version did and test_graph_replay_after_reinit_d caught it).
  // ONE thread per channel then adds the slots in wave order and forms the raw sums in double,
  //   sum t = S1 + n c,   sum t^2 = S2 + 2 c S1 + n c^2      (n = valid rows of the tile),
  // and adds them to the replica row (global double atomics), from where everything stays double.  (Round 3 history: float32
  // sums of t, t^2 lost (|mean|/sigma)^2 digits: 5e-3 on a loss at a ratio of 30; double from the wave level on held a ratio
  // of 100 at the price of double LDS atomics; this form holds 1e3 and beyond — the test pins it — for less work.)
  // The BN hand-over keeps its float slot per wave (its two passes accumulate into their own slots).
  constexpr bool ATOM = !BN;
  constexpr int RED_BYTES = epi_red_bytes(TILE_C, WAVES_P, BN);
    // the tile goes through in NH passes (half of every wave's pixel sub-tiles each) when it does not fit in one (256 x 256: 128 KiB)
  constexpr int NH = (TILE_P * TILE_C * 2 + RED_BYTES + TILE_P * 8 <= RING_BYTES) ? 1
                     : ((NJ % 2 == 0 && TILE_P * TILE_C + RED_BYTES + TILE_P * 8 <= RING_BYTES) ? 2 : 0);
  constexpr bool VIA_LDS = sizeof(T) == 2 && TILE_C >= 64 && NH > 0;
  constexpr int HALF_P = VIA_LDS ? TILE_P / NH : TILE_P;
  constexpr int OUT_BYTES = VIA_LDS ? HALF_P * TILE_C * 2 : 0;
  static_assert(OUT_BYTES + RED_BYTES + TILE_P * 8 <= RING_BYTES, "epilogue LDS exceeds the staging LDS");
  static_assert(!BN || VIA_LDS, "the BatchNorm hand-over is built on the row-store epilogue (bf16, >= 64-channel tiles)");
  // BatchNorm partial sums: lanes -> wave (shuffles) -> workgroup (LDS) -> ONE float atomic per channel and workgroup
  // into one of VFD_STATS_REPLICAS replica rows (spreads the contention of thousands of workgroups adding into the
  // same 2*Cout addresses; bn_from_sums folds the replicas).
  S* red = reinterpret_cast<S*>(smem + OUT_BYTES);     // BN hand-over: [2][TILE_C][WAVES_P] floats
  float* redf = reinterpret_cast<float*>(smem + OUT_BYTES);      // ATOM: [2][TILE_C][WAVES_P] shifted sums | [TILE_C] shifts | valid-row count
  float* cshift = redf + 2 * TILE_C * WAVES_P;
  int* nvalid = reinterpret_cast<int*>(cshift + TILE_C);
  long long* orow = reinterpret_cast<long long*>(smem + OUT_BYTES + RED_BYTES);   // [TILE_P] output offset of a tile row, or -1
  long long opix[NJ];     // direct path: output pixel offset in elements (pixel * Cop), or -1
  bool pvalid[NJ];
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    const int row = wave_p0 + j * 16 + (lane & 15);
    pvalid[j] = row_valid(row);      // cheap form of out_offset(row) >= 0 (statistics mask)
    if constexpr (!VIA_LDS) opix[j] = out_offset(row);
  }
  if (VIA_LDS || want_stats) __syncthreads();      // every wave is done reading the last stage
  if constexpr (VIA_LDS) {
    for (int r = tid; r < TILE_P; r += 64 * NWAVES) orow[r] = out_offset(r);
    if constexpr (BN) __syncthreads();      // the accumulator pass below already needs the row offsets
  }
  if constexpr (ATOM) {
    if (want_stats) {
      if (tid == 0) *nvalid = 0;
      if (wave / WAVES_C == 0 && (lane & 15) == 0) {      // the shift of a channel: its value in tile row 0 (any value would do)
#pragma unroll
        for (int i = 0; i < NI; ++i)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int c = n0 + wave_c0 + i * 16 + cq + r;
            const float b = (p.bias != nullptr && c < p.Cout) ? p.bias[c] : 0.f;
            cshift[wave_c0 + i * 16 + cq + r] = (SCALED ? acc[i][0][r] * p.oscale : acc[i][0][r]) + b;
          }
      }
      if (wave % WAVES_C == 0) {      // valid tile rows: lanes 0..15 hold a sub-tile's 16 rows
        int nv = 0;
#pragma unroll
        for (int j = 0; j < NJ; ++j) nv += __popcll(__ballot(pvalid[j]) & 0xffffull);
        if (lane == 0) atomicAdd(nvalid, nv);
      }
      __syncthreads();      // shifts and count in place before the first wave reads them
    }
  }
  // pass h of NH emits the pixel sub-tiles j in [h*NJ/NH, (h+1)*NJ/NH) of EVERY wave (so that no wave carries its whole
  // accumulator tile across a store phase); the statistics of all sub-tiles are taken in pass 0
  auto body = [&](auto actf, auto hc) {
    constexpr int H = decltype(hc)::value;
    constexpr int JN = VIA_LDS ? NJ / NH : NJ;
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const int c = n0 + wave_c0 + i * 16 + cq;
      float b4[4] = {0.f, 0.f, 0.f, 0.f};
      if (p.bias != nullptr) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if ((c + r) < p.Cout) b4[r] = p.bias[c + r];
      }
      // (a wave's 16 lanes x NJ pixels are added up in float32 — 64 terms, as in rounds 1-2: double shuffles cost 5 % of the
      // 128-channel tile's launch — and the per-wave partials go to double: LDS fold, atomics, the BatchNorm fold.  Relative
      // error of the variance ~ 2.4e-7 (|mean|/sigma)^2 / sqrt(number of wave partials): 3e-6 at a ratio of 30 over 400k pixels)
      float s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};
      float c4[4] = {0.f, 0.f, 0.f, 0.f};
      if constexpr (ATOM) {
        if (H == 0 && want_stats) {
          const float4 cv = *reinterpret_cast<const float4*>(cshift + wave_c0 + i * 16 + cq);
          c4[0] = cv.x; c4[1] = cv.y; c4[2] = cv.z; c4[3] = cv.w;
        }
      }
      // BatchNorm hand-over: xh = x * ka + kb, z = xh * kg + kt for this lane's 4 channels; the producer BatchNorm's input
      // x at this lane's (pixel, 4 channels) positions, all loads of the pass in flight before the first use
      float ka[4], kb[4], kg[4], kt[4];
      uint2 xv[NJ];
      if constexpr (BN) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const bool ok = (c + r) < p.Cout;
          ka[r] = ok ? p.mul.bn_rstd[c + r] : 0.f;
          kb[r] = ok ? -p.mul.bn_mean[c + r] * ka[r] : 0.f;
          kg[r] = ok ? (p.mul.bn_gamma != nullptr ? p.mul.bn_gamma[c + r] : 1.f) : 0.f;
          kt[r] = (ok && p.mul.bn_beta != nullptr) ? p.mul.bn_beta[c + r] : 0.f;
        }
#pragma unroll
        for (int j = H * JN; j < (H + 1) * JN; ++j) {
          const long long off = orow[wave_p0 + j * 16 + (lane & 15)];
          xv[j] = make_uint2(0u, 0u);
          if (off >= 0 && c < p.Cop) xv[j] = *reinterpret_cast<const uint2*>(reinterpret_cast<const bf16_t*>(p.mul.src) + off + c);
        }
      }
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        const bool emit = j >= H * JN && j < (H + 1) * JN;     // compile-time after unrolling
        if (!emit && (H != 0 || BN)) continue;
        float v[4];
        if constexpr (BN) {
          const float xf[4] = {bf2f((bf16_t)(xv[j].x & 0xffffu)), bf2f((bf16_t)(xv[j].x >> 16)), bf2f((bf16_t)(xv[j].y & 0xffffu)), bf2f((bf16_t)(xv[j].y >> 16))};
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float xh = xf[r] * ka[r] + kb[r];
            const float t = (SCALED ? acc[i][j][r] * p.oscale : acc[i][j][r]) * act_grad_from_in(xh * kg[r] + kt[r], p.mul.act, p.mul.slope);
            if (pvalid[j]) { s1[r] += t; s2[r] += t * xh; }
            v[r] = ((c + r) < p.Cout) ? t : 0.f;
          }
        } else {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float t = (SCALED ? acc[i][j][r] * p.oscale : acc[i][j][r]) + b4[r];
            if (H == 0 && want_stats && pvalid[j]) { const float dshift = t - c4[r]; s1[r] += dshift; s2[r] = __builtin_fmaf(dshift, dshift, s2[r]); }      // (explicit fma: the build runs with -ffp-contract=off)
            v[r] = ((c + r) < p.Cout) ? actf(t) : 0.f;   // pad channels stay zero (sigmoid(0) != 0)
          }
        }
        if (!emit) continue;
        if constexpr (VIA_LDS) {
          // 8-byte unit u of tile row (pixel) with row & 15 == n sits at slot u ^ n: the 16 rows of one ds_write_b64
          // group land on 16 different bank pairs, and a pixel's 16-byte chunk c is found whole at c ^ (n >> 1)
          const int n = lane & 15;
          uint2 o;
          if constexpr (SCALED) { o.x = pack2bf_2cvt(v[0], v[1]); o.y = pack2bf_2cvt(v[2], v[3]); }
          else { o.x = pack2bf(v[0], v[1]); o.y = pack2bf(v[2], v[3]); }
          *reinterpret_cast<uint2*>(smem + ((wave / WAVES_C) * (JN * 16) + (j - H * JN) * 16 + n) * (TILE_C * 2) + ((((wave_c0 + i * 16 + cq) >> 2) ^ n) << 3)) = o;
        } else {
          if (opix[j] >= 0 && c < p.Cop) {
            T* dst = yg + opix[j] + c;
            if (p.mul.src != nullptr) {
              const T* ms = reinterpret_cast<const T*>(p.mul.src) + opix[j] + c;
#pragma unroll
              for (int r = 0; r < 4; ++r) v[r] *= act_grad_from_out(Elem<T>::ld(ms + r), p.mul.act, p.mul.slope);
            }
            if constexpr (sizeof(T) == 2) {
              uint2 o; o.x = pack2bf(v[0], v[1]); o.y = pack2bf(v[2], v[3]);
              *reinterpret_cast<uint2*>(dst) = o;
            } else {
              *reinterpret_cast<float4*>(dst) = make_float4(v[0], v[1], v[2], v[3]);
            }
          }
        }
      }
      if ((H == 0 || BN) && want_stats) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float af = s1[r], bf = s2[r];
          af = row16_sum(af);
          bf = row16_sum(bf);
          S a = (S)af, b = (S)bf;
          if ((lane & 15) == 0) {
            const int cl = wave_c0 + i * 16 + cq + r;      // channel within the tile
            if constexpr (ATOM) {
              redf[cl * WAVES_P + (wave / WAVES_C)] = af;      // this wave's slot
              redf[(TILE_C + cl) * WAVES_P + (wave / WAVES_C)] = bf;
            } else {
              if (H != 0) {        // the hand-over sums are taken as the sub-tiles are emitted: pass 1 adds to pass 0 (same lane)
                a += red[cl * WAVES_P + (wave / WAVES_C)];
                b += red[(TILE_C + cl) * WAVES_P + (wave / WAVES_C)];
              }
              red[cl * WAVES_P + (wave / WAVES_C)] = a;
              red[(TILE_C + cl) * WAVES_P + (wave / WAVES_C)] = b;
            }
          }
        }
      }
    }
  };
  // store phase of pass h: the rows of pixel half h leave LDS as whole pixel rows, 16 bytes per lane
  auto store_rows = [&](int h) {
    constexpr int CPRW = TILE_C / 8;          // 16-byte chunks per tile row
    constexpr int RPW = 64 / CPRW;            // tile rows per wave-instruction
    constexpr int NIT = HALF_P / RPW;         // wave-instructions per pass
    constexpr int U = NH == 1 ? 4 : 2;        // rows in flight per lane (two passes: half the waves still hold their tile)
    const int c = lane % CPRW;
    const bool cok = n0 + c * 8 < p.Cop;
    const bf16_t* ms = BN ? nullptr : reinterpret_cast<const bf16_t*>(p.mul.src);     // BN: multiplied in the accumulator pass
    for (int it0 = wave; it0 < NIT; it0 += U * NWAVES) {
      long long off[U];
      uint4 v[U], mv[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int it = it0 + u * NWAVES;
        const int lr = it * RPW + lane / CPRW;               // row of this pass's LDS image = (pixel wave, sub-tile, pixel)
        const int tr = (lr / (HALF_P / WAVES_P)) * (TILE_P / WAVES_P) + h * (HALF_P / WAVES_P) + lr % (HALF_P / WAVES_P);
        off[u] = (it < NIT && cok) ? orow[tr] : -1;
      }
      if (ms != nullptr) {      // the producer's activation output at the same positions: all U loads issued together
#pragma unroll
        for (int u = 0; u < U; ++u) mv[u] = *reinterpret_cast<const uint4*>(ms + (off[u] >= 0 ? off[u] + n0 + c * 8 : 0));
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int it = it0 + u * NWAVES;
        const int row = (it < NIT ? it : wave % NIT) * RPW + lane / CPRW, n = row & 15;
        v[u] = *reinterpret_cast<const uint4*>(smem + row * (TILE_C * 2) + ((c ^ (n >> 1)) << 4));
        if (n & 1) v[u] = make_uint4(v[u].z, v[u].w, v[u].x, v[u].y);
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        if (ms != nullptr) {
          float f[8], m[8];
          load8(reinterpret_cast<const bf16_t*>(&v[u]), f);
          load8(reinterpret_cast<const bf16_t*>(&mv[u]), m);
#pragma unroll
          for (int k = 0; k < 8; ++k) f[k] *= act_grad_from_out(m[k], p.mul.act, p.mul.slope);
          store8(reinterpret_cast<bf16_t*>(&v[u]), f);
        }
        if (off[u] >= 0) *reinterpret_cast<uint4*>(yg + off[u] + n0 + c * 8) = v[u];
      }
    }
  };
  auto dispatch_body = [&](auto hc) {
    const float slope = p.slope;
    constexpr bool FAST = sizeof(T) == 2;     // bf16 output: v_exp/v_rcp forms are exact to far below half an ulp
    switch (p.act) {
      case VFD_ACT_LRELU: body([slope](float t) { return t > 0.f ? t : t * slope; }, hc); break;
      case VFD_ACT_SIGMOID: body([](float t) { return FAST ? fast_sigmoid(t) : 1.f / (1.f + __expf(-t)); }, hc); break;
      case VFD_ACT_TANH: body([](float t) { return FAST ? fast_tanh(t) : tanhf(t); }, hc); break;
      default: body([](float t) { return t; }, hc); break;
    }
  };
  dispatch_body(std::integral_constant<int, 0>());
  if constexpr (!VIA_LDS) {
    if (want_stats) __syncthreads();
  } else {
    __syncthreads();
    store_rows(0);
    if constexpr (NH == 2) {
      __syncthreads();
      dispatch_body(std::integral_constant<int, 1>());
      __syncthreads();
      store_rows(1);
    }
  }
  if (want_stats) {
    if constexpr (ATOM) {
      double* rep = p.stats + (size_t)(stats_replica % VFD_STATS_REPLICAS) * 2 * p.Cop;
      const double n = (double)*nvalid;
      for (int cl = tid; cl < TILE_C; cl += 64 * NWAVES) {
        if (n0 + cl < p.Cout) {
          double s1 = 0.0, s2 = 0.0;
#pragma unroll
          for (int w = 0; w < WAVES_P; ++w) { s1 += (double)redf[cl * WAVES_P + w]; s2 += (double)redf[(TILE_C + cl) * WAVES_P + w]; }
          const double c = (double)cshift[cl];
          atomicAdd(rep + n0 + cl, s1 + n * c);                                  // global_atomic_add_f64: two per channel and workgroup
          atomicAdd(rep + p.Cop + n0 + cl, s2 + 2.0 * c * s1 + n * c * c);
        }
      }
    } else {
      S* rep = p.mul.bn_sums + (size_t)(stats_replica % VFD_STATS_REPLICAS) * 2 * p.Cop;
      for (int t = tid; t < 2 * TILE_C; t += 64 * NWAVES) {
        const int which = t / TILE_C, cl = t - which * TILE_C;
        S v = 0;
#pragma unroll
        for (int w = 0; w < WAVES_P; ++w) v += red[t * WAVES_P + w];
        if (n0 + cl < p.Cout) atomicAdd(rep + which * p.Cop + n0 + cl, v);      // global_atomic_add_f32: one per channel and workgroup
      }
    }
  }
}
