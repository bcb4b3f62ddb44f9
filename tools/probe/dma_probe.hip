// Micro-benchmark (measurement tool, not product code): sustained global->LDS DMA rate per CU as a function of the
// contiguous run per row (64 B .. 1 KiB), the working-set size (L2-resident or not) and the waves per CU.
// Build: hipcc --offload-arch=gfx950 -O3 -o dma_probe dma_probe.hip ; run: ./dma_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef __attribute__((address_space(3))) void lds_void_t;
typedef __attribute__((address_space(1))) const void glb_void_t;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

// Each wave-instruction moves 1 KiB: 64 lanes x 16 B.  ROWB = contiguous bytes per "row"; rows of one instruction are
// row_stride bytes apart (a gather of 1024/ROWB rows).  Each wave walks `iters` instructions, INFLIGHT outstanding.
template <int ROWB, int INFLIGHT, bool REG>
__global__ __launch_bounds__(512) void probe(const char* __restrict__ src, size_t span_mask, int row_stride, int iters, float* sink) {
  __shared__ __attribute__((aligned(16))) char smem[8 * INFLIGHT * 1024 + 1024];
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  constexpr int CPR = ROWB / 16;
  const int row = lane / CPR, chunk = lane % CPR;
  size_t base = ((size_t)blockIdx.x * 16 + wave) * 7919 * 1024;
  char* dst = smem + wave * INFLIGHT * 1024;
  uint4 accv = {0, 0, 0, 0};
  for (int it = 0; it < iters; ++it) {
    size_t off = (base + (size_t)it * (64 / CPR) * row_stride * 1 + (size_t)row * row_stride + chunk * 16) & span_mask & ~(size_t)15;
    if constexpr (REG) {
      uint4 v = *reinterpret_cast<const uint4*>(src + off);
      accv.x ^= v.x; accv.y ^= v.y; accv.z ^= v.z; accv.w ^= v.w;
    } else {
      __builtin_amdgcn_global_load_lds((glb_void_t*)(src + off), (lds_void_t*)(dst + (it % INFLIGHT) * 1024), 16, 0, 0);
      constexpr int N = INFLIGHT - 1;
      __builtin_amdgcn_s_waitcnt((N & 0xF) | (0x7 << 4) | (0xF << 8) | ((N >> 4) << 14));
    }
  }
  __builtin_amdgcn_s_waitcnt(0 | (0x7 << 4) | (0xF << 8));
  __syncthreads();
  if (sink != nullptr && threadIdx.x == 0 && blockIdx.x == 1 << 30) sink[0] = smem[lane] + accv.x + accv.y + accv.z + accv.w;
}

template <int ROWB, int INFLIGHT, bool REG>
void run(const char* src, size_t span, int row_stride, int blocks, int nthreads, const char* tag) {
  const int iters = 4096;
  hipEvent_t a, b;
  CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  probe<ROWB, INFLIGHT, REG><<<blocks, nthreads>>>(src, span - 1, row_stride, 64, nullptr);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(a));
  probe<ROWB, INFLIGHT, REG><<<blocks, nthreads>>>(src, span - 1, row_stride, iters, nullptr);
  CK(hipEventRecord(b));
  CK(hipEventSynchronize(b));
  float ms; CK(hipEventElapsedTime(&ms, a, b));
  const double bytes = (double)blocks * (nthreads / 64) * iters * 1024.0;
  const double tbs = bytes / (ms * 1e-3) / 1e12;
  printf("%-6s rowB %4d stride %6d span %5zu MiB blocks %4d waves/blk %2d inflight %2d : %7.2f TB/s  = %5.1f B/clk/CU (2.4 GHz, 256 CU)\n",
         tag, ROWB, row_stride, span >> 20, blocks, nthreads / 64, INFLIGHT, tbs, tbs * 1e12 / 256 / 2.4e9);
}

int main() {
  const size_t big = (size_t)1 << 30;
  char* src; CK(hipMalloc(&src, big)); CK(hipMemset(src, 1, big));
  // (1) contiguous run per row, L2-resident vs streaming working set
  for (size_t span : {(size_t)2 << 20, (size_t)1 << 30}) {
    run<64, 6, false>(src, span, 512, 512, 512, "dma");
    run<128, 6, false>(src, span, 512, 512, 512, "dma");
    run<256, 6, false>(src, span, 512, 512, 512, "dma");
    run<1024, 6, false>(src, span, 1024, 512, 512, "dma");
    run<64, 12, false>(src, span, 512, 512, 512, "dma");
    run<64, 6, true>(src, span, 512, 512, 512, "reg");
    run<1024, 6, true>(src, span, 1024, 512, 512, "reg");
  }
  // (2) row stride (filter rows are Kw*2 bytes apart, e.g. 8 KiB: no channel-conflict effect)
  for (int stride : {128, 1024, 8192, 8192 + 256}) {
    run<64, 6, false>(src, (size_t)4 << 20, stride, 512, 512, "dma");
    run<128, 6, false>(src, (size_t)4 << 20, stride, 512, 512, "dma");
  }
  return 0;
}
