// Device-side post-processing of the evaluation sweep: the reference thresholds the generator's mask
// at 0.5 and applies a 5 x 5 morphological OPENING per frame on the CPU through cv2 (lib/utils.py:139-152,
// models/mygannet.py:396-397, models/anogan.py:180-181), with a device -> host -> device round trip per test batch.
// Here both steps run on float32 planes [planes][H][W] in HBM (the boundary layout of a 1-channel (N,1,T,H,W) tensor):
//   erode : y = min over the 5 x 5 window, pixels outside the image do not take part (cv2's default border for
//           morphology: BORDER_CONSTANT with +DBL_MAX for erosion, -DBL_MAX for dilation)
//   dilate: y = max over the window
// One thread per pixel, rows of a 5-row window are consecutive 4-byte loads across a wave: HBM / L2 stream.
#include "common.hpp"

namespace {

template <bool ERODE>
__global__ __launch_bounds__(256) void morph5_kernel(const float* __restrict__ x, float* __restrict__ y, long long planes, int H, int W,
                                                     float thr, int binarize) {
  const long long total = planes * H * W;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int w = (int)(i % W);
    const long long t = i / W;
    const int h = (int)(t % H);
    const float* pl = x + (t / H) * (long long)H * W;
    float acc = ERODE ? 3.4e38f : -3.4e38f;
#pragma unroll
    for (int dy = -2; dy <= 2; ++dy) {
      const int hh = h + dy;
      if ((unsigned)hh >= (unsigned)H) continue;
#pragma unroll
      for (int dx = -2; dx <= 2; ++dx) {
        const int ww = w + dx;
        if ((unsigned)ww >= (unsigned)W) continue;
        float v = pl[(long long)hh * W + ww];
        if (binarize) v = v > thr ? 1.f : 0.f;         // threshold() of lib/utils.py:149-152 fused into the first pass
        acc = ERODE ? fminf(acc, v) : fmaxf(acc, v);
      }
    }
    y[i] = acc;
  }
}

}  // namespace

extern "C" int vfd_morph_open5x5(const float* x, float* tmp, float* y, int64_t planes, int H, int W, float threshold, int binarize,
                                 void* stream) {
  VFD_REQUIRE(x && tmp && y && planes > 0 && H > 0 && W > 0, "morph_open5x5: bad arguments");
  VFD_REQUIRE(tmp != y && tmp != x, "morph_open5x5: tmp must be a separate plane buffer");
  const long long total = (long long)planes * H * W;
  long long nb = (total + 255) / 256;
  if (nb > 8192) nb = 8192;
  hipLaunchKernelGGL(morph5_kernel<true>, dim3((unsigned)nb), dim3(256), 0, as_stream(stream), x, tmp, (long long)planes, H, W, threshold, binarize);
  hipLaunchKernelGGL(morph5_kernel<false>, dim3((unsigned)nb), dim3(256), 0, as_stream(stream), (const float*)tmp, y, (long long)planes, H, W, 0.f, 0);
  VFD_CHECK_LAUNCH("morph_open5x5");
  return VFD_OK;
}
