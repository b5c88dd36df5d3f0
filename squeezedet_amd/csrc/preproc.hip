// Image pre-processing of the demo / eval callers on the GPU (reference src/demo.py:186-190,
// src/dataset/imdb.py:101-118):  im = imread(f).astype(float32);  im = cv2.resize(im, (W, H));
// input = im - mc.BGR_MEANS  -> one kernel: uint8 BGR HWC in, NHWC network input (storage dtype) out.
//
// cv2.resize, INTER_LINEAR, float32 (the default interpolation): source coordinate of destination x is
//   fx = (float)((x + 0.5) * scale_x - 0.5) with scale_x = (double)Ws / Wd -- the coordinate is formed in DOUBLE and
//   rounded once to float32, as cv::resize does --;  sx = floor(fx);  fx -= sx;  sx < 0 -> (sx, fx) = (0, 0);
//   sx >= Ws - 1 -> (sx, fx) = (Ws - 1, 0)           (likewise in y),
// rows are interpolated horizontally first (S[sx]*(1-fx) + S[sx+1]*fx), then vertically
// (r0*(1-fy) + r1*fy), all in float32.  The same order is kept here, without FMA contraction.
#include "common.h"

namespace sqdet {

template <typename T>
__global__ __launch_bounds__(256) void preprocess_kernel(const unsigned char* __restrict__ src, T* __restrict__ dst, int N,
                                                         int Hs, int Ws, int Hd, int Wd, float m0, float m1, float m2) {
  const double scale_x = (double)Ws / (double)Wd, scale_y = (double)Hs / (double)Hd;
  const size_t total = (size_t)N * Hd * Wd;
  for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
    const int x = (int)(idx % Wd);
    const int y = (int)((idx / Wd) % Hd);
    const int n = (int)(idx / ((size_t)Wd * Hd));
    float fx = (float)((x + 0.5) * scale_x - 0.5), fy = (float)((y + 0.5) * scale_y - 0.5);
    int sx = (int)floorf(fx), sy = (int)floorf(fy);
    fx -= sx; fy -= sy;
    if (sx < 0) { sx = 0; fx = 0.f; }
    if (sx >= Ws - 1) { sx = Ws - 1; fx = 0.f; }
    if (sy < 0) { sy = 0; fy = 0.f; }
    if (sy >= Hs - 1) { sy = Hs - 1; fy = 0.f; }
    const int sx1 = sx + 1 < Ws ? sx + 1 : sx, sy1 = sy + 1 < Hs ? sy + 1 : sy;
    const unsigned char* r0 = src + ((size_t)n * Hs + sy) * Ws * 3;
    const unsigned char* r1 = src + ((size_t)n * Hs + sy1) * Ws * 3;
    const float ax0 = 1.f - fx, ay0 = 1.f - fy;
    const float mean[3] = {m0, m1, m2};
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const float h0 = (float)r0[sx * 3 + c] * ax0 + (float)r0[sx1 * 3 + c] * fx;
      const float h1 = (float)r1[sx * 3 + c] * ax0 + (float)r1[sx1 * 3 + c] * fx;
      dst[idx * 3 + c] = (T)((h0 * ay0 + h1 * fy) - mean[c]);
    }
  }
}

}  // namespace sqdet

extern "C" int sqdet_preprocess_bgr(const uint8_t* src_bgr_u8, void* dst, int n, int src_h, int src_w, int dst_h,
                                    int dst_w, float mean_b, float mean_g, float mean_r, int dtype,
                                    sqdet_stream_t stream) {
  using namespace sqdet;
  SQDET_REQUIRE(src_bgr_u8 && dst, "preprocess_bgr: null pointer");
  SQDET_REQUIRE(n > 0 && src_h > 0 && src_w > 0 && dst_h > 0 && dst_w > 0, "preprocess_bgr: bad dims");
  SQDET_REQUIRE(dtype == SQDET_F16 || dtype == SQDET_F32, "preprocess_bgr: bad dtype %d", dtype);
  const size_t total = (size_t)n * dst_h * dst_w;
  size_t blocks = (total + 255) / 256;
  if (blocks > 8192) blocks = 8192;
  if (dtype == SQDET_F16)
    hipLaunchKernelGGL(preprocess_kernel<f16>, dim3((unsigned)blocks), dim3(256), 0, as_stream(stream), src_bgr_u8, (f16*)dst,
                       n, src_h, src_w, dst_h, dst_w, mean_b, mean_g, mean_r);
  else
    hipLaunchKernelGGL(preprocess_kernel<float>, dim3((unsigned)blocks), dim3(256), 0, as_stream(stream), src_bgr_u8,
                       (float*)dst, n, src_h, src_w, dst_h, dst_w, mean_b, mean_g, mean_r);
  SQDET_CHECK_HIP(hipGetLastError());
  return SQDET_OK;
}
