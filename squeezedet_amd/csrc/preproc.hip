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

// One 64-thread workgroup = 256 consecutive pixels of one destination row (4 per thread): the row's two source rows and
// the vertical weight are workgroup-uniform, a thread's 4 pixels leave as 24 (fp16) / 48 (fp32) contiguous bytes, and a
// source pixel pair (sx, sx + 1) is ONE unaligned 8-byte load instead of six byte loads.  (The first version -- one thread
// per pixel, 64-bit index arithmetic, byte loads, 2-byte stores -- ran at 0.23 of the HBM peak.)
constexpr int PPX = 4;   // destination pixels per thread

template <typename T>
__global__ __launch_bounds__(64) void preprocess_kernel(const unsigned char* __restrict__ src, T* __restrict__ dst, int N,
                                                        int Hs, int Ws, int Hd, int Wd, float m0, float m1, float m2,
                                                        size_t src_bytes) {
  const double scale_x = (double)Ws / (double)Wd, scale_y = (double)Hs / (double)Hd;
  const int row = blockIdx.y;                     // n * Hd + y
  const int n = row / Hd, y = row - n * Hd;
  float fy = (float)((y + 0.5) * scale_y - 0.5);
  int sy = (int)floorf(fy);
  fy -= sy;
  if (sy < 0) { sy = 0; fy = 0.f; }
  if (sy >= Hs - 1) { sy = Hs - 1; fy = 0.f; }
  const int sy1 = sy + 1 < Hs ? sy + 1 : sy;
  const size_t o0 = ((size_t)n * Hs + sy) * Ws * 3, o1 = ((size_t)n * Hs + sy1) * Ws * 3;
  const float ay0 = 1.f - fy;
  const float mean[3] = {m0, m1, m2};
  const int x0 = (blockIdx.x * 64 + threadIdx.x) * PPX;
  if (x0 >= Wd) return;
  float out[PPX * 3];
#pragma unroll
  for (int p = 0; p < PPX; ++p) {
    const int x = x0 + p < Wd ? x0 + p : Wd - 1;
    float fx = (float)((x + 0.5) * scale_x - 0.5);
    int sx = (int)floorf(fx);
    fx -= sx;
    if (sx < 0) { sx = 0; fx = 0.f; }
    if (sx >= Ws - 1) { sx = Ws - 1; fx = 0.f; }
    const float ax0 = 1.f - fx;
    // bytes [3*sx, 3*sx + 6) of both source rows: pixel sx and its right neighbour (at the last column the neighbour's
    // weight fx is 0 and its bytes are whatever follows).  The 8-byte load must stay inside the buffer.
    unsigned long long q0, q1;
    const size_t b0 = o0 + (size_t)sx * 3, b1 = o1 + (size_t)sx * 3;
    if (b1 + 8 <= src_bytes && b0 + 8 <= src_bytes) {
      q0 = *reinterpret_cast<const unsigned long long*>(src + b0);
      q1 = *reinterpret_cast<const unsigned long long*>(src + b1);
    } else {                                       // the last pixels of the last image
      q0 = q1 = 0;
      for (int k = 0; k < 6; ++k) {
        if (b0 + k < src_bytes) q0 |= (unsigned long long)src[b0 + k] << (8 * k);
        if (b1 + k < src_bytes) q1 |= (unsigned long long)src[b1 + k] << (8 * k);
      }
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const float s00 = (float)(unsigned)((q0 >> (8 * c)) & 255), s01 = (float)(unsigned)((q0 >> (8 * (c + 3))) & 255);
      const float s10 = (float)(unsigned)((q1 >> (8 * c)) & 255), s11 = (float)(unsigned)((q1 >> (8 * (c + 3))) & 255);
      const float h0 = s00 * ax0 + s01 * fx;
      const float h1 = s10 * ax0 + s11 * fx;
      out[p * 3 + c] = (h0 * ay0 + h1 * fy) - mean[c];
    }
  }
  T* d = dst + ((size_t)row * Wd + x0) * 3;
  if (x0 + PPX <= Wd && (reinterpret_cast<uintptr_t>(d) & 7) == 0) {
    if constexpr (sizeof(T) == 2) {
      typedef f16 h4 __attribute__((ext_vector_type(4)));
#pragma unroll
      for (int k = 0; k < 3; ++k)
        reinterpret_cast<h4*>(d)[k] = h4{(f16)out[4 * k], (f16)out[4 * k + 1], (f16)out[4 * k + 2], (f16)out[4 * k + 3]};
    } else {
      typedef float f2 __attribute__((ext_vector_type(2)));
#pragma unroll
      for (int k = 0; k < 6; ++k) reinterpret_cast<f2*>(d)[k] = f2{out[2 * k], out[2 * k + 1]};
    }
  } else if (sizeof(T) == 2 && x0 + PPX <= Wd && (reinterpret_cast<uintptr_t>(d) & 3) == 0) {
    // rows of an even width that is not a multiple of 4 pixels start 4-byte aligned only (1242 x 3 x 2 B): dword stores
    typedef f16 h2 __attribute__((ext_vector_type(2)));
#pragma unroll
    for (int k = 0; k < 6; ++k) reinterpret_cast<h2*>(d)[k] = h2{(f16)out[2 * k], (f16)out[2 * k + 1]};
  } else {
    for (int p = 0; p < PPX && x0 + p < Wd; ++p)
      for (int c = 0; c < 3; ++c) d[p * 3 + c] = (T)out[p * 3 + c];
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
  SQDET_REQUIRE((long)n * dst_h <= 0x7fffffffL / 4, "preprocess_bgr: too many rows");
  const dim3 grid((unsigned)((dst_w + 64 * PPX - 1) / (64 * PPX)), (unsigned)(n * dst_h));
  SQDET_REQUIRE(grid.y <= 65535u * 1024u, "preprocess_bgr: too many rows");
  const size_t src_bytes = (size_t)n * src_h * src_w * 3;
  if (dtype == SQDET_F16)
    hipLaunchKernelGGL(preprocess_kernel<f16>, grid, dim3(64), 0, as_stream(stream), src_bgr_u8, (f16*)dst,
                       n, src_h, src_w, dst_h, dst_w, mean_b, mean_g, mean_r, src_bytes);
  else
    hipLaunchKernelGGL(preprocess_kernel<float>, grid, dim3(64), 0, as_stream(stream), src_bgr_u8,
                       (float*)dst, n, src_h, src_w, dst_h, dst_w, mean_b, mean_g, mean_r, src_bytes);
  SQDET_CHECK_HIP(hipGetLastError());
  return SQDET_OK;
}
