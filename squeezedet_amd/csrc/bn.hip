// Frozen-statistics batch norm of ModelSkeleton._conv_bn_layer (reference src/nn_skeleton.py:374-468)
// folded into the conv it follows.  tf.nn.batch_normalization computes
//   inv = rsqrt(var + eps) * gamma;  y = x * inv + (beta - mean * inv)
// with x = conv2d(in, W) [+ biases]; mean / var are non-trainable variables (:437-438), so the whole
// thing is the conv with W[..., c] * inv[c] and bias (biases[c] - mean[c]) * inv[c] + beta[c].
// A load-time transform: one pass over the kernel, HBM-streaming, one thread per 4 output channels.
#include "common.h"

namespace sqdet {

// w / wf may alias (in-place fold): every thread reads and writes its own 4 elements only.
__global__ __launch_bounds__(256) void fold_bn_kernel(const float* w, const float* __restrict__ cbias,
                                                      const float* __restrict__ gamma, const float* __restrict__ beta,
                                                      const float* __restrict__ mean, const float* __restrict__ var,
                                                      float eps, float* wf, float* __restrict__ bf,
                                                      size_t rows, int cout) {
  const int cv = cout / 4;
  const size_t total = rows * cv;
  for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
    const int c = (int)(idx % cv) * 4;
    f32x4 inv;
#pragma unroll
    for (int e = 0; e < 4; ++e) inv[e] = gamma[c + e] / sqrtf(var[c + e] + eps);
    const f32x4 v = *reinterpret_cast<const f32x4*>(w + idx * 4);
    *reinterpret_cast<f32x4*>(wf + idx * 4) = v * inv;
    if (idx < (size_t)cv) {  // first kernel row's threads also produce the folded bias
#pragma unroll
      for (int e = 0; e < 4; ++e) bf[c + e] = ((cbias ? cbias[c + e] : 0.f) - mean[c + e]) * inv[e] + beta[c + e];
    }
  }
}

int fold_bn_launch(const float* w, const float* cbias, const float* gamma, const float* beta, const float* mean,
                   const float* var, float eps, float* wf, float* bf, int k, int cin, int cout, hipStream_t st) {
  SQDET_REQUIRE(w && gamma && beta && mean && var && wf && bf, "fold_batchnorm: null pointer");
  SQDET_REQUIRE(k > 0 && cin > 0 && cout > 0 && eps >= 0.f, "fold_batchnorm: bad dims");
  SQDET_UNSUPPORTED(cout % 4 != 0, "fold_batchnorm: cout %d not a multiple of 4", cout);
  const size_t rows = (size_t)k * k * cin;
  const size_t total = rows * (cout / 4);
  size_t blocks = (total + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(fold_bn_kernel, dim3((unsigned)blocks), dim3(256), 0, st, w, cbias, gamma, beta, mean, var, eps, wf,
                     bf, rows, cout);
  SQDET_CHECK_HIP(hipGetLastError());
  return SQDET_OK;
}

}  // namespace sqdet

extern "C" int sqdet_fold_batchnorm(const float* w_hwio, const float* conv_bias, const float* gamma, const float* beta,
                                    const float* mean, const float* var, float eps, float* w_folded, float* b_folded,
                                    int k, int cin, int cout, sqdet_stream_t stream) {
  return sqdet::fold_bn_launch(w_hwio, conv_bias, gamma, beta, mean, var, eps, w_folded, b_folded, k, cin, cout,
                               sqdet::as_stream(stream));
}
