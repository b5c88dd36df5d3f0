// filter_prediction, top-N branch, the stand-alone launches: one 1024-thread workgroup per image (or fewer workgroups walking
// several images) around filter_body.h's per-image body, and the chip-wide score kernel for preds whose scores the ConvDet
// epilogue did not already write.
#include "filter_body.h"

namespace sqdet {

constexpr int FT = 1024;     // threads

// One thread per anchor over the whole chip (computing the scores inside the per-image workgroup would put 17 anchors x 5 expf
// per thread on ONE CU per image: 27 us of a 58 us kernel).
template <typename T>
__global__ __launch_bounds__(256) void score_kernel(DecodeArgs d, float* __restrict__ probs, int A, int total) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  const int img = i / A, an = i - img * A;
  const int cell = an / d.apg, k = an - cell * d.apg;
  const T* p = reinterpret_cast<const T*>(d.preds) + ((size_t)img * d.cells + cell) * d.apg * (d.C + 5);
  int bc;
  probs[i] = decode_score<T>(p, k, d.apg, d.C, &bc);
}

// Grid: one workgroup per image, or FEWER (n_img > gridDim.x: a workgroup walks images blockIdx.x, + gridDim.x, ...).
template <bool FUSED, typename T>
__global__ __launch_bounds__(FT) void filter_topn_fast(FilterArgs a, DecodeArgs d, int n_img) {
  __shared__ FastLds<FT> s;
  for (int img = blockIdx.x; img < n_img; img += gridDim.x) {
    filter_one_image<FUSED, T, FT>(a, d, img, s);
    __syncthreads();   // (the next image reuses the LDS state)
  }
}

int filter_topn_fast_launch(const FilterArgs& a, int n, hipStream_t st, bool* handled) {
  *handled = false;
  if (!a.use_topn || a.top_n > 64 || a.A > 20480) return SQDET_OK;
  hipLaunchKernelGGL((filter_topn_fast<false, float>), dim3(n), dim3(FT), 0, st, a, DecodeArgs{}, n);
  SQDET_CHECK_HIP(hipGetLastError());
  *handled = true;
  return SQDET_OK;
}

int detect_topn_fused_launch(const FilterArgs& a, const DecodeArgs& d, int n, hipStream_t st, bool* handled, bool scores_ready, int max_wgs) {
  *handled = false;
  if (!a.use_topn || a.top_n > 64 || a.A > 20480) return SQDET_OK;
  const int total = n * a.A;
  const int wgs = max_wgs > 0 && max_wgs < n ? max_wgs : n;
  float* scores = const_cast<float*>(a.probs);
  if (d.dtype == SQDET_F16) {
    if (!scores_ready) hipLaunchKernelGGL((score_kernel<f16>), dim3((total + 255) / 256), dim3(256), 0, st, d, scores, a.A, total);
    hipLaunchKernelGGL((filter_topn_fast<true, f16>), dim3(wgs), dim3(FT), 0, st, a, d, n);
  } else {
    if (!scores_ready) hipLaunchKernelGGL((score_kernel<float>), dim3((total + 255) / 256), dim3(256), 0, st, d, scores, a.A, total);
    hipLaunchKernelGGL((filter_topn_fast<true, float>), dim3(wgs), dim3(FT), 0, st, a, d, n);
  }
  SQDET_CHECK_HIP(hipGetLastError());
  *handled = true;
  return SQDET_OK;
}

}  // namespace sqdet
