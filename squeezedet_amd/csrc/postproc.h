// Shared by postproc.hip (generic filter kernel) and filter_fast.hip (top-N <= 64 fast path).
#pragma once
#include "common.h"

namespace sqdet {

// Order-preserving map float -> uint32 (larger float <=> larger uint), then a 64-bit
// composite key (key32 << 32 | anchor index): all keys are distinct, and a DESCENDING sort of
// them is "descending prob, ties -> higher anchor index first" (the tie rule this repo
// defines; the reference's unstable argsort leaves ties unspecified -- SURVEY.md 9.4).
__device__ __forceinline__ unsigned int order_key32(float p) {
  unsigned int b = __float_as_uint(p);
  return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__device__ __forceinline__ unsigned long long make_key(float p, int idx) {
  return ((unsigned long long)order_key32(p) << 32) | (unsigned int)idx;
}

struct FilterArgs {
  const float* boxes;
  const float* probs;
  const int64_t* cls;
  float* out_boxes;
  float* out_probs;
  int32_t* out_cls;
  int32_t* out_index;
  int32_t* out_count;
  int A, C, top_n, max_out, cap;  // cap: power-of-two LDS capacity (>= candidates)
  int use_topn;
  double nms_thresh;
  float prob_thresh;
};

// utils/util.py:32-54 batch_iou(boxes = lower-ranked j, box = higher-ranked i), float32 op for op
// (this code is only ever compiled with -ffp-contract=off).
__device__ __forceinline__ float iou_center(const f32x4& bj, const f32x4& bi) {
  const float lr = fmaxf(fminf(bj[0] + 0.5f * bj[2], bi[0] + 0.5f * bi[2]) -
                         fmaxf(bj[0] - 0.5f * bj[2], bi[0] - 0.5f * bi[2]), 0.0f);
  const float tb = fmaxf(fminf(bj[1] + 0.5f * bj[3], bi[1] + 0.5f * bi[3]) -
                         fmaxf(bj[1] - 0.5f * bj[3], bi[1] - 0.5f * bi[3]), 0.0f);
  const float inter = lr * tb;
  const float uni = bj[2] * bj[3] + bi[2] * bi[3] - inter;
  return inter / uni;
}

int filter_topn_fast_launch(const FilterArgs& a, int n, hipStream_t st, bool* handled);

}  // namespace sqdet
