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

// ---- interpret_output arithmetic, shared by interpret_kernel (postproc.hip) and the fused decode + filter kernel
// (filter_fast.hip); both translation units are compiled with -ffp-contract=off ----
struct DecodeArgs {
  const void* preds;      // [n, cells, apg*(C+5)] in dtype storage
  const float* anchors;   // [A,4] float32
  int cells, apg, C, dtype;
  float w1, h1, thr, slope;
};

// score = max_c(softmax(class logits)_c * sigmoid(conf)), class = first argmax (nn_skeleton.py:150-170, 274-283), on the
// float32 VALUES of the stored logits.  ONE definition for the score kernel, the filter kernel's re-decode and the score
// pass in the ConvDet epilogue (convdet.hip): the three must agree bit for bit (all compiled with -ffp-contract=off).
__device__ __forceinline__ float score_from_logits(const float* lg, int C, float conf_logit, int* bestc_out) {
  float mx = lg[0];
  for (int c = 1; c < C; ++c) mx = fmaxf(mx, lg[c]);
  float sum = expf(lg[0] - mx);
  for (int c = 1; c < C; ++c) sum = sum + expf(lg[c] - mx);
  const float inv = 1.0f / sum;
  const float conf = 1.0f / (1.0f + expf(-conf_logit));
  float best = 0.f;
  int bestc = 0;
  for (int c = 0; c < C; ++c) {
    const float pr = (expf(lg[c] - mx) * inv) * conf;
    if (c == 0 || pr > best) { best = pr; bestc = c; }
  }
  *bestc_out = bestc;
  return best;
}

template <typename T>
__device__ __forceinline__ float decode_score(const T* p, int k, int apg, int C, int* bestc_out) {
  const T* lgp = p + k * C;
  if (C == 3) {      // (the reference's CLASSES: a register array)
    const float lg[3] = {(float)lgp[0], (float)lgp[1], (float)lgp[2]};
    return score_from_logits(lg, 3, (float)p[apg * C + k], bestc_out);
  }
  float mx = (float)lgp[0];
  for (int c = 1; c < C; ++c) mx = fmaxf(mx, (float)lgp[c]);
  float sum = expf((float)lgp[0] - mx);
  for (int c = 1; c < C; ++c) sum = sum + expf((float)lgp[c] - mx);
  const float inv = 1.0f / sum;
  const float conf = 1.0f / (1.0f + expf(-(float)p[apg * C + k]));
  float best = 0.f;
  int bestc = 0;
  for (int c = 0; c < C; ++c) {
    const float pr = (expf((float)lgp[c] - mx) * inv) * conf;
    if (c == 0 || pr > best) { best = pr; bestc = c; }
  }
  *bestc_out = bestc;
  return best;
}

// box deltas -> (cx, cy, w, h): stretching, safe_exp, trimming, bbox_transform_inv (nn_skeleton.py:173-233, utils/util.py:167-231)
template <typename T>
__device__ __forceinline__ f32x4 decode_box(const T* p, int k, int apg, int C, const f32x4& an, float w1, float h1, float thr, float slope) {
  const T* dl = p + apg * (C + 1) + 4 * k;
  const float dx = (float)dl[0], dy = (float)dl[1], dw = (float)dl[2], dh = (float)dl[3];
  const float cx = an[0] + dx * an[2];
  const float cy = an[1] + dy * an[3];
  const float ew = dw > thr ? slope * ((dw - thr) + 1.0f) : expf(dw);
  const float eh = dh > thr ? slope * ((dh - thr) + 1.0f) : expf(dh);
  const float bw = an[2] * ew;
  const float bh = an[3] * eh;
  float xmin = cx - bw / 2.0f, ymin = cy - bh / 2.0f, xmax = cx + bw / 2.0f, ymax = cy + bh / 2.0f;
  xmin = fminf(fmaxf(0.0f, xmin), w1);
  ymin = fminf(fmaxf(0.0f, ymin), h1);
  xmax = fmaxf(fminf(w1, xmax), 0.0f);
  ymax = fmaxf(fminf(h1, ymax), 0.0f);
  const float w2 = xmax - xmin + 1.0f;
  const float h2 = ymax - ymin + 1.0f;
  f32x4 ob;
  ob[0] = xmin + 0.5f * w2;
  ob[1] = ymin + 0.5f * h2;
  ob[2] = w2;
  ob[3] = h2;
  return ob;
}

int filter_topn_fast_launch(const FilterArgs& a, int n, hipStream_t st, bool* handled);
// interpret_output + filter_prediction (top-N branch): a chip-wide score kernel, then the filter kernel with boxes decoded for the
// selected anchors only; a.probs = scratch [n, A] (scores, read back by the mass-tie fallback), a.boxes / a.cls unused
// scores_ready: a.probs already holds the scores (written by the ConvDet epilogue, sqdet_convdet_fwd): the score kernel is skipped
// max_wgs > 0: at most that many workgroups (each walks several images)
int detect_topn_fused_launch(const FilterArgs& a, const DecodeArgs& d, int n, hipStream_t st, bool* handled, bool scores_ready = false,
                             int max_wgs = 0);

}  // namespace sqdet
