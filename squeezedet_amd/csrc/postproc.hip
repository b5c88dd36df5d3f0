// interpret_output + filter_prediction for gfx950.
//
//   interpret_kernel   replaces ModelSkeleton._add_interpretation_graph
//                      (reference src/nn_skeleton.py:142-283) + util.safe_exp /
//                      bbox_transform / bbox_transform_inv (src/utils/util.py:167-231).
//   filter_kernel      replaces ModelSkeleton.filter_prediction (nn_skeleton.py:696-734) +
//                      util.nms / util.batch_iou (utils/util.py:32-76).
//
// This translation unit is compiled with -ffp-contract=off: the reference evaluates every
// decode / IoU expression as separate rounded float32 operations (TF-CPU / NumPy), so no
// mul+add may be fused here or clipped boxes and NMS decisions stop being bit-exact.
#include "postproc.h"

namespace sqdet {

// ------------------------------------------------------------------ interpret_output
template <typename T>
__global__ __launch_bounds__(256) void interpret_kernel(const T* __restrict__ preds, const float* __restrict__ anchors,
                                                        float* __restrict__ det_boxes, float* __restrict__ det_probs,
                                                        int64_t* __restrict__ det_class, float* __restrict__ pcp,
                                                        float* __restrict__ pconf, int n, int cells, int apg, int C,
                                                        float w1, float h1, float thr, float slope) {
  const int A = cells * apg;
  const int ch = apg * (C + 5);
  const long total = (long)n * A;
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    const int b = (int)(idx / A);
    const int a = (int)(idx - (long)b * A);
    const int cell = a / apg, k = a - cell * apg;
    const T* p = preds + ((size_t)b * cells + cell) * ch;
    // class probabilities: softmax over channels [k*C, k*C+C) (nn_skeleton.py:150-160); confidence: sigmoid of channel
    // apg*C + k (:163-170); score = max_c(class_prob * conf), class = first argmax (:274-283) -- decode_score, postproc.h
    int bestc;
    const float best = decode_score<T>(p, k, apg, C, &bestc);
    if (pcp || pconf) {   // the optional per-class outputs: the same expressions again
      const T* lg = p + k * C;
      float mx = (float)lg[0];
      for (int c = 1; c < C; ++c) mx = fmaxf(mx, (float)lg[c]);
      float sum = expf((float)lg[0] - mx);
      for (int c = 1; c < C; ++c) sum = sum + expf((float)lg[c] - mx);
      const float inv = 1.0f / sum;
      if (pcp) for (int c = 0; c < C; ++c) pcp[idx * C + c] = expf((float)lg[c] - mx) * inv;
      if (pconf) pconf[idx] = 1.0f / (1.0f + expf(-(float)p[apg * C + k]));
    }
    // box deltas -> box: stretching, safe_exp, trimming, bbox_transform_inv -- decode_box, postproc.h
    const f32x4 an = *reinterpret_cast<const f32x4*>(anchors + (size_t)a * 4);
    const f32x4 ob = decode_box<T>(p, k, apg, C, an, w1, h1, thr, slope);
    *reinterpret_cast<f32x4*>(det_boxes + idx * 4) = ob;
    det_probs[idx] = best;
    det_class[idx] = (int64_t)bestc;
  }
}

// ------------------------------------------------------------------ filter_prediction
// In-LDS bitonic sort of n (power of two) 64-bit keys by 256 threads.
template <bool DESC>
__device__ void bitonic_sort(unsigned long long* keys, int n) {
  for (int k = 2; k <= n; k <<= 1) {
    for (int j = k >> 1; j > 0; j >>= 1) {
      for (int t = threadIdx.x; t < (n >> 1); t += blockDim.x) {
        const int i = ((t / j) * (j << 1)) + (t % j);
        const bool up = (i & k) == 0;
        const unsigned long long a = keys[i], b = keys[i + j];
        const bool swap = DESC ? ((a < b) == up) : ((a > b) == up);
        if (swap) { keys[i] = b; keys[i + j] = a; }
      }
      __syncthreads();
    }
  }
}

// One 256-thread workgroup per image.
// LDS: keys[cap] u64 | boxes[cap] f32x4 | clsv[cap] i32 | keep[cap] i32 | hist[256] | scan[256] | misc[8]
__global__ __launch_bounds__(256) void filter_kernel(FilterArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned long long* keys = reinterpret_cast<unsigned long long*>(smem);
  f32x4* sbox = reinterpret_cast<f32x4*>(smem + (size_t)a.cap * 8);
  int* scls = reinterpret_cast<int*>(smem + (size_t)a.cap * 24);
  int* skeep = scls + a.cap;
  int* hist = skeep + a.cap;
  int* scan = hist + 256;
  int* misc = scan + 256;  // [0]=digit [1]=above [2]=counter

  const int img = blockIdx.x;
  const int tid = threadIdx.x;
  const float* probs = a.probs + (size_t)img * a.A;
  const float* boxes = a.boxes + (size_t)img * a.A * 4;
  const int64_t* cls = a.cls + (size_t)img * a.A;
  float* ob = a.out_boxes + (size_t)img * a.max_out * 4;
  float* op = a.out_probs + (size_t)img * a.max_out;
  int32_t* oc = a.out_cls + (size_t)img * a.max_out;
  int32_t* oi = a.out_index + (size_t)img * a.max_out;

  int M;                           // number of candidates entering NMS
  unsigned long long T = 0;        // top-N branch: key of the top_n-th largest entry
  if (a.use_topn) {
    // ---- exact radix select of the top_n-th largest composite key, MSB byte first ----
    unsigned long long prefix = 0;
    int remaining = a.top_n;
    for (int byte = 7; byte >= 0; --byte) {
      if (byte < 4 && ((unsigned int)(a.A - 1) >> (8 * byte)) == 0) continue;  // index byte is all-zero
      hist[tid] = 0;
      __syncthreads();
      const int hs = 8 * (byte + 1);
      for (int i = tid; i < a.A; i += 256) {
        const unsigned long long key = make_key(probs[i], i);
        const bool match = byte == 7 || (key >> hs) == (prefix >> hs);
        if (match) atomicAdd(&hist[(int)((key >> (8 * byte)) & 255)], 1);
      }
      __syncthreads();
      scan[tid] = hist[tid];
      __syncthreads();
      for (int off = 1; off < 256; off <<= 1) {  // inclusive suffix sum
        const int v = tid + off < 256 ? scan[tid + off] : 0;
        __syncthreads();
        scan[tid] += v;
        __syncthreads();
      }
      const int mine = scan[tid];
      const int above = tid == 255 ? 0 : scan[tid + 1];
      if (mine >= remaining && above < remaining) { misc[0] = tid; misc[1] = above; }
      __syncthreads();
      prefix |= (unsigned long long)misc[0] << (8 * byte);
      remaining -= misc[1];
      __syncthreads();
    }
    T = prefix;
    M = a.top_n;
  } else {
    // ---- threshold branch: count entries with prob > PROB_THRESH (nn_skeleton.py:716-720) ----
    if (tid == 0) misc[2] = 0;
    __syncthreads();
    int local = 0;
    for (int i = tid; i < a.A; i += 256) local += probs[i] > a.prob_thresh ? 1 : 0;
    atomicAdd(&misc[2], local);
    __syncthreads();
    M = misc[2];
    __syncthreads();
    if (M > a.max_out) {
      if (tid == 0) a.out_count[img] = -M;
      return;
    }
  }

  // ---- compaction of the candidates' keys, then descending sort: rank r = position ----
  int n2 = 1;
  while (n2 < M) n2 <<= 1;
  if (n2 < 2) n2 = 2;
  if (tid == 0) misc[2] = 0;
  for (int i = tid; i < n2; i += 256) keys[i] = 0ull;
  __syncthreads();
  for (int i = tid; i < a.A; i += 256) {
    const float p = probs[i];
    const unsigned long long key = make_key(p, i);
    const bool take = a.use_topn ? key >= T : p > a.prob_thresh;
    if (take) {
      const int slot = atomicAdd(&misc[2], 1);
      if (slot < n2) keys[slot] = key;
    }
  }
  __syncthreads();
  bitonic_sort<true>(keys, n2);

  for (int r = tid; r < M; r += 256) {
    const int idx = (int)(keys[r] & 0xffffffffull);
    sbox[r] = *reinterpret_cast<const f32x4*>(boxes + (size_t)idx * 4);
    scls[r] = (int)cls[idx];
  }
  __syncthreads();

  // ---- the reference's non-greedy NMS (utils/util.py:56-76): box r is dropped iff ANY
  //      higher-ranked same-class box i has IoU(i, r) > threshold (no "is i still kept").
  for (int r = tid; r < M; r += 256) {
    const f32x4 bj = sbox[r];
    const int cj = scls[r];
    bool keep = cj >= 0 && cj < a.C;
    for (int i = 0; i < r && keep; ++i) {
      if (scls[i] != cj) continue;
      const f32x4 bi = sbox[i];
      const float ov = iou_center(bj, bi);
      if ((double)ov > a.nms_thresh) keep = false;
    }
    skeep[r] = keep ? 1 : 0;
  }
  __syncthreads();

  // ---- output order: class ascending, then descending prob (top-N branch: the arrays were
  //      re-ordered by prob, nn_skeleton.py:711-715) or ascending anchor index (threshold
  //      branch: arrays stay in anchor order, nn_skeleton.py:716-733).
  //      key2 = cls << 40 | order-field << 12 | rank ; dropped entries sort last.
  for (int r = tid; r < n2; r += 256) {
    unsigned long long k2 = ~0ull;
    if (r < M && skeep[r]) {
      const unsigned long long ordf = a.use_topn ? (unsigned long long)r : (keys[r] & 0xffffffffull);
      k2 = ((unsigned long long)scls[r] << 40) | (ordf << 12) | (unsigned long long)r;
    }
    reinterpret_cast<unsigned long long*>(smem + (size_t)a.cap * 32 + 4096)[r] = k2;
  }
  __syncthreads();
  unsigned long long* keys2 = reinterpret_cast<unsigned long long*>(smem + (size_t)a.cap * 32 + 4096);
  bitonic_sort<false>(keys2, n2);

  if (tid == 0) misc[2] = 0;
  __syncthreads();
  int kept_local = 0;
  for (int o = tid; o < n2; o += 256) {
    const unsigned long long k2 = keys2[o];
    if (k2 != ~0ull) {
      const int r = (int)(k2 & 0xfffull);
      const int idx = (int)(keys[r] & 0xffffffffull);
      *reinterpret_cast<f32x4*>(ob + (size_t)o * 4) = sbox[r];
      op[o] = probs[idx];
      oc[o] = scls[r];
      oi[o] = idx;
      ++kept_local;
    }
  }
  atomicAdd(&misc[2], kept_local);
  __syncthreads();
  const int kept = misc[2];
  for (int o = kept + tid; o < a.max_out; o += 256) {
    *reinterpret_cast<f32x4*>(ob + (size_t)o * 4) = f32x4{0.f, 0.f, 0.f, 0.f};
    op[o] = 0.f;
    oc[o] = -1;
    oi[o] = -1;
  }
  if (tid == 0) a.out_count[img] = kept;
}

}  // namespace sqdet

using namespace sqdet;

extern "C" int sqdet_interpret_output(const void* preds, const float* anchors, float* det_boxes, float* det_probs,
                                      int64_t* det_class, float* pred_class_probs, float* pred_conf, int n, int gh,
                                      int gw, int apg, int classes, float img_w, float img_h, float exp_thresh,
                                      int dtype, sqdet_stream_t stream) {
  SQDET_REQUIRE(preds && anchors && det_boxes && det_probs && det_class, "interpret_output: null pointer");
  SQDET_REQUIRE(n > 0 && gh > 0 && gw > 0 && apg > 0 && classes > 0, "interpret_output: bad dims");
  SQDET_REQUIRE(dtype == SQDET_F16 || dtype == SQDET_F32, "interpret_output: bad dtype %d", dtype);
  const long total = (long)n * gh * gw * apg;
  long blocks = (total + 255) / 256;
  if (blocks > 8192) blocks = 8192;
  // util.safe_exp: slope = np.exp(thresh) (float64) used as a float32 constant
  const float slope = (float)exp((double)exp_thresh);
  const float w1 = img_w - 1.0f, h1 = img_h - 1.0f;
  if (dtype == SQDET_F16)
    hipLaunchKernelGGL(interpret_kernel<f16>, dim3((unsigned)blocks), dim3(256), 0, as_stream(stream),
                       (const f16*)preds, anchors, det_boxes, det_probs, det_class, pred_class_probs, pred_conf, n,
                       gh * gw, apg, classes, w1, h1, exp_thresh, slope);
  else
    hipLaunchKernelGGL(interpret_kernel<float>, dim3((unsigned)blocks), dim3(256), 0, as_stream(stream),
                       (const float*)preds, anchors, det_boxes, det_probs, det_class, pred_class_probs, pred_conf, n,
                       gh * gw, apg, classes, w1, h1, exp_thresh, slope);
  SQDET_CHECK_HIP(hipGetLastError());
  return SQDET_OK;
}

extern "C" int sqdet_filter_prediction(const float* boxes, const float* probs, const int64_t* cls, float* out_boxes,
                                       float* out_probs, int32_t* out_cls, int32_t* out_index, int32_t* out_count,
                                       int n, int num_anchors, int classes, int top_n, int max_out, double nms_thresh,
                                       float prob_thresh, sqdet_stream_t stream) {
  SQDET_REQUIRE(boxes && probs && cls && out_boxes && out_probs && out_cls && out_index && out_count,
                "filter_prediction: null pointer");
  SQDET_REQUIRE(n > 0 && num_anchors > 0 && classes > 0 && max_out > 0, "filter_prediction: bad dims");
  SQDET_UNSUPPORTED(classes >= (1 << 23), "filter_prediction: too many classes");
  FilterArgs a;
  a.boxes = boxes; a.probs = probs; a.cls = cls;
  a.out_boxes = out_boxes; a.out_probs = out_probs; a.out_cls = out_cls; a.out_index = out_index; a.out_count = out_count;
  a.A = num_anchors; a.C = classes; a.top_n = top_n; a.max_out = max_out;
  a.use_topn = (top_n > 0 && top_n < num_anchors) ? 1 : 0;  // nn_skeleton.py:711
  a.nms_thresh = nms_thresh; a.prob_thresh = prob_thresh;
  int need = a.use_topn ? top_n : (max_out < num_anchors ? max_out : num_anchors);
  SQDET_REQUIRE(!a.use_topn || max_out >= top_n, "filter_prediction: max_out %d < top_n %d", max_out, top_n);
  int cap = 2;
  while (cap < need) cap <<= 1;
  SQDET_UNSUPPORTED(cap > 1024, "filter_prediction: more than 1024 NMS candidates per image (%d) not supported", need);
  a.cap = cap;
  bool handled = false;
  const int rc = filter_topn_fast_launch(a, n, as_stream(stream), &handled);
  if (rc != SQDET_OK || handled) return rc;
  const size_t lds = (size_t)cap * 32 + 4096 + (size_t)cap * 8;
  hipLaunchKernelGGL(filter_kernel, dim3(n), dim3(256), lds, as_stream(stream), a);
  SQDET_CHECK_HIP(hipGetLastError());
  return SQDET_OK;
}

static int detect_filter_impl(const void* preds, const float* anchors, float* scratch_probs, float* out_boxes,
                              float* out_probs, int32_t* out_cls, int32_t* out_index, int32_t* out_count, int n, int gh,
                              int gw, int apg, int classes, float img_w, float img_h, float exp_thresh, int top_n,
                              int max_out, double nms_thresh, int dtype, bool scores_ready, int max_wgs, sqdet_stream_t stream) {
  SQDET_REQUIRE(preds && anchors && scratch_probs && out_boxes && out_probs && out_cls && out_index && out_count,
                "detect_filter: null pointer");
  SQDET_REQUIRE(n > 0 && gh > 0 && gw > 0 && apg > 0 && classes > 0 && max_out >= top_n, "detect_filter: bad dims");
  SQDET_REQUIRE(dtype == SQDET_F16 || dtype == SQDET_F32, "detect_filter: bad dtype %d", dtype);
  const int A = gh * gw * apg;
  FilterArgs a;
  a.boxes = nullptr; a.probs = scratch_probs; a.cls = nullptr;
  a.out_boxes = out_boxes; a.out_probs = out_probs; a.out_cls = out_cls; a.out_index = out_index; a.out_count = out_count;
  a.A = A; a.C = classes; a.top_n = top_n; a.max_out = max_out; a.cap = 0;
  a.use_topn = (top_n > 0 && top_n < A) ? 1 : 0;
  a.nms_thresh = nms_thresh; a.prob_thresh = 0.f;
  DecodeArgs d;
  d.preds = preds; d.anchors = anchors; d.cells = gh * gw; d.apg = apg; d.C = classes; d.dtype = dtype;
  d.w1 = img_w - 1.0f; d.h1 = img_h - 1.0f; d.thr = exp_thresh; d.slope = (float)exp((double)exp_thresh);
  bool handled = false;
  const int rc = detect_topn_fused_launch(a, d, n, as_stream(stream), &handled, scores_ready, max_wgs);
  if (rc != SQDET_OK) return rc;
  SQDET_UNSUPPORTED(!handled, "detect_filter: needs the top-N branch with 0 < top_n <= 64 < anchors <= 20480 (else interpret_output + filter_prediction)");
  return SQDET_OK;
}

extern "C" int sqdet_detect_filter(const void* preds, const float* anchors, float* scratch_probs, float* out_boxes,
                                   float* out_probs, int32_t* out_cls, int32_t* out_index, int32_t* out_count, int n, int gh,
                                   int gw, int apg, int classes, float img_w, float img_h, float exp_thresh, int top_n,
                                   int max_out, double nms_thresh, int dtype, sqdet_stream_t stream) {
  return detect_filter_impl(preds, anchors, scratch_probs, out_boxes, out_probs, out_cls, out_index, out_count, n, gh, gw, apg,
                            classes, img_w, img_h, exp_thresh, top_n, max_out, nms_thresh, dtype, false, 0, stream);
}

extern "C" int sqdet_detect_filter_scored(const void* preds, const float* anchors, const float* scores, float* out_boxes,
                                          float* out_probs, int32_t* out_cls, int32_t* out_index, int32_t* out_count, int n,
                                          int gh, int gw, int apg, int classes, float img_w, float img_h, float exp_thresh,
                                          int top_n, int max_out, double nms_thresh, int dtype, int max_workgroups,
                                          sqdet_stream_t stream) {
  return detect_filter_impl(preds, anchors, const_cast<float*>(scores), out_boxes, out_probs, out_cls, out_index, out_count, n, gh,
                            gw, apg, classes, img_w, img_h, exp_thresh, top_n, max_out, nms_thresh, dtype, true, max_workgroups, stream);
}
