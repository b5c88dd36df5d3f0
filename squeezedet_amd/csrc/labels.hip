// Training labels on the GPU: ground-truth box -> responsible anchor assignment and the dense label tensors
// the loss consumes (reference src/dataset/imdb.py:195-239 `read_batch` + src/train.py:163-224 `_load_data`
// + src/utils/util.py:139-158 `sparse_to_dense`).  In the reference this is per-image Python: for every box a
// 16 848-wide batch_iou, a full argsort and set bookkeeping, then Python-loop scatters.
//
// One 1024-thread workgroup per image:
//   * zero the image's rows of input_mask / box_delta_input / box_input / labels;
//   * boxes are taken IN ORDER (earlier boxes claim anchors first, imdb.py:199): all threads evaluate
//     batch_iou (utils/util.py:32-54, float64 like mc.ANCHOR_BOX) against their slice of the anchors,
//     skipping anchors already claimed, and the workgroup reduces to the largest overlap (> 0); exact ties
//     go to the HIGHER anchor index (np.argsort(...)[::-1] with a stable sort);
//   * if no free anchor overlaps, the free anchor of smallest squared distance wins, ties to the LOWER
//     index (imdb.py:222-229);
//   * delta = ((cx-ax)/aw, (cy-ay)/ah, log(w/aw), log(h/ah)) in float64 (:231-236), stored as float32 -- the
//     dtype of the reference's placeholders (nn_skeleton.py:86-97).
// A box assigned to an anchor that an earlier box of the same image already owns cannot happen here (the
// claimed set is per image), which is exactly the duplicate filter of train.py:175-186.
#include "common.h"

namespace sqdet {

struct Cand {
  double v;
  int idx;
};

// better(a, b): a wins over b.  MAXMODE: larger v, ties -> higher idx.  else: smaller v, ties -> lower idx.
template <bool MAXMODE>
__device__ __forceinline__ bool better(const Cand& a, const Cand& b) {
  if (a.idx < 0) return false;
  if (b.idx < 0) return true;
  if (MAXMODE) return a.v > b.v || (a.v == b.v && a.idx > b.idx);
  return a.v < b.v || (a.v == b.v && a.idx < b.idx);
}

template <bool MAXMODE>
__device__ __forceinline__ Cand block_best(Cand c, Cand* red) {
  // wave reduction, then one candidate per wave through LDS
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    Cand o;
    o.v = __shfl_down(c.v, off);
    o.idx = __shfl_down(c.idx, off);
    if (better<MAXMODE>(o, c)) c = o;
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  __syncthreads();                       // red[] free (previous use finished)
  if (lane == 0) red[wave] = c;
  __syncthreads();
  Cand b = red[0];
  for (int w = 1; w < (int)(blockDim.x >> 6); ++w)
    if (better<MAXMODE>(red[w], b)) b = red[w];
  return b;
}

__global__ __launch_bounds__(1024) void build_labels_kernel(const double* __restrict__ anchors, const double* __restrict__ gt,
                                                            const int* __restrict__ gt_cls, const int* __restrict__ gt_count,
                                                            float* __restrict__ mask, float* __restrict__ delta,
                                                            float* __restrict__ box, float* __restrict__ labels,
                                                            int* __restrict__ aidx_out, int A, int M, int C) {
  extern __shared__ unsigned int taken[];            // A bits
  __shared__ Cand red[16];
  const int b = blockIdx.x;
  const int words = (A + 31) / 32;
  for (int i = threadIdx.x; i < words; i += blockDim.x) taken[i] = 0u;
  float* m = mask + (size_t)b * A;
  float* d = delta + (size_t)b * A * 4;
  float* bx = box + (size_t)b * A * 4;
  float* lb = labels + (size_t)b * A * C;
  for (int i = threadIdx.x; i < A; i += blockDim.x) m[i] = 0.f;
  for (int i = threadIdx.x; i < A * 4; i += blockDim.x) { d[i] = 0.f; bx[i] = 0.f; }
  for (int i = threadIdx.x; i < A * C; i += blockDim.x) lb[i] = 0.f;
  __syncthreads();
  int n = gt_count[b];
  if (n > M) n = M;
  for (int i = 0; i < M; ++i)
    if (threadIdx.x == 0 && i >= n) aidx_out[(size_t)b * M + i] = -1;
  for (int i = 0; i < n; ++i) {
    const double* g = gt + ((size_t)b * M + i) * 4;
    const double gx = g[0], gy = g[1], gw = g[2], gh = g[3];
    const double gl = gx - 0.5 * gw, gr = gx + 0.5 * gw, gtp = gy - 0.5 * gh, gb = gy + 0.5 * gh;
    const double garea = gw * gh;
    Cand c;
    c.v = 0.0; c.idx = -1;
    for (int a = threadIdx.x; a < A; a += blockDim.x) {
      if (taken[a >> 5] & (1u << (a & 31))) continue;
      const double ax = anchors[a * 4], ay = anchors[a * 4 + 1], aw = anchors[a * 4 + 2], ah = anchors[a * 4 + 3];
      double lr = fmin(ax + 0.5 * aw, gr) - fmax(ax - 0.5 * aw, gl);
      lr = lr > 0.0 ? lr : 0.0;
      double tb = fmin(ay + 0.5 * ah, gb) - fmax(ay - 0.5 * ah, gtp);
      tb = tb > 0.0 ? tb : 0.0;
      const double inter = lr * tb;
      const double iou = inter / (aw * ah + garea - inter);
      Cand t;
      t.v = iou; t.idx = a;
      if (iou > 0.0 && better<true>(t, c)) c = t;
    }
    Cand best = block_best<true>(c, red);
    if (best.idx < 0) {                              // every free anchor has zero overlap: nearest one
      c.v = 0.0; c.idx = -1;
      for (int a = threadIdx.x; a < A; a += blockDim.x) {
        if (taken[a >> 5] & (1u << (a & 31))) continue;
        const double d0 = gx - anchors[a * 4], d1 = gy - anchors[a * 4 + 1], d2 = gw - anchors[a * 4 + 2], d3 = gh - anchors[a * 4 + 3];
        Cand t;
        t.v = ((d0 * d0 + d1 * d1) + d2 * d2) + d3 * d3;
        t.idx = a;
        if (better<false>(t, c)) c = t;
      }
      best = block_best<false>(c, red);
    }
    if (threadIdx.x == 0) {
      const int a = best.idx;                        // A >= number of boxes: a free anchor always exists
      aidx_out[(size_t)b * M + i] = a;
      if (a >= 0) {
        taken[a >> 5] |= 1u << (a & 31);
        const double ax = anchors[a * 4], ay = anchors[a * 4 + 1], aw = anchors[a * 4 + 2], ah = anchors[a * 4 + 3];
        m[a] = 1.f;
        d[a * 4 + 0] = (float)((gx - ax) / aw);
        d[a * 4 + 1] = (float)((gy - ay) / ah);
        d[a * 4 + 2] = (float)log(gw / aw);
        d[a * 4 + 3] = (float)log(gh / ah);
        bx[a * 4 + 0] = (float)gx; bx[a * 4 + 1] = (float)gy; bx[a * 4 + 2] = (float)gw; bx[a * 4 + 3] = (float)gh;
        const int cls = gt_cls[(size_t)b * M + i];
        if (cls >= 0 && cls < C) lb[a * C + cls] = 1.f;
      }
    }
    __syncthreads();                                 // taken[] update visible before the next box
  }
}

}  // namespace sqdet

extern "C" int sqdet_build_labels(const double* anchors_f64, const double* gt_boxes_f64, const int* gt_classes,
                                  const int* gt_counts, float* input_mask, float* box_delta_input, float* box_input,
                                  float* labels, int* anchor_index, int batch, int num_anchors, int max_objects,
                                  int classes, sqdet_stream_t stream) {
  using namespace sqdet;
  SQDET_REQUIRE(anchors_f64 && gt_boxes_f64 && gt_classes && gt_counts && input_mask && box_delta_input && box_input &&
                    labels && anchor_index, "build_labels: null pointer");
  SQDET_REQUIRE(batch > 0 && num_anchors > 0 && max_objects > 0 && classes > 0, "build_labels: bad dims");
  SQDET_UNSUPPORTED(max_objects > num_anchors, "build_labels: more boxes per image than anchors");
  const size_t lds = (size_t)((num_anchors + 31) / 32) * 4;
  SQDET_UNSUPPORTED(lds > 60000, "build_labels: too many anchors (%d)", num_anchors);
  hipLaunchKernelGGL(build_labels_kernel, dim3((unsigned)batch), dim3(1024), lds, as_stream(stream), anchors_f64, gt_boxes_f64,
                     gt_classes, gt_counts, input_mask, box_delta_input, box_input, labels, anchor_index, num_anchors,
                     max_objects, classes);
  SQDET_CHECK_HIP(hipGetLastError());
  return SQDET_OK;
}
