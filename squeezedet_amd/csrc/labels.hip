// Training labels on the GPU: ground-truth box -> responsible anchor assignment and the dense label tensors
// the loss consumes (reference src/dataset/imdb.py:195-239 `read_batch` + src/train.py:163-224 `_load_data`
// + src/utils/util.py:139-158 `sparse_to_dense`).  In the reference this is per-image Python: for every box a
// 16 848-wide batch_iou, a full argsort and set bookkeeping, then Python-loop scatters.
//
// Three steps (the first version did all of it in one workgroup per image -- 16 MB of zeroing and up to 32 sequential
// 16 848-anchor float64 sweeps per workgroup: 333 us for a 20-image batch):
//   1. labels_zero_kernel clears the four dense tensors (a kernel, not hipMemsetAsync: inside a captured hipGraph the memset
//      nodes misbehaved -- a replayed training step went non-finite after ~30 replays);
//   2. labels_best_kernel, one workgroup per (image, box): the box's best anchor IGNORING what earlier boxes claimed --
//      if that anchor turns out to be free it is also the best among the free ones, in both modes below;
//   3. labels_resolve_kernel, one workgroup per image: boxes in order; a candidate that is still free is accepted, a
//      claimed one (rare) triggers the full sweep over the free anchors; then one thread per box writes its rows.
// The selection rules (unchanged):
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

// The best anchor of box g among the anchors whose bit in `taken` is clear (taken == nullptr: among all): largest
// overlap > 0, else nearest.  All threads of the workgroup call it and get the result.
__device__ int best_anchor(const double* __restrict__ anchors, int A, const double* g, const unsigned int* taken, Cand* red) {
    const double gx = g[0], gy = g[1], gw = g[2], gh = g[3];
    const double gl = gx - 0.5 * gw, gr = gx + 0.5 * gw, gtp = gy - 0.5 * gh, gb = gy + 0.5 * gh;
    const double garea = gw * gh;
    Cand c;
    c.v = 0.0; c.idx = -1;
    for (int a = threadIdx.x; a < A; a += blockDim.x) {
      if (taken && (taken[a >> 5] & (1u << (a & 31)))) continue;
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
        if (taken && (taken[a >> 5] & (1u << (a & 31)))) continue;
        const double d0 = gx - anchors[a * 4], d1 = gy - anchors[a * 4 + 1], d2 = gw - anchors[a * 4 + 2], d3 = gh - anchors[a * 4 + 3];
        Cand t;
        t.v = ((d0 * d0 + d1 * d1) + d2 * d2) + d3 * d3;
        t.idx = a;
        if (better<false>(t, c)) c = t;
      }
      best = block_best<false>(c, red);
    }
    return best.idx;                                 // A >= number of boxes: a free anchor always exists
}

__global__ __launch_bounds__(256) void labels_zero_kernel(float* __restrict__ p0, size_t n0, float* __restrict__ p1, size_t n1,
                                                          float* __restrict__ p2, size_t n2, float* __restrict__ p3, size_t n3) {
  const size_t total = n0 + n1 + n2 + n3;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    if (i < n0) p0[i] = 0.f;
    else if (i < n0 + n1) p1[i - n0] = 0.f;
    else if (i < n0 + n1 + n2) p2[i - n0 - n1] = 0.f;
    else p3[i - n0 - n1 - n2] = 0.f;
  }
}

__global__ __launch_bounds__(256) void labels_best_kernel(const double* __restrict__ anchors, const double* __restrict__ gt,
                                                          const int* __restrict__ gt_count, int* __restrict__ aidx_out,
                                                          int A, int M) {
  __shared__ Cand red[16];
  const int i = blockIdx.x, b = blockIdx.y;
  int n = gt_count[b];
  if (n > M) n = M;
  if (i >= n) {
    if (threadIdx.x == 0) aidx_out[(size_t)b * M + i] = -1;
    return;
  }
  const int a = best_anchor(anchors, A, gt + ((size_t)b * M + i) * 4, nullptr, red);
  if (threadIdx.x == 0) aidx_out[(size_t)b * M + i] = a;
}

__global__ __launch_bounds__(1024) void labels_resolve_kernel(const double* __restrict__ anchors, const double* __restrict__ gt,
                                                              const int* __restrict__ gt_cls, const int* __restrict__ gt_count,
                                                              float* __restrict__ mask, float* __restrict__ delta,
                                                              float* __restrict__ box, float* __restrict__ labels,
                                                              int* __restrict__ aidx_out, int A, int M, int C) {
  extern __shared__ unsigned int taken[];            // A bits
  __shared__ Cand red[16];
  const int b = blockIdx.x;
  const int words = (A + 31) / 32;
  for (int i = threadIdx.x; i < words; i += blockDim.x) taken[i] = 0u;
  __syncthreads();
  int n = gt_count[b];
  if (n > M) n = M;
  int mine = -1;                                     // thread i keeps box i's anchor
  for (int i = 0; i < n; ++i) {
    int a = aidx_out[(size_t)b * M + i];             // best anchor ignoring the claims of earlier boxes
    const bool clash = a < 0 || (taken[a >> 5] & (1u << (a & 31)));   // workgroup-uniform
    if (clash) a = best_anchor(anchors, A, gt + ((size_t)b * M + i) * 4, taken, red);
    __syncthreads();                                 // everyone has read taken[] / red[]
    if (threadIdx.x == 0 && a >= 0) taken[a >> 5] |= 1u << (a & 31);
    if ((int)threadIdx.x == i) mine = a;
    __syncthreads();                                 // taken[] update visible before the next box
  }
  const int i = threadIdx.x;
  if (i < n) {
    const int a = mine;
    aidx_out[(size_t)b * M + i] = a;
    if (a >= 0) {
      const double* g = gt + ((size_t)b * M + i) * 4;
      const double gx = g[0], gy = g[1], gw = g[2], gh = g[3];
      float* m = mask + (size_t)b * A;
      float* d = delta + (size_t)b * A * 4;
      float* bx = box + (size_t)b * A * 4;
      float* lb = labels + (size_t)b * A * C;
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
  SQDET_UNSUPPORTED(max_objects > 1024, "build_labels: more than 1024 boxes per image");
  hipStream_t st = as_stream(stream);
  const size_t ba = (size_t)batch * num_anchors;
  hipLaunchKernelGGL(labels_zero_kernel, dim3(2048), dim3(256), 0, st, input_mask, ba, box_delta_input, ba * 4, box_input, ba * 4,
                     labels, ba * classes);
  hipLaunchKernelGGL(labels_best_kernel, dim3((unsigned)max_objects, (unsigned)batch), dim3(256), 0, st, anchors_f64, gt_boxes_f64,
                     gt_counts, anchor_index, num_anchors, max_objects);
  hipLaunchKernelGGL(labels_resolve_kernel, dim3((unsigned)batch), dim3(1024), lds, st, anchors_f64, gt_boxes_f64,
                     gt_classes, gt_counts, input_mask, box_delta_input, box_input, labels, anchor_index, num_anchors,
                     max_objects, classes);
  SQDET_CHECK_HIP(hipGetLastError());
  return SQDET_OK;
}
