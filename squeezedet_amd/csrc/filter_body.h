// The per-image body of filter_prediction's top-N branch (TOP_N_DETECTION <= 64 < A): shared by the stand-alone filter
// kernel (filter_fast.hip: one 1024-thread workgroup per image) and by the RIDER workgroups of the fire_chain launches
// (chain.hip: 512 threads -- the serving step's decode + filter of the previous batch on the CUs those launches leave idle).
// Replaces ModelSkeleton.filter_prediction (reference src/nn_skeleton.py:696-734) + util.nms / util.batch_iou
// (src/utils/util.py:32-76); results are identical to the generic kernel in postproc.hip (same total order, same float32
// IoU, same float64 threshold compare) -- only the selection algorithm differs:
//
//   1. every thread keeps its anchors' 32-bit order-preserving prob keys in registers and publishes its maximum;
//   2. L = the TOP_N-th largest of the 128 maxima of thread groups (each group = 128 x ... anchors), found by an all-pairs rank
//      in LDS (the group's threads share its 128 comparisons).  At least TOP_N anchors have key >= L, so every top-N anchor
//      has key >= L;
//   3. the anchors with key >= L (typically ~TOP_N..4*TOP_N of 16848) are compacted into LDS as
//      64-bit composite keys (prob key << 32 | anchor: all distinct);
//   4. all-pairs rank among the candidates: rank r < TOP_N <=> selected, and r IS the position in
//      the descending order -> no sort pass, no histogram atomics;
//   5. wave 0 decodes the <= 64 ranked boxes, all waves share the 64 x 64 IoU pairs of the non-greedy NMS (the
//      kernel is pure latency: one wave walking 63 dependent IoU chains was most of it), wave 0 emits the survivors
//      ordered by class, then descending prob (positions from per-class ballots).
// If more than FCAP anchors tie at >= L (e.g. a constant score map) the image falls back to an exact radix select
// (decided on the device, uniformly per image).  Everything here must be compiled with -ffp-contract=off.
#pragma once
#include "postproc.h"

namespace sqdet {

constexpr int FCAP = 2048;   // candidate capacity
constexpr int FG = 128;      // thread groups whose maxima bound the threshold (>= 64 needed)

template <int NT>
struct FastLds {
  unsigned int wmax[FG];
  unsigned long long cand[FCAP];
  unsigned long long sel[64];
  unsigned long long supp;   // bit r: box r is suppressed by a higher-ranked same-class box
  f32x4 box[64];
  int cls[64];
  unsigned int L;
  int count;
  int fallback;
  int hist[256], scan[256], misc[4];   // the radix-select fallback's scratch
};

// Exact radix select of the top_n-th largest composite key over ALL anchors (the generic
// algorithm, used only when the candidate list overflows).  Returns the key; all threads get it.
template <int NT>
__device__ unsigned long long slow_select(const float* probs, int A, int top_n, int* hist, int* scan, int* misc) {
  const int tid = threadIdx.x;
  unsigned long long prefix = 0;
  int remaining = top_n;
  for (int byte = 7; byte >= 0; --byte) {
    if (byte < 4 && ((unsigned int)(A - 1) >> (8 * byte)) == 0) continue;
    if (tid < 256) hist[tid] = 0;
    __syncthreads();
    const int hs = 8 * (byte + 1);
    for (int i = tid; i < A; i += NT) {
      const unsigned long long key = make_key(probs[i], i);
      const bool match = byte == 7 || (key >> hs) == (prefix >> hs);
      if (match) atomicAdd(&hist[(int)((key >> (8 * byte)) & 255)], 1);
    }
    __syncthreads();
    if (tid < 256) scan[tid] = hist[tid];
    __syncthreads();
    for (int off = 1; off < 256; off <<= 1) {
      int v = 0;
      if (tid < 256) v = tid + off < 256 ? scan[tid + off] : 0;
      __syncthreads();
      if (tid < 256) scan[tid] += v;
      __syncthreads();
    }
    if (tid < 256) {
      const int mine = scan[tid];
      const int above = tid == 255 ? 0 : scan[tid + 1];
      if (mine >= remaining && above < remaining) { misc[0] = tid; misc[1] = above; }
    }
    __syncthreads();
    prefix |= (unsigned long long)misc[0] << (8 * byte);
    remaining -= misc[1];
    __syncthreads();
  }
  return prefix;
}


// One image.  NT threads (1024 or 512: a multiple of 128 x a power of two); ME = anchors per thread held in registers
// (A <= NT * ME); a thread group = GS = NT / 128 consecutive threads.  FUSED (T = the storage type of preds): a.probs holds the
// scores (the ConvDet epilogue's or the score kernel's); wave 0 decodes boxes and classes of the <= 64 selected anchors only,
// with the float expressions of interpret_kernel (postproc.h): identical picks and outputs.  All NT threads must call it
// (barriers inside); the caller synchronises before the LDS state is reused.
template <bool FUSED, typename T, int NT>
__device__ __forceinline__ void filter_one_image(const FilterArgs& a, const DecodeArgs& d, int img, FastLds<NT>& s) {
  constexpr int ME = 20480 / NT;
  constexpr int GS = NT / FG;
  static_assert(NT % FG == 0 && (GS & (GS - 1)) == 0 && GS <= 8 && NT >= 512, "thread groups");
  const int tid = threadIdx.x;
  const float* probs = a.probs + (size_t)img * a.A;
  const T* pimg = FUSED ? reinterpret_cast<const T*>(d.preds) + (size_t)img * d.cells * d.apg * (d.C + 5) : nullptr;
  const float* boxes = a.boxes + (size_t)img * a.A * 4;
  const int64_t* cls = a.cls + (size_t)img * a.A;
  float* ob = a.out_boxes + (size_t)img * a.max_out * 4;
  float* op = a.out_probs + (size_t)img * a.max_out;
  int32_t* oc = a.out_cls + (size_t)img * a.max_out;
  int32_t* oi = a.out_index + (size_t)img * a.max_out;
  const int M = a.top_n;  // 1 <= M <= 64, M < A

  // ---- 1. keys in registers + per-thread maximum.  Slots beyond A hold key 0 and are never
  //         selected (every selection below also tests the index).
  unsigned int key[ME];
  unsigned int mx = 0;
#pragma unroll
  for (int e = 0; e < ME; ++e) {
    const int i = tid + e * NT;
    key[e] = i < a.A ? order_key32(probs[i]) : 0u;
    mx = key[e] > mx ? key[e] : mx;
  }
  // maximum of every group of 8 consecutive threads (128 disjoint groups of <= 160 anchors)
  unsigned int gmx = mx;
  {
#pragma unroll
    for (int m = 1; m < GS; m <<= 1) {
      const unsigned int o = __shfl_xor(gmx, m);
      gmx = o > gmx ? o : gmx;
    }
  }
  if ((tid % GS) == 0) s.wmax[tid / GS] = gmx;
  if (tid == 0) { s.count = 0; s.fallback = 0; }
  __syncthreads();

  // ---- 2. L = M-th largest of the 128 group maxima (all-pairs rank; ties broken by group id): the 8 threads of group
  //         tid>>3 each compare against 16 of the maxima.  Each group maximum is a distinct anchor, so >= M anchors have
  //         key >= L.
  {
    const int g = tid / GS, part = tid % GS;
    const unsigned int mine = s.wmax[g];
    int rank = 0;
#pragma unroll
    for (int u = 0; u < FG / GS; ++u) {
      const int t = part * (FG / GS) + u;
      const unsigned int o = s.wmax[t];
      rank += (o > mine || (o == mine && t < g)) ? 1 : 0;
    }
#pragma unroll
    for (int m = 1; m < GS; m <<= 1) rank += __shfl_xor(rank, m);
    if (part == 0 && rank == M - 1) s.L = mine;
  }
  __syncthreads();
  const unsigned int L = s.L;

  // ---- 3. compaction of the candidates (key >= L) ----
#pragma unroll
  for (int e = 0; e < ME; ++e) {
    const int i = tid + e * NT;
    if (i < a.A && key[e] >= L) {
      const int slot = atomicAdd(&s.count, 1);
      if (slot < FCAP) s.cand[slot] = ((unsigned long long)key[e] << 32) | (unsigned int)i;
    }
  }
  __syncthreads();
  int C = s.count;
  if (C > FCAP) {
    // too many ties at the boundary: exact radix select over all anchors, then re-compact
    const unsigned long long TH = slow_select<NT>(probs, a.A, M, s.hist, s.scan, s.misc);
    if (tid == 0) s.count = 0;
    __syncthreads();
#pragma unroll
    for (int e = 0; e < ME; ++e) {
      const int i = tid + e * NT;
      const unsigned long long k64 = ((unsigned long long)key[e] << 32) | (unsigned int)i;
      if (i < a.A && k64 >= TH) {
        const int slot = atomicAdd(&s.count, 1);
        if (slot < FCAP) s.cand[slot] = k64;
      }
    }
    __syncthreads();
    C = s.count;  // == M
  }

  // ---- 4. all-pairs rank among the candidates: rank < M <=> in the top-N, rank = position.  8 threads per candidate,
  //         each against every 8th key ----
  for (int q0 = 0; q0 < C; q0 += NT / 8) {
    const int q = q0 + (tid >> 3), part = tid & 7;
    const unsigned long long mine = q < C ? s.cand[q] : 0ull;
    int rank = 0;
    for (int t = part; t < C; t += 8) rank += s.cand[t] > mine ? 1 : 0;
    rank += __shfl_xor(rank, 1);
    rank += __shfl_xor(rank, 2);
    rank += __shfl_xor(rank, 4);
    if (part == 0 && q < C && rank < M) s.sel[rank] = mine;
  }
  __syncthreads();
  // ---- 5. wave 0 decodes / fetches the <= 64 ranked boxes; the 64 x 64 IoU pairs of the NMS are dealt over all 16
  //         waves (lane r of wave w: box r against boxes 4w..4w+3 -- one wave doing all 63 dependent IoU chains was 13 of
  //         the kernel's 20 us); wave 0 emits the survivors ordered by class, then rank ----
  const int r = tid & 63, wv = tid >> 6;
  int idx = 0, c = -1;
  f32x4 bj = {0.f, 0.f, 0.f, 0.f};
  float pj = 0.f;
  if (wv == 0) {
    if (r < M) {
      idx = (int)(s.sel[r] & 0xffffffffull);
      if constexpr (FUSED) {
        const int cell = idx / d.apg, k = idx - cell * d.apg;
        const T* p = pimg + (size_t)cell * d.apg * (d.C + 5);
        pj = decode_score<T>(p, k, d.apg, d.C, &c);
        bj = decode_box<T>(p, k, d.apg, d.C, *reinterpret_cast<const f32x4*>(d.anchors + (size_t)idx * 4), d.w1, d.h1, d.thr, d.slope);
      } else {
        bj = *reinterpret_cast<const f32x4*>(boxes + (size_t)idx * 4);
        c = (int)cls[idx];
        pj = probs[idx];
      }
    }
    s.box[r] = bj;
    s.cls[r] = c;
    if (r == 0) s.supp = 0ull;
  }
  __syncthreads();
  // the reference's non-greedy NMS (utils/util.py:56-76): r is dropped iff ANY higher-ranked
  // same-class box has IoU > threshold (compared in float64, as under the reference's NumPy 1.12)
  {
    const f32x4 br = s.box[r];
    const int cr = s.cls[r];
    bool sup = false;
#pragma unroll
    for (int u = 0; u < 64 / (NT / 64); ++u) {
      const int i = wv * (64 / (NT / 64)) + u;
      const float ov = iou_center(br, s.box[i]);
      if (i < r && r < M && s.cls[i] == cr && (double)ov > a.nms_thresh) sup = true;
    }
    const unsigned long long m = __ballot(sup);
    if (r == 0 && m) atomicOr(&s.supp, m);
  }
  __syncthreads();
  if (wv == 0) {
  const bool keep = r < M && c >= 0 && c < a.C && !((s.supp >> r) & 1ull);
  // output position: kept entries ordered by class, then rank (nn_skeleton.py:726-733)
  const unsigned long long km = __ballot(keep);
  const int kept = __popcll(km);
  int pos = 0;
  {
    const unsigned long long below = r == 0 ? 0ull : (~0ull >> (64 - r));
    for (int cc = 0; cc < a.C; ++cc) {   // uniform trip count
      const unsigned long long mc = __ballot(c == cc) & km;
      pos += cc < c ? __popcll(mc) : (cc == c ? __popcll(mc & below) : 0);
    }
  }
  if (keep) {
    *reinterpret_cast<f32x4*>(ob + (size_t)pos * 4) = bj;
    op[pos] = pj;
    oc[pos] = c;
    oi[pos] = idx;
  }
  for (int o = kept + r; o < a.max_out; o += 64) {
    *reinterpret_cast<f32x4*>(ob + (size_t)o * 4) = f32x4{0.f, 0.f, 0.f, 0.f};
    op[o] = 0.f;
    oc[o] = -1;
    oi[o] = -1;
  }
  if (r == 0) a.out_count[img] = kept;
  }
}

}  // namespace sqdet
