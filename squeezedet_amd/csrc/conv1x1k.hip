// Deep-K 1x1 convolution (+bias +ReLU) as a persistent STREAMING kernel for gfx950: the squeeze1x1 layers of fire5..fire11
// (K = 256..768 channels -> 32..96 couts) and the other deep 1x1s of SqueezeDet+ / ResNet50 (reference src/nets/squeezeDet.py:95-100,
// src/nn_skeleton.py:471-563).  These layers are purely HBM-bound -- 10-30 FLOP per byte, the matrix pipe has an hour to spare -- and
// conv1x1_tile (gemm1x1.hip), which stages the ACTIVATIONS through LDS in 4-chunk stages behind barriers, ran them at about half of what a
// plain copy of the same bytes reaches (profiles/r03_fire_1x1_standalone.txt).  Here the roles are swapped:
//   * the WEIGHTS of the workgroup's cout group live in LDS for the kernel's life ([K chunk][tile][64 lanes][16 B]: <= 64 KiB), read as
//     A fragments per MFMA (1 KiB per MFMA: the LDS has bandwidth to burn at this arithmetic intensity);
//   * the activations are read EXACTLY ONCE, straight into B-fragment registers: a wave owns one 16-pixel block at a time and issues
//     its loads in groups of KB = 8 chunks -- eight 16-byte loads in flight per lane, 8 KiB per wave, no LDS staging, no barrier in the
//     loop; the next group's loads are issued before the current group's MFMAs (two register sets), the first group's before the
//     weight copy; with 4 waves per SIMD the other waves' MFMAs and stores hide the rest of the latency;
//   * persistent grid: ONE 16-wave workgroup per CU (the weight image is copied once per CU), waves grid-stride over the 16-pixel blocks,
//     one cout group per workgroup.
// Accumulation order per output = chunk ascending, as conv1x1_tile and the generic kernel: bitwise the same results.
#include "conv_common.h"

namespace sqdet {
namespace {

struct K1Args {
  ConvArgs c;
  int nblocks;     // 16-pixel blocks
  int nchunk;      // 64-byte K chunks
  int wgs_per_group;
  unsigned x_bytes, y_bytes;   // extents of the two tensors (buffer resources: offsets beyond them read zeros / store nothing)
};

constexpr int KB = 8;    // chunks per load group

template <typename T, int NT, int NWAVES>
__global__ __launch_bounds__(NWAVES * 64) void conv1x1_deepk(K1Args a) {
  constexpr int KG = Tr<T>::KG;
  constexpr int KC = 4 * KG;
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int j = lane & 15, g = lane >> 4;
  const int group = blockIdx.x / a.wgs_per_group, wg = blockIdx.x - group * a.wgs_per_group;
  const int nchunk = a.nchunk;
  const int stride = a.wgs_per_group * NWAVES;
  // Loads and stores are raw buffer operations with an out-of-range offset where the old code had a branch (pixels past the end, K
  // chunks past the last one, the channel padding of the last chunk, groups past the wave's last block): the tile loop is straight-line
  // code around ONE wave-uniform branch (the epilogue) and hipcc's wait-count pass counts -- `vmcnt(8 + stores)` in front of a group's
  // MFMAs.  With exec-masked loads in blocks of their own it put `s_waitcnt vmcnt(0)` in front of every MFMA group: the "two groups in
  // flight" were one, and a wave waited for its own stores (rounds 4: 0.36-0.60 of 8 TB/s stand-alone).
  const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.c.x), 0, a.x_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc(a.c.y, 0, a.y_bytes, 0x00020000);
  constexpr unsigned OOB = 0xfffffff0u;
  constexpr unsigned ES = sizeof(T);
  const unsigned xrow = (unsigned)a.c.Cin * ES, yrow = (unsigned)a.c.y_cstride * ES;

  // a load group = KB consecutive K chunks of one 16-pixel block; groups are walked block-major, the NEXT group's loads are issued
  // before the current group's MFMAs (two register sets)
  auto issue = [&](int blk, int c0, i32x4 (&bf)[KB]) {
    const int p = blk * 16 + j;
    const unsigned pb = p < a.c.P ? (unsigned)p * xrow + (unsigned)(g * KG) * ES : OOB;
#pragma unroll
    for (int u = 0; u < KB; ++u) {
      const int c = c0 + u;
      // (chunks past the last one and the channel padding of the last chunk read as zeros; the packed weights are zero there too)
      bf[u] = __builtin_amdgcn_raw_buffer_load_b128(rx, (pb != OOB && c * KC + g * KG < a.c.Cin) ? pb + (unsigned)(c * KC) * ES : OOB, 0, 0);
    }
  };
  // two pointers walk the same sequence of groups: (ib, ic) = the next group to REQUEST, (cb_, cc_) = the next group to COMPUTE; two
  // groups (16 loads per lane) are in flight ahead of the compute pointer, the first two ahead of the weight copy
  int ib = wg * NWAVES + wave, ic = 0;
  int cb_ = ib, cc_ = 0;
  auto advance_issue = [&]() {
    const bool last = ic + KB >= nchunk;
    ib = last ? ib + stride : ib;
    ic = last ? 0 : ic + KB;
  };
  i32x4 bfa[KB], bfb[KB];
  issue(ib, ic, bfa); advance_issue();
  issue(ib, ic, bfb); advance_issue();

  // ---- the group's weights -> LDS (once) ----
  {
    const i32x4* src = reinterpret_cast<const i32x4*>(a.c.wp) + (size_t)group * nchunk * NT * 64;
    for (int i = threadIdx.x; i < nchunk * NT * 64; i += NWAVES * 64) reinterpret_cast<i32x4*>(lds)[i] = src[i];
  }
  const int cb = group * 16 * NT + g * 4 * NT;          // this lane's 4*NT consecutive couts
  f32x4 bias[NT];
  int nt_valid = 0;
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    const bool ok = cb + t * 4 < a.c.Cout;
    bias[t] = ok ? *reinterpret_cast<const f32x4*>(a.c.bias + cb + t * 4) : f32x4{0.f, 0.f, 0.f, 0.f};
    nt_valid += ok ? 1 : 0;
  }
  // 16-byte stores of float16 tile pairs need whole cout groups and 16-byte aligned row segments (wave-uniform: the group's first cout)
  const bool wide = sizeof(T) == 2 && (NT & 1) == 0 && (group + 1) * 16 * NT <= a.c.Cout &&
                    ((reinterpret_cast<uintptr_t>(a.c.y) + ((size_t)a.c.y_coffset + (size_t)group * 16 * NT) * ES) & 15) == 0 && (yrow & 15) == 0;
  __syncthreads();

  const unsigned char* wl = lds + lane * 16;
  f32x4 acc[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
  auto step = [&](i32x4 (&cur)[KB]) {
    const bool last = cc_ + KB >= nchunk;
#pragma unroll
    for (int u = 0; u < KB; ++u) {
      // a group's chunks past the last one multiply ZERO fragments (out-of-range loads) with the last chunk's weights: + 0, no branch
      const int cw = cc_ + u < nchunk ? cc_ + u : nchunk - 1;         // wave-uniform
#pragma unroll
      for (int t = 0; t < NT; ++t) mma16<T>(acc[t], *reinterpret_cast<const i32x4*>(wl + (cw * NT + t) * 1024), cur[u]);
    }
    // the registers just consumed take the group two ahead (past the wave's last block: out of range, nothing is fetched)
    issue(ib, ic, cur); advance_issue();
    if (last) {
      const int p = cb_ * 16 + j;
      const unsigned po = p < a.c.P ? (unsigned)p * yrow + (unsigned)(a.c.y_coffset + cb) * ES : OOB;
      f32x4 v[NT];
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        v[t] = acc[t] + bias[t];
        if (a.c.relu) {
          v[t][0] = fmaxf(v[t][0], 0.f); v[t][1] = fmaxf(v[t][1], 0.f);
          v[t][2] = fmaxf(v[t][2], 0.f); v[t][3] = fmaxf(v[t][3], 0.f);
        }
      }
      if (wide) {
#pragma unroll
        for (int t = 0; t + 1 < NT; t += 2) {
          const f16x8 h = {(f16)v[t][0], (f16)v[t][1], (f16)v[t][2], (f16)v[t][3],
                           (f16)v[t + 1][0], (f16)v[t + 1][1], (f16)v[t + 1][2], (f16)v[t + 1][3]};
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(i32x4, h), ry, po != OOB ? po + (unsigned)(t * 4) * ES : OOB, 0, 0);
        }
      } else {
#pragma unroll
        for (int t = 0; t < NT; ++t) {
          const unsigned off = (po != OOB && t < nt_valid) ? po + (unsigned)(t * 4) * ES : OOB;
          if constexpr (sizeof(T) == 2) {
            const f16x4 h = {(f16)v[t][0], (f16)v[t][1], (f16)v[t][2], (f16)v[t][3]};
            __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(i32x2, h), ry, off, 0, 0);
          } else {
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(i32x4, v[t]), ry, off, 0, 0);
          }
        }
      }
#pragma unroll
      for (int t = 0; t < NT; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    cb_ = last ? cb_ + stride : cb_;
    cc_ = last ? 0 : cc_ + KB;
  };
  // (a trip's second step past the wave's last block computes on zeros and stores nothing)
  while (cb_ < a.nblocks) {
    step(bfa);
    step(bfb);
  }
}

template <typename T, int NT>
bool launch_k1(K1Args& a, int ngroups, hipStream_t st) {
  // ONE 16-wave workgroup per CU: the weight image is copied once per CU, four waves per SIMD keep >= 16 KiB of loads per SIMD in flight
  constexpr int NWAVES = 16;
  const size_t lds = (size_t)a.nchunk * NT * 1024;
  auto kern = &conv1x1_deepk<T, NT, NWAVES>;
  static PerDevice once;
  (void)once.run([&] { return hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); });
  // persistent grid: as many workgroups per CU as are RESIDENT together (registers, the LDS weight image, 32 waves), split over the
  // cout groups
  // (asked once per LDS size and instantiation: the query costs tens of microseconds of host time)
  // (keyed by LDS size and device: slot key = lds * 64 + device + 1)
  // (one 64-bit word per slot = key << 8 | answer, relaxed atomics: host threads may launch concurrently; a full table evicts round-robin)
  static std::atomic<uint64_t> slots[8];
  static std::atomic<unsigned> victim{0};
  const uint64_t lds_key = (uint64_t)lds * 64 + (uint64_t)(current_device() & 63) + 1;
  int per_cu = 0;
  for (int i = 0; i < 8; ++i) {
    const uint64_t v = slots[i].load(std::memory_order_relaxed);
    if ((v >> 8) == lds_key) per_cu = (int)(v & 0xff);
  }
  if (per_cu == 0) {
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, reinterpret_cast<const void*>(kern), NWAVES * 64, lds) != hipSuccess || per_cu < 1) {
      (void)hipGetLastError();
      return false;
    }
    if (per_cu > 255) per_cu = 255;
    int slot = -1;
    for (int i = 0; i < 8 && slot < 0; ++i)
      if (slots[i].load(std::memory_order_relaxed) == 0) slot = i;
    if (slot < 0) slot = (int)(victim.fetch_add(1, std::memory_order_relaxed) & 7);
    slots[slot].store(lds_key << 8 | (uint64_t)per_cu, std::memory_order_relaxed);
  }
  int wgs = cu_count() * per_cu / ngroups;
  const int need = (a.nblocks + NWAVES - 1) / NWAVES;
  if (wgs > need) wgs = need;
  if (wgs < 1) wgs = 1;
  a.wgs_per_group = wgs;
  hipLaunchKernelGGL(kern, dim3((unsigned)(wgs * ngroups)), dim3(NWAVES * 64), lds, st, a);
  return true;
}

template <typename T>
bool dispatch_k1(K1Args& a, int nt, int ngroups, hipStream_t st) {
  switch (nt) {
    case 1: return launch_k1<T, 1>(a, ngroups, st);
    case 2: return launch_k1<T, 2>(a, ngroups, st);
    case 3: return launch_k1<T, 3>(a, ngroups, st);
    case 4: return launch_k1<T, 4>(a, ngroups, st);
    case 5: return launch_k1<T, 5>(a, ngroups, st);
    case 6: return launch_k1<T, 6>(a, ngroups, st);
    default: return false;
  }
}

}  // namespace

// Plain (no channel slice, no accumulate, no mask) REDUCING (Cin >= 2 Cout) 1x1 / stride-1 convs with 5..48 K chunks whose cout group's
// weights fit 152 KiB of LDS and that have enough pixels to stream (>= 8192).  *handled = false: conv1x1_pipe / conv1x1_tile / the
// generic kernel take it.  ("dbg" 51 / 57 / 61..79: never)
int conv1x1_deepk_launch(const ConvArgs& c, const ConvGeom& g, int dtype, hipStream_t st, bool* handled) {
  *handled = false;
  if (conv_algo() != 0 || tune(TUNE_DBG) == 51 || tune(TUNE_DBG) == 57 || (tune(TUNE_DBG) >= 61 && tune(TUNE_DBG) <= 79)) return SQDET_OK;
  if (c.k != 1 || c.stride != 1 || c.pt != 0 || c.pl != 0 || g.gather || c.accum || c.relu_of) return SQDET_OK;
  const int esz = dtype == SQDET_F16 ? 2 : 4;
  if (c.x_coffset != 0 || c.x_cstride != c.Cin || (c.Cin * esz) % 16 != 0) return SQDET_OK;
  // ("dbg" 53 / 54: in-step A/B of the pixel threshold -- 20000 / 65536 instead of 8192; stand-alone the tile kernel wins on most maps
  //  below 60 k pixels, profiles/r04_conv1x1_shapes_ab.txt, inside the steps this kernel does: profiles/r05_conv1x1_threshold_ab.txt)
  const int min_pixels = tune(TUNE_DBG) == 53 ? 20000 : (tune(TUNE_DBG) == 54 ? 65536 : 8192);
  if (g.nchunk < 5 || g.nchunk > 48 || (size_t)g.nchunk * g.nt * 1024 > 152 * 1024 || c.P < min_pixels) return SQDET_OK;
  // REDUCING convs only (Cin >= 2 Cout: the fire modules' squeeze1x1, ResNet50's branch2a): every cout group's workgroups stream all
  // the pixels, so an expanding conv re-reads its input once per group (256 -> 1024: 16 times).  Stand-alone the tile kernel wins every
  // shape with Cout > Cin / 2 but one (profiles/r05_conv1x1_shapes_ab.txt: 1.06-1.20x), in-step SqueezeDet+ -- whose deep 1x1s all are
  // such -- runs 3.3 % faster without this kernel (profiles/r05_conv1x1_threshold_ab.txt).  ("dbg" 56: the round-4 rule, any ratio)
  if (2 * c.Cout > c.Cin && tune(TUNE_DBG) != 56) return SQDET_OK;
  const size_t xb = (size_t)c.P * c.Cin * esz, yb = (size_t)c.P * c.y_cstride * esz;
  if (xb >= (1ull << 31) || yb >= (1ull << 31)) return SQDET_OK;   // 32-bit buffer offsets
  K1Args a;
  a.c = c;
  a.x_bytes = (unsigned)xb; a.y_bytes = (unsigned)yb;
  a.nblocks = (c.P + 15) / 16;
  a.nchunk = g.nchunk;
  const bool ok = dtype == SQDET_F16 ? dispatch_k1<f16>(a, g.nt, g.ngroups, st) : dispatch_k1<float>(a, g.nt, g.ngroups, st);
  if (!ok) return SQDET_OK;
  SQDET_CHECK_HIP(hipGetLastError());
  *handled = true;
  return SQDET_OK;
}

}  // namespace sqdet
