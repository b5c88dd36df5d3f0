// Conv backward-filter (+ bias gradient) for gfx950:
//   dW[ty][tx][ci][co] = sum_p X[p @ (ty,tx)][ci] * dY[p][co]        (stride-1 SAME convs, k = 1 or 3)
//   dbias[co]          = sum_p dY[p][co]
// (the gradients tf.gradients produces for tf.nn.conv2d + tf.nn.bias_add, nn_skeleton.py:539-542, which
// ModelSkeleton._add_train_graph feeds to the optimizer, nn_skeleton.py:343-349).  X / dY are float32 (the
// reference's training dtype; exact-f32 MFMA v_mfma_f32_16x16x4_f32) or float16 (mixed-precision training;
// v_mfma_f32_16x16x32_f16); dW / dbias are always accumulated and written in float32.
//
// A GEMM with M = ci, N = co and K = pixels.  One workgroup owns a (ci tile) x (co tile) block of dW for ALL the taps
// of its tap group (1 tap of a 1x1, one kernel row or all nine taps of a 3x3), so X and dY are read once per tap group
// instead of once per tap.  K is walked in stages of 4 rows x 16 columns of one image: the stage's dY tile and X tile
// (with the 1-pixel halo the taps need; TF SAME zero padding by out-of-range buffer loads, which return zeros) go
// global -> registers -> LDS one stage ahead of the MFMAs, and every tap's A fragment is the same LDS image read at a
// shifted pixel offset.  Both operands are K-major in the MFMA (a lane holds consecutive PIXELS of one channel) while
// the LDS image is [pixel][channel] as it arrives from HBM: float32 reads one element per lane (ds_read_b32), float16
// uses the gfx950 transpose read (ds_read_b64_tr_b16: a 16-lane group fetches a [4 pixels][16 channels] block and each
// lane receives one channel's 4 pixels).  blockIdx.z owns every ksplit-th stage and writes a partial slab;
// slab_reduce2 sums the slabs in a fixed order, so the result is deterministic (no atomics).  The bias gradient rides
// along as one more MFMA per k-step with an all-ones A operand (row 0 of the product is the column sum of dY), in the
// waves that own ci tile 0.
#include <type_traits>

#include "conv_common.h"

namespace sqdet {

struct WgArgs {
  const void* x;
  const void* dy;
  float* partial;
  int H, W, Cin, Cout, k;
  int x_cstride, x_coffset, dy_cstride, dy_coffset;
  int BY, BX, nstages, ksplit, ci_tiles, do_bias;
  unsigned x_bytes, dy_bytes;
  size_t slab_stride;  // floats per slab = k*k*Cin*Cout + Cout
};

constexpr unsigned kOOB = 0xfffffff0u;
typedef short s16x4 __attribute__((ext_vector_type(4)));

template <typename T, int MT, int NT, int WM, int TAPS>
__global__ __launch_bounds__(256, 2) void wgrad_tile(WgArgs a) {
  constexpr bool HALF = sizeof(T) == 2;
  constexpr int EV = 16 / sizeof(T);               // elements per 16-byte load
  constexpr int KSTEPS = HALF ? 2 : 16;            // MFMA k-steps per 64-pixel stage (K = 32 / 4 pixels)
  constexpr int WN = 4 / WM;
  constexpr int MTILE = WM * MT * 16, NTILE = WN * NT * 16;
  constexpr int MV = MTILE / EV, NV = NTILE / EV;
  constexpr int HR = TAPS == 9 ? 6 : 4, HC = TAPS == 1 ? 16 : 18;
  // row pitches (elements) that are odd multiples of 16: the pixel rows one LDS read touches land in disjoint banks
  constexpr int XS = ((MTILE / 16) & 1) ? MTILE : MTILE + 16;
  constexpr int DS = ((NTILE / 16) & 1) ? NTILE : NTILE + 16;
  constexpr int XN = HR * HC * MV, DN = 64 * NV;
  constexpr int NXL = (XN + 255) / 256, NDL = (DN + 255) / 256;
  constexpr int TY = TAPS == 9 ? 3 : 1, TX = TAPS == 1 ? 1 : 3;
  __shared__ __attribute__((aligned(16))) T xs[HR * HC * XS];
  __shared__ __attribute__((aligned(16))) T dsm[64 * DS];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave % WM, wn = wave / WM;
  const int i = lane & 15, kk = lane >> 4;
  const int tg = blockIdx.x / a.ci_tiles;           // tap group: the kernel row when TAPS == 3
  const int ci_tile = blockIdx.x - tg * a.ci_tiles;
  const int ci0 = ci_tile * MTILE, co0 = blockIdx.y * NTILE;
  const int py = TAPS == 9 ? 1 : (TAPS == 3 ? 1 - tg : 0), px = TAPS == 1 ? 0 : 1;
  const bool bias_wave = a.do_bias && ci_tile == 0 && tg == 0 && wm == 0;

  const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.x), 0, a.x_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rd = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.dy), 0, a.dy_bytes, 0x00020000);

  f32x4 acc[TAPS][MT][NT], accb[NT];
#pragma unroll
  for (int t = 0; t < TAPS; ++t)
#pragma unroll
    for (int mi = 0; mi < MT; ++mi)
#pragma unroll
      for (int ni = 0; ni < NT; ++ni) acc[t][mi][ni] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int ni = 0; ni < NT; ++ni) accb[ni] = f32x4{0.f, 0.f, 0.f, 0.f};

  i32x4 xr[NXL], dr[NDL];
  auto issue = [&](int s) {
    const int n = s / (a.BY * a.BX);
    const int rem = s - n * (a.BY * a.BX);
    const int by = rem / a.BX;
    const int y0 = by * 4, x0 = (rem - by * a.BX) * 16;
#pragma unroll
    for (int j = 0; j < NXL; ++j) {
      const int idx = tid + 256 * j;
      const int pix = idx / MV, cv = idx - pix * MV;
      const int hr = pix / HC, hc = pix - hr * HC;
      const int iy = y0 + hr - py, ix = x0 + hc - px, ci = ci0 + EV * cv;
      const bool ok = idx < XN && (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W && ci < a.Cin;
      const unsigned off = ((unsigned)((n * a.H + iy) * a.W + ix) * (unsigned)a.x_cstride + (unsigned)(a.x_coffset + ci)) * (unsigned)sizeof(T);
      xr[j] = __builtin_amdgcn_raw_buffer_load_b128(rx, ok ? off : kOOB, 0, 0);
    }
#pragma unroll
    for (int j = 0; j < NDL; ++j) {
      const int idx = tid + 256 * j;
      const int pp = idx / NV, cv = idx - pp * NV;
      const int y = y0 + (pp >> 4), xx = x0 + (pp & 15), co = co0 + EV * cv;
      const bool ok = idx < DN && y < a.H && xx < a.W && co < a.Cout;
      const unsigned off = ((unsigned)((n * a.H + y) * a.W + xx) * (unsigned)a.dy_cstride + (unsigned)(a.dy_coffset + co)) * (unsigned)sizeof(T);
      dr[j] = __builtin_amdgcn_raw_buffer_load_b128(rd, ok ? off : kOOB, 0, 0);
    }
  };
  auto commit = [&]() {
#pragma unroll
    for (int j = 0; j < NXL; ++j) {
      const int idx = tid + 256 * j;
      const int pix = idx / MV, cv = idx - pix * MV;
      if (XN % 256 == 0 || idx < XN) *reinterpret_cast<i32x4*>(&xs[pix * XS + EV * cv]) = xr[j];
    }
#pragma unroll
    for (int j = 0; j < NDL; ++j) {
      const int idx = tid + 256 * j;
      const int pp = idx / NV, cv = idx - pp * NV;
      if (DN % 256 == 0 || idx < DN) *reinterpret_cast<i32x4*>(&dsm[pp * DS + EV * cv]) = dr[j];
    }
  };

  // Per-lane LDS bases.  float32: lane (i, kk) reads channel i of pixel 4*ks + kk.  float16: k = 8*kk + e of a k-step
  // is stage pixel 32*ks + 8*kk + e = (row 2*ks + (kk >> 1), column 8*(kk & 1) + e); a transpose read serves 4 of
  // them, the lane addressing pixel (i >> 2) of the four and channels 4*(i & 3).. of its 16-channel tile.
  const T* xa = HALF ? xs + ((kk >> 1) * HC + 8 * (kk & 1) + (i >> 2)) * XS + wm * (MT * 16) + 4 * (i & 3)
                     : xs + kk * XS + wm * (MT * 16) + i;
  const T* da = HALF ? dsm + (8 * kk + (i >> 2)) * DS + wn * (NT * 16) + 4 * (i & 3) : dsm + kk * DS + wn * (NT * 16) + i;

  // one k-step's operands: the dY fragment of every co tile and the X fragment of every (tap, ci tile); fetched from
  // LDS one k-step ahead of the MFMAs that use them (the scheduling fences keep the compiler from sinking the reads
  // back to their first use, where every MFMA group would wait out an LDS round trip)
  using Frag = typename std::conditional<HALF, f16x8, float>::type;
  Frag bfr[2][NT], afr[2][TAPS][MT];
  auto tr8 = [&](const T* p0, const T* p1) {   // two transpose reads -> the 8 consecutive-k halves of one lane
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(p0));
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(p1));
    typedef short s16x8 __attribute__((ext_vector_type(8)));
    return __builtin_bit_cast(f16x8, s16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]});
  };
  auto fetch = [&](int ks, int slot) {
    if constexpr (HALF) {
#pragma unroll
      for (int ni = 0; ni < NT; ++ni) bfr[slot][ni] = tr8(da + (32 * ks) * DS + ni * 16, da + (32 * ks + 4) * DS + ni * 16);
#pragma unroll
      for (int ty = 0; ty < TY; ++ty)
#pragma unroll
        for (int tx = 0; tx < TX; ++tx)
#pragma unroll
          for (int mi = 0; mi < MT; ++mi) {
            const T* p = xa + ((2 * ks + ty) * HC + tx) * XS + mi * 16;
            afr[slot][ty * TX + tx][mi] = tr8(p, p + 4 * XS);
          }
    } else {
      const int r = ks >> 2, c0 = 4 * (ks & 3);
#pragma unroll
      for (int ni = 0; ni < NT; ++ni) bfr[slot][ni] = da[(ks * 4) * DS + ni * 16];
#pragma unroll
      for (int ty = 0; ty < TY; ++ty)
#pragma unroll
        for (int tx = 0; tx < TX; ++tx)
#pragma unroll
          for (int mi = 0; mi < MT; ++mi) afr[slot][ty * TX + tx][mi] = xa[((r + ty) * HC + c0 + tx) * XS + mi * 16];
    }
  };
  auto mma = [&](Frag av, Frag bv, f32x4 c) {
    if constexpr (HALF) return __builtin_amdgcn_mfma_f32_16x16x32_f16(av, bv, c, 0, 0, 0);
    else return __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, c, 0, 0, 0);
  };
  Frag ones;
  if constexpr (HALF) ones = f16x8{(f16)1, (f16)1, (f16)1, (f16)1, (f16)1, (f16)1, (f16)1, (f16)1};
  else ones = 1.0f;
  auto compute = [&](auto with_bias) {
    fetch(0, 0);
#pragma unroll
    for (int ks = 0; ks < KSTEPS; ++ks) {
      const int cur = ks & 1;
      if (ks + 1 < KSTEPS) fetch(ks + 1, cur ^ 1);
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (decltype(with_bias)::value) {
#pragma unroll
        for (int ni = 0; ni < NT; ++ni) accb[ni] = mma(ones, bfr[cur][ni], accb[ni]);
      }
#pragma unroll
      for (int t = 0; t < TAPS; ++t)
#pragma unroll
        for (int mi = 0; mi < MT; ++mi)
#pragma unroll
          for (int ni = 0; ni < NT; ++ni) acc[t][mi][ni] = mma(afr[cur][t][mi], bfr[cur][ni], acc[t][mi][ni]);
      __builtin_amdgcn_sched_barrier(0);
    }
  };

  int s = blockIdx.z;
  if (s < a.nstages) {
    issue(s);
    commit();
  }
  __syncthreads();
  for (; s < a.nstages; s += a.ksplit) {
    const bool more = s + a.ksplit < a.nstages;
    if (more) issue(s + a.ksplit);
    if (bias_wave) compute(std::true_type{});
    else compute(std::false_type{});
    __syncthreads();
    if (more) commit();
    __syncthreads();
  }

  // D layout: col (co) = lane & 15, row (ci) = 4 * (lane >> 4) + reg
  float* out = a.partial + (size_t)blockIdx.z * a.slab_stride;
#pragma unroll
  for (int t = 0; t < TAPS; ++t) {
    const int tap = TAPS == 3 ? tg * 3 + t : t;
#pragma unroll
    for (int mi = 0; mi < MT; ++mi)
#pragma unroll
      for (int ni = 0; ni < NT; ++ni) {
        const int co = co0 + (wn * NT + ni) * 16 + i;
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
          const int ci = ci0 + (wm * MT + mi) * 16 + kk * 4 + rr;
          if (ci < a.Cin && co < a.Cout) out[((size_t)tap * a.Cin + ci) * a.Cout + co] = acc[t][mi][ni][rr];
        }
      }
  }
  if (bias_wave && kk == 0) {
#pragma unroll
    for (int ni = 0; ni < NT; ++ni) {
      const int co = co0 + (wn * NT + ni) * 16 + i;
      if (co < a.Cout) out[(size_t)a.k * a.k * a.Cin * a.Cout + co] = accb[ni][0];
    }
  }
}

// dw[e] = scale * sum_z partial[z][e] (+ decay * w[e]), dbias[c] = scale * sum_z partial[z][count + c], in a FIXED order:
// deterministic.  scale undoes the loss scaling of mixed-precision training (1 otherwise).
// Four lanes share an element (lane q of the group takes slabs q, q + 4, ...), eight loads in flight each, then two xor
// steps: ((q0 + q1) + (q2 + q3)).  (One thread per element with four loads in flight walked up to 128 dependent L2 round
// trips: 15 us per launch, 31 launches = 0.48 ms of a 3.8 ms mixed-precision SqueezeDet step.)
__device__ __forceinline__ void slab_reduce_body(const float* __restrict__ partial, float* __restrict__ dw, float* __restrict__ dbias,
                                                 const float* __restrict__ w, float decay, float scale, size_t count, int cout,
                                                 size_t stride, int nslabs, unsigned block, unsigned nblocks, int wide_ok) {
  const size_t total = count + (dbias ? (size_t)cout : 0);
  // Large gradients with few slabs (ResNet50's res4: 0.26 - 0.59 M elements, 8 - 16 slabs; the 4-lane form above reads 64-byte
  // pieces of four slabs per wave instruction and ran this reduction at 1.7 TB/s): one thread per FOUR consecutive elements, all
  // slabs as independent 16-byte loads, combined in exactly the order of the 4-lane form (lane q's slabs q, q + 4, ..; then
  // (q0 + q1) + (q2 + q3)) -- the same bits.
  if (wide_ok && nslabs <= 16 && total >= 65536 && ((count | stride | (size_t)cout) & 3) == 0 && (reinterpret_cast<uintptr_t>(partial) & 15) == 0) {
    const size_t n4 = total >> 2, nthr4 = (size_t)nblocks * blockDim.x;
    for (size_t e4 = (size_t)block * blockDim.x + threadIdx.x; e4 < n4; e4 += nthr4) {
      const size_t e = e4 << 2;
      f32x4 p[16];
#pragma unroll
      for (int z = 0; z < 16; ++z)
        p[z] = z < nslabs ? *reinterpret_cast<const f32x4*>(partial + (size_t)z * stride + e) : f32x4{0.f, 0.f, 0.f, 0.f};
      const bool is_w = e < count;
      float* dst = is_w ? dw + e : dbias + (e - count);
      float out[4];
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        float sq[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          // (absent slabs leave the 4-lane form's accumulators at +0: `0.f + x` and `+ 0.f` are kept, they turn -0 into +0)
          const float a0 = 0.f + p[q][c];
          const float a1 = q + 4 < nslabs ? 0.f + p[q + 4][c] : 0.f;
          const float a2 = q + 8 < nslabs ? 0.f + p[q + 8][c] : 0.f;
          const float a3 = q + 12 < nslabs ? 0.f + p[q + 12][c] : 0.f;
          sq[q] = ((a0 + a1) + (a2 + a3)) + 0.f;
          if (q >= nslabs) sq[q] = 0.f;
        }
        float s = (sq[0] + sq[1]) + (sq[2] + sq[3]);
        s *= scale;
        if (is_w && w) s += decay * w[e + c];
        out[c] = s;
      }
      if ((reinterpret_cast<uintptr_t>(dst) & 15) == 0) {
        *reinterpret_cast<f32x4*>(dst) = f32x4{out[0], out[1], out[2], out[3]};
      } else {
        dst[0] = out[0]; dst[1] = out[1]; dst[2] = out[2]; dst[3] = out[3];
      }
    }
    return;
  }
  const int q = threadIdx.x & 3;
  const size_t nthr = (size_t)nblocks * (blockDim.x >> 2);
  // every lane of a 4-lane group runs the same trip count (the shuffles below are wave-wide): elements rounded up per group
  for (size_t e0 = (size_t)block * (blockDim.x >> 2) + (threadIdx.x >> 2); e0 < ((total + nthr - 1) / nthr) * nthr; e0 += nthr) {
    const bool live = e0 < total;
    const size_t e = live ? e0 : 0;
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    int z = q;
    for (; z + 28 < nslabs; z += 32) {
#pragma unroll
      for (int u = 0; u < 8; ++u) acc[u] += partial[(size_t)(z + 4 * u) * stride + e];
    }
#pragma unroll
    for (int u = 0; u < 8; ++u)
      if (z + 4 * u < nslabs) acc[u] += partial[(size_t)(z + 4 * u) * stride + e];
    float s = ((acc[0] + acc[1]) + (acc[2] + acc[3])) + ((acc[4] + acc[5]) + (acc[6] + acc[7]));
    s += __shfl_xor(s, 1);
    s += __shfl_xor(s, 2);
    s *= scale;
    if (live && q == 0) {
      if (e < count) {
        if (w) s += decay * w[e];
        dw[e] = s;
      } else {
        dbias[e - count] = s;
      }
    }
  }
}

__global__ void slab_reduce2_kernel(const float* __restrict__ partial, float* __restrict__ dw, float* __restrict__ dbias,
                                    const float* __restrict__ w, float decay, float scale, size_t count, int cout,
                                    size_t stride, int nslabs, int wide_ok) {
  slab_reduce_body(partial, dw, dbias, w, decay, scale, count, cout, stride, nslabs, blockIdx.x, gridDim.x, wide_ok);
}

// The slab reductions of MANY backward-filter launches in one launch (a training step's 31: each was a launch of its own
// behind its wgrad_tile, 10 us apiece): workgroup b serves the item whose block range holds b, with exactly the arithmetic
// of slab_reduce2_kernel for that item's own block count -- bitwise the per-layer results.
struct ReduceItem {
  const float* partial;
  float* dw;
  float* dbias;
  const float* w;
  size_t count, stride;
  float decay;
  int cout, nslabs;
  unsigned first_block, nblocks;
  int wide_ok;      // the four-elements-per-thread form may be used ("dbg" 52: never -- the A/B of the two forms' bits)
};

__global__ void slab_reduce_many_kernel(const ReduceItem* __restrict__ items, int n, float scale) {
  int lo = 0, hi = n - 1;           // the last item whose first_block <= blockIdx.x
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (items[mid].first_block <= blockIdx.x) lo = mid; else hi = mid - 1;
  }
  const ReduceItem it = items[lo];
  slab_reduce_body(it.partial, it.dw, it.dbias, it.w, it.decay, scale, it.count, it.cout, it.stride, it.nslabs,
                   blockIdx.x - it.first_block, it.nblocks, it.wide_ok);
}

namespace {

struct WgPlan {
  int variant;   // index into the instantiation table below
  int taps;      // taps per workgroup: 1 (k = 1), 3 or 9 (k = 3)
  int mtile, ntile;
  int ci_tiles, co_tiles, BY, BX, nstages, ksplit;
  size_t count, slab_stride;
};

// variants: {MT, NT, WM} -> tile (WM*MT*16) x ((4/WM)*NT*16)
enum { V_64x16, V_64x32, V_64x48, V_64x80, V_64x64, V_32x64, V_16x64, V_48x64 };

WgPlan wgrad_plan(int n, int h, int w, int cin, int cout, int k) {
  WgPlan p;
  const long P = (long)n * h * w;
  if (cin <= 16) { p.variant = V_16x64; p.mtile = 16; p.ntile = 64; }
  else if (cin <= 48 && cin > 32) { p.variant = V_48x64; p.mtile = 48; p.ntile = 64; }
  else if (k == 1 && cout <= 16) { p.variant = V_64x16; p.mtile = 64; p.ntile = 16; }
  else if (k == 1 && cout <= 32) { p.variant = V_64x32; p.mtile = 64; p.ntile = 32; }
  else if (k == 1 && (cout <= 48 || cout == 96)) { p.variant = V_64x48; p.mtile = 64; p.ntile = 48; }
  else if (k == 3 && cout > 64 && cout <= 80) { p.variant = V_64x80; p.mtile = 64; p.ntile = 80; }
  else if (cin % 64 == 0) { p.variant = V_64x64; p.mtile = 64; p.ntile = 64; }
  else { p.variant = V_32x64; p.mtile = 32; p.ntile = 64; }
  // all nine taps in one workgroup where the accumulators fit (the few-channel early maps, where X / dY traffic
  // matters most); one kernel row per workgroup elsewhere
  p.taps = k == 1 ? 1 : ((P >= 100000 && (p.variant == V_16x64 || p.variant == V_32x64)) ? 9 : 3);
  p.ci_tiles = (cin + p.mtile - 1) / p.mtile;
  p.co_tiles = (cout + p.ntile - 1) / p.ntile;
  p.BY = (h + 3) / 4;
  p.BX = (w + 15) / 16;
  p.nstages = n * p.BY * p.BX;
  p.count = (size_t)k * k * cin * cout;
  p.slab_stride = p.count + (size_t)cout;
  const int tiles = (k * k / p.taps) * p.ci_tiles * p.co_tiles;
  int ks = 512 / tiles;                                               // one round of two resident workgroups per CU
  if (ks > p.nstages / 4) ks = p.nstages / 4;                         // at least four stages each
  const long cap = (16L << 20) / (long)p.slab_stride;                 // bound the slab buffer: ks * |dW| <= 16 M floats
  if (ks > cap) ks = (int)cap;
  if (ks < 1) ks = 1;
  p.ksplit = ks;
  return p;
}

template <typename T, int MT, int NT, int WM, bool NINE>
void launch_taps(const WgPlan& p, dim3 grid, hipStream_t st, const WgArgs& a) {
  if (p.taps == 1) hipLaunchKernelGGL((wgrad_tile<T, MT, NT, WM, 1>), grid, dim3(256), 0, st, a);
  else if (p.taps == 3) hipLaunchKernelGGL((wgrad_tile<T, MT, NT, WM, 3>), grid, dim3(256), 0, st, a);
  else if constexpr (NINE) hipLaunchKernelGGL((wgrad_tile<T, MT, NT, WM, 9>), grid, dim3(256), 0, st, a);
}

template <typename T>
void launch_variant(const WgPlan& p, dim3 grid, hipStream_t st, const WgArgs& a) {
  switch (p.variant) {
    case V_64x16: hipLaunchKernelGGL((wgrad_tile<T, 1, 1, 4, 1>), grid, dim3(256), 0, st, a); break;
    case V_64x32: hipLaunchKernelGGL((wgrad_tile<T, 1, 2, 4, 1>), grid, dim3(256), 0, st, a); break;
    case V_64x48: hipLaunchKernelGGL((wgrad_tile<T, 1, 3, 4, 1>), grid, dim3(256), 0, st, a); break;
    case V_64x80: hipLaunchKernelGGL((wgrad_tile<T, 1, 5, 4, 3>), grid, dim3(256), 0, st, a); break;
    case V_64x64: launch_taps<T, 2, 2, 2, false>(p, grid, st, a); break;
    case V_32x64: launch_taps<T, 1, 2, 2, true>(p, grid, st, a); break;
    case V_16x64: launch_taps<T, 1, 1, 1, true>(p, grid, st, a); break;
    default: launch_taps<T, 3, 1, 1, false>(p, grid, st, a); break;
  }
}

}  // namespace
}  // namespace sqdet

using namespace sqdet;

extern "C" size_t sqdet_conv2d_bwd_filter_workspace_bytes(int n, int h, int w, int cin, int cout, int k) {
  if (n <= 0 || h <= 0 || w <= 0 || cin <= 0 || cout <= 0 || (k != 1 && k != 3)) return 0;
  const WgPlan p = wgrad_plan(n, h, w, cin, cout, k);
  return (size_t)p.ksplit * p.slab_stride * sizeof(float);
}

static int reduce_blocks(size_t total) {
  int blocks = (int)((total + 63) / 64);   // 64 elements (x 4 lanes) per 256-thread workgroup
  return blocks > 4096 ? 4096 : blocks;
}

// dw_hwio == NULL: the partial slabs only (sqdet_conv2d_nhwc_bwd_filter_partial)
static int bwd_filter_impl(const void* x, const void* dy, float* dw_hwio, float* dbias, int want_bias, const float* w_hwio_for_decay,
                           float weight_decay, float grad_scale, float* workspace, int n, int h, int w, int cin, int cout, int k,
                           int x_cstride, int x_coffset, int dy_cstride, int dy_coffset, int dtype, sqdet_stream_t stream) {
  SQDET_REQUIRE(x && dy && workspace, "conv2d_bwd_filter: null pointer");
  SQDET_REQUIRE((k == 1 || k == 3) && n > 0 && h > 0 && w > 0 && cin > 0 && cout > 0, "conv2d_bwd_filter: bad dims");
  SQDET_REQUIRE(dtype == SQDET_F16 || dtype == SQDET_F32, "conv2d_bwd_filter: bad dtype");
  const int ev = dtype == SQDET_F16 ? 8 : 4;
  SQDET_UNSUPPORTED(cin % ev || cout % ev || x_cstride % ev || x_coffset % ev || dy_cstride % ev || dy_coffset % ev,
                    "conv2d_bwd_filter: channel counts / strides must be multiples of 16 bytes");
  const size_t P = (size_t)n * h * w, es = dtype_size(dtype);
  SQDET_UNSUPPORTED(P * x_cstride * es >= 0xfffffff0ull || P * dy_cstride * es >= 0xfffffff0ull,
                    "conv2d_bwd_filter: tensors of 4 GiB or more");
  hipStream_t st = as_stream(stream);
  const WgPlan p = wgrad_plan(n, h, w, cin, cout, k);
  WgArgs a;
  a.x = x; a.dy = dy; a.partial = workspace;
  a.H = h; a.W = w; a.Cin = cin; a.Cout = cout; a.k = k;
  a.x_cstride = x_cstride; a.x_coffset = x_coffset; a.dy_cstride = dy_cstride; a.dy_coffset = dy_coffset;
  a.BY = p.BY; a.BX = p.BX; a.nstages = p.nstages; a.ksplit = p.ksplit; a.ci_tiles = p.ci_tiles;
  a.do_bias = want_bias;
  a.x_bytes = (unsigned)(P * x_cstride * es);
  a.dy_bytes = (unsigned)(P * dy_cstride * es);
  a.slab_stride = p.slab_stride;
  const dim3 grid((k * k / p.taps) * p.ci_tiles, p.co_tiles, p.ksplit);
  if (dtype == SQDET_F16) launch_variant<f16>(p, grid, st, a);
  else launch_variant<float>(p, grid, st, a);
  SQDET_CHECK_HIP(hipGetLastError());
  if (!dw_hwio) return SQDET_OK;
  hipLaunchKernelGGL(slab_reduce2_kernel, dim3(reduce_blocks(p.count + (dbias ? (size_t)cout : 0))), dim3(256), 0, st, workspace,
                     dw_hwio, dbias, w_hwio_for_decay, weight_decay, grad_scale, p.count, cout, p.slab_stride, p.ksplit,
                     tune(TUNE_DBG) == 52 ? 0 : 1);
  SQDET_CHECK_HIP(hipGetLastError());
  return SQDET_OK;
}

extern "C" int sqdet_conv2d_nhwc_bwd_filter(const void* x, const void* dy, float* dw_hwio, float* dbias,
                                            const float* w_hwio_for_decay, float weight_decay, float grad_scale,
                                            float* workspace, int n, int h, int w, int cin, int cout, int k, int x_cstride,
                                            int x_coffset, int dy_cstride, int dy_coffset, int dtype, sqdet_stream_t stream) {
  SQDET_REQUIRE(dw_hwio, "conv2d_bwd_filter: null pointer");
  return bwd_filter_impl(x, dy, dw_hwio, dbias, dbias != nullptr, w_hwio_for_decay, weight_decay, grad_scale, workspace, n, h, w,
                         cin, cout, k, x_cstride, x_coffset, dy_cstride, dy_coffset, dtype, stream);
}

extern "C" int sqdet_conv2d_nhwc_bwd_filter_partial(const void* x, const void* dy, float* workspace, int want_bias, int n, int h,
                                                    int w, int cin, int cout, int k, int x_cstride, int x_coffset,
                                                    int dy_cstride, int dy_coffset, int dtype, sqdet_stream_t stream) {
  return bwd_filter_impl(x, dy, nullptr, nullptr, want_bias != 0, nullptr, 0.f, 1.f, workspace, n, h, w, cin, cout, k, x_cstride,
                         x_coffset, dy_cstride, dy_coffset, dtype, stream);
}

extern "C" size_t sqdet_slab_reduce_many_table_bytes(int n_items) {
  return n_items > 0 ? (size_t)n_items * sizeof(ReduceItem) : 0;
}

extern "C" int sqdet_slab_reduce_many_prepare(const float* const* workspaces, float* const* dws, float* const* dbiases,
                                              const float* const* w_for_decay, const float* decays, const int* n, const int* h,
                                              const int* w, const int* cin, const int* cout, const int* k, int n_items,
                                              void* table_host, int* total_blocks) {
  SQDET_REQUIRE(workspaces && dws && dbiases && w_for_decay && decays && n && h && w && cin && cout && k && table_host &&
                    total_blocks && n_items > 0, "slab_reduce_many_prepare: bad arguments");
  ReduceItem* t = static_cast<ReduceItem*>(table_host);
  unsigned next = 0;
  for (int i = 0; i < n_items; ++i) {
    SQDET_REQUIRE(workspaces[i] && dws[i] && (k[i] == 1 || k[i] == 3) && n[i] > 0 && h[i] > 0 && w[i] > 0 && cin[i] > 0 && cout[i] > 0,
                  "slab_reduce_many_prepare: bad item %d", i);
    const WgPlan p = wgrad_plan(n[i], h[i], w[i], cin[i], cout[i], k[i]);
    ReduceItem& it = t[i];
    it.partial = workspaces[i]; it.dw = dws[i]; it.dbias = dbiases[i]; it.w = w_for_decay[i];
    it.count = p.count; it.stride = p.slab_stride; it.decay = decays[i]; it.cout = cout[i]; it.nslabs = p.ksplit;
    it.first_block = next;
    it.wide_ok = tune(TUNE_DBG) == 52 ? 0 : 1;
    it.nblocks = (unsigned)reduce_blocks(p.count + (dbiases[i] ? (size_t)cout[i] : 0));
    next += it.nblocks;
  }
  *total_blocks = (int)next;
  return SQDET_OK;
}

extern "C" int sqdet_slab_reduce_many(const void* table_dev, int n_items, int total_blocks, float grad_scale, sqdet_stream_t stream) {
  SQDET_REQUIRE(table_dev && n_items > 0 && total_blocks > 0, "slab_reduce_many: bad arguments");
  hipLaunchKernelGGL(slab_reduce_many_kernel, dim3((unsigned)total_blocks), dim3(256), 0, as_stream(stream),
                     static_cast<const ReduceItem*>(table_dev), n_items, grad_scale);
  SQDET_CHECK_HIP(hipGetLastError());
  return SQDET_OK;
}
