// Fused fire module for gfx950: squeeze1x1 -> (expand1x1 || expand3x3) -> concat in ONE launch
// (reference src/nets/squeezeDet.py:81-106).  The squeeze tensor never touches HBM: HBM traffic is
// the fire input + the concat output + the three weight sets (SURVEY.md 8d "fused fire kernel").
//
// One 256-thread workgroup owns an 8 x 16 tile of output pixels of one image.
//   Phase A (squeeze): the 10 x 18 halo of squeeze outputs the 3x3 expand needs is computed on
//     MFMA straight from the NHWC input (B fragments are 16-byte global loads, weights the A
//     operand; K-loop register double-buffered), bias + ReLU, rounded to the storage type and
//     written to LDS in the bank-conflict-free [K-chunk][pixel][4 x 16 B, slot g ^ ((pixel>>1)&3)]
//     layout of conv3x3.hip.  Halo pixels outside the image are ZERO (the expand convs' SAME
//     padding pads the squeeze tensor, not the input).  The halo recompute costs 180/128 of the
//     squeeze FLOPs -- the squeeze is 15-20 % of a module.
//   Phase B (expand): each wave computes all 8 tile rows x NTW cout tiles per work item, items
//     alternate over the expand3x3 tiles (9 taps) and the expand1x1 tiles (centre tap), weights
//     prefetched one step ahead; results go to their channel range of the concat tensor.
// Numerics are identical to the unfused kernels (same chunk order, same MFMA, same roundings).
// Measured per segment (-DSQDET_FIRE_TIMING + tools/ff_timing.py, fire10 at batch 32): phase A 31 % of a wave's
// life (12 % waiting for chunk loads / the per-chunk barrier), the expand K loops 52 % (the matrix pipe ~75 % busy
// inside them), weight refill + epilogue stores 17 %.  Starting the second workgroup of every CU late, so that its
// phase A would sit under the first one's phase B, was tried and is monotonically SLOWER (+4 us per 16 k cycles; the
// co-resident pairs are blocks i and i + 256 and start within ~130 cycles of each other: ff_timing.py --placement).
// Eight instead of four input chunks in flight (decoupled from the weight ring's depth) is 5 % SLOWER as well.
#include "conv_common.h"

namespace sqdet {

constexpr int FROWS = 8, FCOLS = 16;
constexpr int FHP = (FROWS + 2) * (FCOLS + 2);   // 180 halo pixels
constexpr int FCHUNK = FHP * 64;                 // bytes per 64-byte K-chunk of the squeeze tile
constexpr int FBLK = (FHP + 15) / 16;            // 12 pixel blocks in phase A (3 per wave)

struct FireArgs {
  const void* x;
  void* y;
  const void *ws, *w1, *w3;
  const float *bs, *b1, *b3;
  int N, H, W, Cin, S, E1, E3;
  int tiles_x, tiles_y;
  int nch_x;       // 64-byte chunks of the input channels
  int x_pieces;    // Cin*sizeof(T)/16
  int nch_s;       // 64-byte chunks of the squeeze channels
  int e_nt;        // tiles per packed group of the expand convs (same for both)
  int e1_tiles, e3_tiles;   // cout tiles (all groups) of expand1x1 / expand3x3
  unsigned x_bytes;         // size of the input tensor (32-bit buffer offsets)
  void* sq_out;             // not NULL: the squeeze tensor [N,H,W,S] is ALSO written (training: its backward needs it)
};

// MT = tile rows per phase-B work item: 8 (a wave walks the whole tile per cout item) or 4 (two row
// blocks per cout item -- the small fire modules have only 2-4 cout items, which left waves idle or
// paired one 9-tap item with one 1-tap item; with MT = 4 every wave gets the same work and the
// accumulators halve, so 4 workgroups fit on a CU instead of 2).
// -DSQDET_FIRE_TIMING (experiments only): per-wave s_memtime totals of the kernel's segments (tools/fire_timing.py)
#ifdef SQDET_FIRE_TIMING
__device__ unsigned long long g_ff_timing[2048 * 8];
#define FFT_MARK(k) do { const unsigned long long now_ = __builtin_amdgcn_s_memtime(); ft_acc[k] += now_ - ft_last; ft_last = now_; } while (0)
#else
#define FFT_MARK(k) do {} while (0)
#endif

template <typename T, int NTS, int NTW, int MT>
__global__ __launch_bounds__(256, (MT == 4 && NTW <= 4 ? 4 : 2)) void fire_fused(FireArgs a) {
  constexpr int KG = Tr<T>::KG;
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int j = lane & 15, g = lane >> 4;
#ifdef SQDET_FIRE_TIMING
  unsigned long long ft_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, ft_last = __builtin_amdgcn_s_memtime();
  // slot 7: where and when this workgroup started -- HW_ID (cu / sh / se) and XCC_ID above the low 40 bits of the clock
  ft_acc[7] = (ft_last & 0xFFFFFFFFFFull) | ((unsigned long long)(__builtin_amdgcn_s_getreg((31 << 11) | 4) & 0xFFFF) << 40) |
              ((unsigned long long)(__builtin_amdgcn_s_getreg((31 << 11) | 20) & 0xF) << 56);
#endif
  // XCD-aware tile order: workgroup b runs on XCD b % 8 (own L2 each); every XCD gets a contiguous band of
  // tiles so the halos shared by neighbouring tiles are fetched into ONE L2 instead of up to eight.
  int b = (int)((blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3));   // gridDim.x is a multiple of 8
  if (b >= a.N * a.tiles_x * a.tiles_y) return;
  const int tx = b % a.tiles_x; b /= a.tiles_x;
  const int ty = b % a.tiles_y;
  const int n = b / a.tiles_y;
  const int oy0 = ty * FROWS, ox0 = tx * FCOLS;
  float* bl = reinterpret_cast<float*>(lds + a.nch_s * FCHUNK);                 // biases [b1 | b3 | bs]
  unsigned char* wring = lds + a.nch_s * FCHUNK + ((a.E1 + a.E3 + a.S) * 4 + 15) / 16 * 16;   // squeeze-weight ring

  // ---------------------------------------------------------------- phase A: squeeze on the halo
  // The K loop has only MB*NTS <= 18 MFMAs per 64-byte chunk of input channels, far less than a memory
  // latency, and up to 16 chunks: with a one-chunk look-ahead every chunk cost a full latency (17 us of a
  // 62 us fire10).  Now PD chunks are in flight: the input fragments in registers (raw buffer loads: an
  // out-of-range offset returns the zero padding without a branch, so the load count per chunk is fixed),
  // the squeeze weights straight into an LDS ring by global_load_lds (no registers; every wave fetches
  // FW of the NTS fragments and all four waves read all of them).  One barrier per chunk; the ring has
  // PD+1 slots, so the refill issued after the barrier of chunk c lands in the slot chunk c-1 just left.
  {
    constexpr int MB = FBLK / 4;   // 3 pixel blocks per wave
    constexpr int PD = 4;          // chunks in flight
    constexpr int RS = PD + 1;     // ring slots
    constexpr int FW = (NTS + 3) / 4;   // weight fragments fetched per wave per chunk
    constexpr int PER = MB + FW;   // memory instructions per wave per chunk
    int P[MB];
    bool inimg[MB];
    unsigned xoff[MB];
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.x), 0, a.x_bytes, 0x00020000);
    constexpr unsigned OOB = 0xfffffff0u;
#pragma unroll
    for (int mb = 0; mb < MB; ++mb) {
      P[mb] = (wave * MB + mb) * 16 + j;
      const int r = P[mb] / (FCOLS + 2), c = P[mb] - r * (FCOLS + 2);
      const int iy = oy0 - 1 + r, ix = ox0 - 1 + c;
      inimg[mb] = P[mb] < FHP && iy >= 0 && iy < a.H && ix >= 0 && ix < a.W;
      xoff[mb] = inimg[mb] ? (unsigned)((((n * a.H + iy) * a.W + ix) * a.Cin + g * KG) * (int)sizeof(T)) : OOB;
    }
    i32x4 xq[PD][MB];
    const i32x4* wsg = reinterpret_cast<const i32x4*>(a.ws) + lane;
    auto issue = [&](int c, i32x4 (&xr)[MB]) {          // chunk c (uniform c < nch_x)
      const bool k_ok = c * 4 + g < a.x_pieces;
#pragma unroll
      for (int mb = 0; mb < MB; ++mb)
        xr[mb] = __builtin_amdgcn_raw_buffer_load_b128(rx, (k_ok && xoff[mb] != OOB) ? xoff[mb] + c * 64 : OOB, 0, 0);
      unsigned char* slot = wring + (c % RS) * (NTS * 1024);
#pragma unroll
      for (int f = 0; f < FW; ++f) {
        const int t = (wave + 4 * f) % NTS;               // (duplicates when NTS is not a multiple of 4: same bytes)
        __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)(wsg + (c * NTS + t) * 64),
                                         (void __attribute__((address_space(3)))*)(slot + t * 1024), 16, 0, 0);
      }
    };
#pragma unroll
    for (int u = 0; u < PD; ++u)
      if (u < a.nch_x) issue(u, xq[u]);
    // biases -> LDS (read back with ds_read: off the global-memory critical path); visible after the first barrier
    for (int i = threadIdx.x; i < a.E1 + a.E3 + a.S; i += 256)
      bl[i] = i < a.E1 ? a.b1[i] : (i < a.E1 + a.E3 ? a.b3[i - a.E1] : a.bs[i - a.E1 - a.E3]);
    // zero the channel padding of the last squeeze chunk (S not a multiple of 64 bytes)
    const int s_pieces = a.S * (int)sizeof(T) / 16;
    const int pad = a.nch_s * 4 - s_pieces;
    for (int idx = threadIdx.x; idx < FHP * pad; idx += 256) {
      const int PP = idx / pad, q = s_pieces + (idx - PP * pad);
      *reinterpret_cast<i32x4*>(lds + (q >> 2) * FCHUNK + PP * 64 + (((q & 3) ^ ((PP >> 1) & 3)) << 4)) = i32x4{0, 0, 0, 0};
    }
    f32x4 acc[MB][NTS];
#pragma unroll
    for (int mb = 0; mb < MB; ++mb)
#pragma unroll
      for (int t = 0; t < NTS; ++t) acc[mb][t] = f32x4{0.f, 0.f, 0.f, 0.f};
    FFT_MARK(0);
#pragma unroll 1
    for (int c0 = 0; c0 < a.nch_x; c0 += PD) {
#pragma unroll
      for (int u = 0; u < PD; ++u) {
        const int c = c0 + u;
        if (c >= a.nch_x) break;
        // everything up to chunk c has landed when at most the later chunks' instructions are outstanding
        const int later = min(PD - 1, a.nch_x - 1 - c);
        // (0x0070: lgkmcnt(0) too -- this wave's LDS writes / reads are done before it arrives)
        if (later >= 3) __builtin_amdgcn_s_waitcnt(0x0070 | ((3 * PER) & 0xF) | (((3 * PER) >> 4) << 14));
        else if (later == 2) __builtin_amdgcn_s_waitcnt(0x0070 | ((2 * PER) & 0xF) | (((2 * PER) >> 4) << 14));
        else if (later == 1) __builtin_amdgcn_s_waitcnt(0x0070 | ((1 * PER) & 0xF) | (((1 * PER) >> 4) << 14));
        else __builtin_amdgcn_s_waitcnt(0x0070);
        // A bare s_barrier: __syncthreads() carries a workgroup fence, and with LDS-DMA in flight that fence is an
        // `s_waitcnt vmcnt(0)` -- it drained the PD-deep pipeline at every chunk (fire6-11: 338 -> 320 us).  The
        // compiler still puts one vmcnt(0) before the first MFMA of every PD chunks (buffer loads and LDS-DMA are
        // mixed event types to its wait-count pass, which then distrusts the counted waits); hiding the input loads
        // in inline asm removes that one too and measures the same, so the plain builtins stay.
        asm volatile("" ::: "memory");
        __builtin_amdgcn_s_barrier();                     // every wave's share of the weights is in the ring
        asm volatile("" ::: "memory");
        FFT_MARK(1);
        const unsigned char* slot = wring + (c % RS) * (NTS * 1024);
        i32x4 af[NTS];
#pragma unroll
        for (int t = 0; t < NTS; ++t) af[t] = *reinterpret_cast<const i32x4*>(slot + t * 1024 + lane * 16);
#pragma unroll
        for (int mb = 0; mb < MB; ++mb)
#pragma unroll
          for (int t = 0; t < NTS; ++t) mma16<T>(acc[mb][t], af[t], xq[u][mb]);
        if (c + PD < a.nch_x) issue(c + PD, xq[u]);
        FFT_MARK(2);
      }
    }
    // bias + ReLU -> storage type -> LDS squeeze tile; lane = pixel P, channels g*4*NTS + 4t .. +4
#pragma unroll
    for (int t = 0; t < NTS; ++t) {
      const int ch0 = g * 4 * NTS + 4 * t;
      if (ch0 < a.S) {
        const f32x4 bias = *reinterpret_cast<const f32x4*>(bl + a.E1 + a.E3 + ch0);
        const int q = ch0 / KG;                         // 16-byte piece of the pixel's channel vector
        const int sub = (ch0 - q * KG) * (int)sizeof(T);  // byte offset inside the piece (0 or 8 for f16, 0 for f32)
#pragma unroll
        for (int mb = 0; mb < MB; ++mb) {
          if (P[mb] < FHP) {
            f32x4 v = acc[mb][t] + bias;
            v[0] = fmaxf(v[0], 0.f); v[1] = fmaxf(v[1], 0.f); v[2] = fmaxf(v[2], 0.f); v[3] = fmaxf(v[3], 0.f);
            if (!inimg[mb]) v = f32x4{0.f, 0.f, 0.f, 0.f};   // SAME padding of the squeeze tensor
            unsigned char* dst = lds + (q >> 2) * FCHUNK + P[mb] * 64 + (((q & 3) ^ ((P[mb] >> 1) & 3)) << 4) + sub;
            store4<T>(reinterpret_cast<T*>(dst), v);
            if (a.sq_out) {     // the tile's OWN 8 x 16 pixels (not the halo ring) that lie inside the image
              const int hr = P[mb] / (FCOLS + 2), hc = P[mb] - hr * (FCOLS + 2);
              if (inimg[mb] && hr >= 1 && hr <= FROWS && hc >= 1 && hc <= FCOLS)
                store4<T>(reinterpret_cast<T*>(a.sq_out) + ((size_t)(n * a.H + oy0 + hr - 1) * a.W + ox0 + hc - 1) * a.S + ch0, v);
            }
          }
        }
      }
    }
    FFT_MARK(3);
  }
  __syncthreads();
  FFT_MARK(4);

  // ---------------------------------------------------------------- phase B: expand3x3 + expand1x1
  // On the small late maps the whole grid is ONE wave of workgroups: the kernel time is a single workgroup's
  // critical path, so every exposed memory latency on it counts.  Hence: biases come from LDS (filled at
  // kernel start), the weight fragments run through a ring of WD steps kept in flight (they come from L2,
  // ~1-2k cycles away under load, and one step is only MT*NTW <= 24 MFMAs), and the NEXT item's first
  // fragments are requested before the current item's epilogue, so they land while the stores go out.
  constexpr int RB = FROWS / MT;   // row blocks per cout item
  constexpr int WD = (MT == 4 && NTW <= 4) ? 2 : (NTW <= 3 ? 4 : 2);   // even: the B-fragment ping-pong below keys on u & 1
  T* y = reinterpret_cast<T*>(a.y);
  const int ox = ox0 + j;
  const int ctot = a.E1 + a.E3;
  const int n3 = a.e3_tiles / NTW, n1 = a.e1_tiles / NTW;
  const int nw = (n3 + n1) * RB;   // work items: heavy (9-tap) ones first, then the 1-tap ones
  struct Item {
    bool is3;
    int steps, group, n0, m0;
    const i32x4* wbase;
  };
  auto item_of = [&](int witem) {
    Item it;
    const int item = witem / RB;
    it.m0 = (witem - item * RB) * MT;
    it.is3 = item < n3;
    const int tile0 = (it.is3 ? item : item - n3) * NTW;
    it.group = tile0 / a.e_nt;
    it.n0 = tile0 - it.group * a.e_nt;
    it.steps = (it.is3 ? 9 : 1) * a.nch_s;
    it.wbase = reinterpret_cast<const i32x4*>(it.is3 ? a.w3 : a.w1) + ((size_t)it.group * (it.is3 ? 9 : 1) * a.nch_s * a.e_nt + it.n0) * 64 + lane;
    return it;
  };
  // Steps walk chunk-major, tap-minor -- the accumulation order of conv3x3_tile, so the result is bitwise
  // the unfused one; the packed fragment of (tap, chunk) sits at step tap*nch_s + chunk.  expand1x1 = the
  // centre tap only.
  auto frag = [&](const Item& it, int s) {
    const int c = it.is3 ? s / 9 : s;
    const int tap = it.is3 ? s - c * 9 : 0;
    return it.wbase + (size_t)(tap * a.nch_s + c) * a.e_nt * 64;
  };
  i32x4 afr[WD][NTW];
  auto fill = [&](const Item& it) {
#pragma unroll
    for (int u = 0; u < WD; ++u) {
      if (u < it.steps) {
        const i32x4* wp = frag(it, u);
#pragma unroll
        for (int t = 0; t < NTW; ++t) afr[u][t] = wp[t * 64];
      }
    }
  };
  // Work items are dealt round-robin: over the workgroups that share this tile first (gridDim.y > 1 on SMALL batches,
  // where the tiles alone cannot fill the chip -- batch 1 has 15 of them: every such workgroup repeats the cheap
  // squeeze phase and takes its share of the expand items), then over the four waves.
  const int w0 = (int)blockIdx.y + (int)gridDim.y * wave, wstep = 4 * (int)gridDim.y;
  if (w0 < nw) fill(item_of(w0));
  for (int witem = w0; witem < nw; witem += wstep) {
    const Item it = item_of(witem);
    const int m0 = it.m0;
    f32x4 acc[MT][NTW];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int t = 0; t < NTW; ++t) acc[m][t] = f32x4{0.f, 0.f, 0.f, 0.f};
    // B fragments (activations, from the LDS squeeze tile) are double-buffered one step ahead as well: a step
    // whose MFMAs had to wait for its own ds_reads left the matrix pipe idle for an LDS latency every 24 MFMAs.
    auto load_bf = [&](int s, i32x4 (&bf)[MT]) {
      const int c = it.is3 ? s / 9 : s;
      const int tap = it.is3 ? s - c * 9 : 4;
      const int dy = tap / 3, dx = tap - dy * 3;
      const unsigned char* lchunk = lds + c * FCHUNK;
      const int P0 = dy * (FCOLS + 2) + j + dx;
      const int h0 = P0 >> 1;
#pragma unroll
      for (int m = 0; m < MT; ++m) {
        const int slot = g ^ ((h0 + m0 + m) & 3);
        bf[m] = *reinterpret_cast<const i32x4*>(lchunk + (P0 + (FCOLS + 2) * (m0 + m)) * 64 + (slot << 4));
      }
    };
    i32x4 bfp[2][MT];
    load_bf(0, bfp[0]);
#pragma unroll 1
    for (int s0 = 0; s0 < it.steps; s0 += WD) {
#pragma unroll
      for (int u = 0; u < WD; ++u) {
        const int s = s0 + u;
        if (s >= it.steps) break;
        if (s + 1 < it.steps) load_bf(s + 1, bfp[(u + 1) & 1]);
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
          for (int t = 0; t < NTW; ++t) mma16<T>(acc[m][t], afr[u][t], bfp[u & 1][m]);
        if (s + WD < it.steps) {
          const i32x4* wp = frag(it, s + WD);
#pragma unroll
          for (int t = 0; t < NTW; ++t) afr[u][t] = wp[t * 64];
        }
      }
    }
    FFT_MARK(5);
    if (witem + wstep < nw) fill(item_of(witem + wstep));   // the ring is empty here: next item's first fragments
    // epilogue: bias + ReLU, 4*NTW consecutive channels per lane into the concat tensor
    const int cout = it.is3 ? a.E3 : a.E1;
    const int coff = it.is3 ? a.E1 : 0;
    const int cb = it.group * 16 * a.e_nt + g * 4 * a.e_nt + it.n0 * 4;
    f32x4 bias[NTW];
    int nt_valid = 0;
#pragma unroll
    for (int t = 0; t < NTW; ++t) {
      const bool ok = cb + t * 4 < cout;
      bias[t] = ok ? *reinterpret_cast<const f32x4*>(bl + coff + cb + t * 4) : f32x4{0.f, 0.f, 0.f, 0.f};
      nt_valid += ok ? 1 : 0;
    }
    if (ox < a.W) {
#pragma unroll
      for (int m = 0; m < MT; ++m) {
        const int oy = oy0 + m0 + m;
        if (oy >= a.H) break;
        T* dst = y + (((size_t)n * a.H + oy) * a.W + ox) * ctot + coff + cb;
        f32x4 v[NTW];
#pragma unroll
        for (int t = 0; t < NTW; ++t) {
          v[t] = acc[m][t] + bias[t];
          v[t][0] = fmaxf(v[t][0], 0.f); v[t][1] = fmaxf(v[t][1], 0.f);
          v[t][2] = fmaxf(v[t][2], 0.f); v[t][3] = fmaxf(v[t][3], 0.f);
        }
        store_couts<T, NTW>(dst, v, nt_valid);
      }
    }
    FFT_MARK(6);
  }
#ifdef SQDET_FIRE_TIMING
  if (lane == 0 && blockIdx.x * 4 + wave < 2048)
    for (int k = 0; k < 8; ++k) g_ff_timing[(blockIdx.x * 4 + wave) * 8 + k] = ft_acc[k];
#endif
}

template <typename T, int NTS, int NTW, int MT>
static void launch_ff(const FireArgs& a, size_t lds, hipStream_t st) {
  static PerDevice once;   // > 64 KiB of dynamic LDS has to be allowed once per kernel and device
  if (lds > 65536)
    (void)once.run([] { return hipFuncSetAttribute(reinterpret_cast<const void*>(&fire_fused<T, NTS, NTW, MT>),
                                                   hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); });
  // few tiles (small batches): split the expand items of a tile over up to items/4 workgroups (one item per wave)
  const int tiles = a.N * a.tiles_x * a.tiles_y;
  const int items = (a.e3_tiles / NTW + a.e1_tiles / NTW) * (FROWS / MT);
  int split = 1;
  if (tiles < 256 && tune(TUNE_DBG) != 25) {
    split = 512 / (tiles > 0 ? tiles : 1);
    if (split > items / 4) split = items / 4;
    if (split < 1) split = 1;
  }
  const dim3 grid((unsigned)((tiles + 7) / 8 * 8), (unsigned)split);
  hipLaunchKernelGGL((fire_fused<T, NTS, NTW, MT>), grid, dim3(256), lds, st, a);
}

template <typename T, int NTS>
static bool dispatch_fire_ntw(const FireArgs& a, int ntw, size_t lds, hipStream_t st) {
  // few cout items (fire2..5: 2 or 4): split the tile rows too, so all four waves carry equal work
  const bool split_rows = (a.e3_tiles + a.e1_tiles) / ntw <= 4;
  switch (ntw) {
    case 2:
      if (split_rows) launch_ff<T, NTS, 2, 4>(a, lds, st);
      else launch_ff<T, NTS, 2, 8>(a, lds, st);
      return true;
    case 3:
      // (96-cout groups.  A whole group -- 6 tiles -- per item for half of the rows gives a lane 24 consecutive
      // channels = 16-byte stores instead of 8-byte ones, but measured equal: not instantiated.)
      if (split_rows) launch_ff<T, NTS, 3, 4>(a, lds, st);
      else launch_ff<T, NTS, 3, 8>(a, lds, st);
      return true;
    case 4:
      if (split_rows) launch_ff<T, NTS, 4, 4>(a, lds, st);
      else launch_ff<T, NTS, 4, 8>(a, lds, st);
      return true;
    default: return false;
  }
}

template <typename T>
static bool dispatch_fire(const FireArgs& a, int nts, int ntw, size_t lds, hipStream_t st) {
  switch (nts) {
    case 1: return dispatch_fire_ntw<T, 1>(a, ntw, lds, st);
    case 2: return dispatch_fire_ntw<T, 2>(a, ntw, lds, st);
    case 3: return dispatch_fire_ntw<T, 3>(a, ntw, lds, st);
    case 4: return dispatch_fire_ntw<T, 4>(a, ntw, lds, st);
    case 6: return dispatch_fire_ntw<T, 6>(a, ntw, lds, st);
    default: return false;
  }
}

// Same checks as fire_fused_launch, for the executor's plan-time decision.
bool fire_fused_eligible(int cin, int s, int e1, int e3, int dtype) {
  if (conv_algo() != 0) return false;
  if (tune(TUNE_FIRE_FUSE) != 3 && fire_stream_eligible(cin, s, e1, e3, dtype)) return true;
  const int esz = dtype == SQDET_F16 ? 2 : 4;
  const ConvGeom gs = conv_geom(1, cin, s, dtype), g1 = conv_geom(1, s, e1, dtype), g3 = conv_geom(3, s, e3, dtype);
  if (gs.gather || g1.gather || g3.gather || gs.ngroups != 1) return false;
  if (!(gs.nt == 1 || gs.nt == 2 || gs.nt == 3 || gs.nt == 4 || gs.nt == 6)) return false;
  if (g1.nt != g3.nt || g1.nchunk != g3.nchunk) return false;
  if ((cin * esz) % 16 != 0 || (s * esz) % 16 != 0 || s % 4 != 0 || e1 % 8 != 0 || e3 % 8 != 0) return false;
  const int ntw = g1.nt == 6 ? 3 : (g1.nt == 4 ? 4 : (g1.nt == 2 ? 2 : 0));
  if (!ntw) return false;
  if ((g1.nt * g1.ngroups) % ntw || (g3.nt * g3.ngroups) % ntw) return false;
  return (size_t)g1.nchunk * FCHUNK + (size_t)(e1 + e3 + s) * 4 + 16 + 5 * (size_t)gs.nt * 1024 <= 80000;
}

// *handled = false: not eligible, run the three convs separately.
int fire_fused_launch(const void* x, const void* ws, const float* bs, const void* w1, const float* b1, const void* w3,
                      const float* b3, void* y, int n, int h, int w, int cin, int s, int e1, int e3, int dtype,
                      hipStream_t st, bool* handled) {
  return fire_fused_launch_keep(x, ws, bs, w1, b1, w3, b3, nullptr, y, n, h, w, cin, s, e1, e3, dtype, st, handled);
}

// sq_out != NULL: the squeeze tensor is written as well (sqdet_fire_fwd_keep)
int fire_fused_launch_keep(const void* x, const void* ws, const float* bs, const void* w1, const float* b1, const void* w3,
                           const float* b3, void* sq_out, void* y, int n, int h, int w, int cin, int s, int e1, int e3, int dtype,
                           hipStream_t st, bool* handled) {
  *handled = false;
  if (conv_algo() != 0) return SQDET_OK;
  if (tune(TUNE_FIRE_FUSE) != 3) {   // large maps with few channels: the persistent streaming kernel (fire2.hip)
    const int rc = fire_stream_launch_keep(x, ws, bs, w1, b1, w3, b3, sq_out, y, n, h, w, cin, s, e1, e3, dtype, st, handled);
    if (rc != SQDET_OK || *handled) return rc;
  }
  const int esz = dtype == SQDET_F16 ? 2 : 4;
  const ConvGeom gs = conv_geom(1, cin, s, dtype), g1 = conv_geom(1, s, e1, dtype), g3 = conv_geom(3, s, e3, dtype);
  if (gs.gather || g1.gather || g3.gather || gs.ngroups != 1) return SQDET_OK;
  if (g1.nt != g3.nt || g1.nchunk != g3.nchunk) return SQDET_OK;
  if ((cin * esz) % 16 != 0 || (s * esz) % 16 != 0 || s % 4 != 0 || e1 % 8 != 0 || e3 % 8 != 0) return SQDET_OK;
  int ntw = g1.nt == 6 ? 3 : (g1.nt == 4 ? 4 : (g1.nt == 2 ? 2 : 0));
  if (!ntw) return SQDET_OK;
  const int e1_tiles = g1.nt * g1.ngroups, e3_tiles = g3.nt * g3.ngroups;
  if (e1_tiles % ntw || e3_tiles % ntw) return SQDET_OK;
  // squeeze tile + biases + the 5-slot squeeze-weight ring
  const size_t lds = (size_t)g1.nchunk * FCHUNK + (size_t)(e1 + e3 + s) * 4 + 16 + 5 * (size_t)gs.nt * 1024;
  if (lds > 80000) return SQDET_OK;
  if ((long)n * h * w * cin * esz >= (1L << 31)) return SQDET_OK;   // 32-bit buffer offsets
  FireArgs a;
  a.x = x; a.y = y; a.ws = ws; a.w1 = w1; a.w3 = w3; a.bs = bs; a.b1 = b1; a.b3 = b3;
  a.N = n; a.H = h; a.W = w; a.Cin = cin; a.S = s; a.E1 = e1; a.E3 = e3;
  a.tiles_x = (w + FCOLS - 1) / FCOLS; a.tiles_y = (h + FROWS - 1) / FROWS;
  a.nch_x = gs.nchunk; a.x_pieces = cin * esz / 16; a.nch_s = g1.nchunk;
  a.e_nt = g1.nt; a.e1_tiles = e1_tiles; a.e3_tiles = e3_tiles;
  a.x_bytes = (unsigned)((long)n * h * w * cin * esz);
  a.sq_out = sq_out;
  if ((long)n * a.tiles_x * a.tiles_y > 0x7fffffffL) return SQDET_OK;
  const bool ok = dtype == SQDET_F16 ? dispatch_fire<f16>(a, gs.nt, ntw, lds, st) : dispatch_fire<float>(a, gs.nt, ntw, lds, st);
  if (!ok) return SQDET_OK;
  SQDET_CHECK_HIP(hipGetLastError());
  *handled = true;
  return SQDET_OK;
}

}  // namespace sqdet

#ifdef SQDET_FIRE_TIMING
extern "C" int sqdet_debug_ff_timing(unsigned long long* host, int count) {
  return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(sqdet::g_ff_timing), sizeof(unsigned long long) * count);
}
#endif
