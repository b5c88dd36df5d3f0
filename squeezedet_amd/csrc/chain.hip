// Fire-module CHAIN kernel for gfx950 (float16): the expand half of one fire module and the squeeze of the NEXT
// one in a single launch (reference src/nets/squeezeDet.py:58-69,81-106: fire6 ... fire11 run back to back on the
// 24 x 78 map, and the only reader of fire_i's concat tensor is fire_{i+1}'s squeeze1x1).
//
//   sq_in  [n,h,w,S]   the module's squeeze tensor (relu(conv1x1(x)), produced by the previous launch)
//   e      = concat(relu(conv1x1(sq_in, W1) + b1), relu(conv3x3(sq_in, W3) + b3))        -- never stored unless y != NULL
//   sq_out = relu(conv1x1(e, Ws2) + bs2)  [n,h,w,S2]                                      -- the next module's squeeze tensor
//
// What crosses HBM is two tensors of 48-96 channels instead of two of 384-768: the late-map launches stop being a
// "load everything, then compute, then store everything" single round and become MFMA-bound.
//
// Structure (one 256-thread workgroup per CU, one wave per SIMD, up to 512 registers per lane):
//   * a workgroup owns the SAME 8 x 16 tile of TWO images (256 pixels); wave w owns image w>>1, tile rows
//     4*(w&1) .. +4.  Waves split PIXELS, so every wave walks all couts -- which is what lets the next squeeze
//     accumulate in registers in the canonical chunk order (bitwise the unfused result).
//   * every weight the workgroup needs -- W1, W3 and Ws2, 0.3-0.9 MB -- arrives as ONE linear stream through a
//     6-stage x 12 KiB LDS ring filled by global_load_lds (each wave fetches a quarter of every 4-KiB slot, all
//     four waves read all of it): the workgroup's 256 pixels share one weight stream.  Stage k is made
//     available by a counted `s_waitcnt vmcnt(9)` + one bare s_barrier, placed one slot BEFORE stage k-1
//     ends; the refill issued behind that barrier goes into the buffer stage k-2 has left.
//   * cout blocks of 64 (four 16-wide MFMA tiles, 16 accumulators per wave): expand1x1 blocks first, then the
//     expand3x3 blocks -- ascending concat channels.  The pack-time row permutation gives lane group g of a tile
//     PAIR the 8 consecutive couts 32p + 8g .. +8, so after bias + ReLU + float16 rounding the accumulators of a
//     pair ARE the B fragment of one 64-byte K chunk of the next squeeze (D layout == B layout, nothing moves):
//     48 more MFMAs per block chain it into the squeeze accumulators.
//   * the 3x3 taps of a K chunk read 18 LDS fragments (6 halo rows x 3 column shifts) ONCE for nine taps.
// Accumulation orders are those of conv3x3_tile / conv1x1_stream / fire_fused (chunk-major, tap-minor; squeeze
// chunks ascending), so the result is bitwise the three-launch path's.
#include "conv_common.h"
#include "chain.h"
#include "filter_body.h"

namespace sqdet {

namespace {

constexpr int CROWS = 8, CCOLS = 16;
constexpr int CHP = (CROWS + 2) * (CCOLS + 2);   // 180 halo pixels of one image's tile
constexpr int CCHUNK = CHP * 64;                 // bytes of one 64-byte K chunk of a tile
constexpr int STAGE_B = 12288;                   // 3 slots of 4 KiB
constexpr int NIMG = 2;

struct ChainGeom {
  int nch, nsq, nb1, nb3, chain, base3, per3, nstages;
};

__host__ __device__ inline ChainGeom chain_geom(int s, int e1, int e3, int s2) {
  ChainGeom g;
  g.nch = (s * 2 + 63) / 64;
  g.nsq = s2 / 16;
  g.nb1 = e1 / 64;
  g.nb3 = e3 / 64;
  g.chain = s2 > 0 ? 1 : 0;
  g.base3 = g.nb1 * (1 + g.chain);
  g.per3 = 3 * g.nch + g.chain;
  g.nstages = g.base3 + g.nb3 * g.per3;
  return g;
}

struct ChainArgs {
  const void* sq_in;
  void* sq_out;
  void* y;
  const unsigned char* stream;
  const float *b1, *b3, *bs2;
  int N, H, W, S, E1, E3, S2;
  int tiles_x, tiles_y;
  int nb1, nb3, nstages;
  unsigned in_bytes, out_bytes, y_bytes;
  // RIDERS (chain.h): workgroups behind the `per_xcd` chain slots of every XCD that run filter_prediction's top-N branch
  // for images of the PREVIOUS batch instead of a tile
  int per_xcd;             // chain workgroups per XCD (grid = 8 * (per_xcd + riders per XCD))
  ChainRide ride;
};

// One 1-KiB piece of the weight stream straight into LDS (no registers): global address = wave-uniform `sbase` +
// per-lane `voff`, LDS address = M0 (wave-uniform) + lane * 16.  Hidden from hipcc's wait-count pass on purpose:
// its completion is counted by hand (vm_wait below).  (hipcc has no other use for M0 in this kernel: gfx9 LDS
// instructions do not read it.)
__device__ __forceinline__ void glds16(unsigned voff, const unsigned char* sbase, unsigned lds_dst) {
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(sbase), "s"(lds_dst) : "memory", "m0");
}

// -DSQDET_CHAIN_TIMELINE (experiments only, tools/chain_timeline.py): eight s_memrealtime (100 MHz) stamps per workgroup of the
// ring kernel -- entry, loads issued, squeeze tile in LDS, first weight stage landed, end of the expand1x1 blocks, end of the
// expand3x3 blocks, ring drained, last store issued -- kept in scalar registers and written once at the very end (a store in
// the loop would join the hand-counted vmcnt queue).
#ifdef SQDET_CHAIN_TIMELINE
__device__ unsigned long long g_chain_tl[4096 * 8];
#define CTL(k) do { ctl[k] = wall_clock64(); } while (0)
#else
#define CTL(k) do {} while (0)
#endif

template <int N>
__device__ __forceinline__ void vm_wait() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

__device__ __forceinline__ i32x4 pack8(const f32x4& a, const f32x4& b) {
  f16x8 h = {(f16)a[0], (f16)a[1], (f16)a[2], (f16)a[3], (f16)b[0], (f16)b[1], (f16)b[2], (f16)b[3]};
  return __builtin_bit_cast(i32x4, h);
}

__device__ __forceinline__ f32x4 relu4(f32x4 v) {
  v[0] = fmaxf(v[0], 0.f); v[1] = fmaxf(v[1], 0.f); v[2] = fmaxf(v[2], 0.f); v[3] = fmaxf(v[3], 0.f);
  return v;
}

// NCH: 64-byte chunks of the squeeze channels (2 or 3); NSQ: 16-wide tiles of the next squeeze (0 = none, 3, 4, 6);
// WY: the concat tensor is written; RING: ring stages (6, or 4 where the exchange area leaves no room for more).
//
// Eight waves, two per SIMD (<= 256 registers): wave = (pixel group pg = wave & 3, cout half h = wave >> 2).  The
// two waves of a pixel group sit on the same SIMD and own the two tile PAIRS of every 64-cout block, i.e. the two
// K chunks that block contributes to the next squeeze; they swap their rounded float16 results through a 4-KiB
// LDS slot each (behind the barrier the chain stage has anyway) and each accumulates HALF of the squeeze's cout
// tiles over both chunks, in the canonical order.  While one wave of a SIMD waits -- for a barrier, an LDS
// fragment, the ~60 cycles an LDS-DMA issue takes -- the other one keeps the matrix pipe busy.
// DBG (experiments only, wrong results): 1 no vmcnt waits, 2 + no weight stream, 3 + no barriers, 4 + B fragments loaded once,
// 5 + A fragments loaded once, 6 + no block epilogue / chain.
template <int NCH, int NSQ, bool WY, int RING, int DBG = 0>
__global__ __launch_bounds__(512, 2) void fire_chain(ChainArgs a) {
  using T = f16;
  constexpr int LOOK = RING - 2;            // stages issued ahead of the one being consumed
  constexpr int TH = (NSQ + 1) / 2;         // squeeze tiles per wave of a pair
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int pg = wave & 3, h = wave >> 2;
  const int j = lane & 15, g = lane >> 4;
#ifdef SQDET_CHAIN_TIMELINE
  unsigned long long ctl[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#endif
  CTL(0);

  if ((int)(blockIdx.x >> 3) >= a.per_xcd) {
    // ---- RIDER: this workgroup sits on a CU the launch would leave idle (240 chain workgroups on 256 CUs at batch 32) and
    // does the decode + top-N + NMS of image(s) of the PREVIOUS batch (filter_body.h), straight into the caller's output
    // rows (pinned host memory in the serving loop).  No side stream, no events, no extra launch: see chain.h.
    const int rider = (int)((blockIdx.x >> 3) - a.per_xcd) * 8 + (int)(blockIdx.x & 7);
    FastLds<512>& fs = *reinterpret_cast<FastLds<512>*>(lds);
    for (int im = rider; im < a.ride.nimg; im += a.ride.nriders) {
      filter_one_image<true, f16, 512>(a.ride.fa, a.ride.da, a.ride.img0 + im, fs);
      __syncthreads();
    }
    return;
  }
  int b = (int)((blockIdx.x & 7) * a.per_xcd + (blockIdx.x >> 3));   // XCD-banded order
  const int npairs = (a.N + 1) >> 1;
  if (b >= npairs * a.tiles_x * a.tiles_y) return;
  const int tx = b % a.tiles_x; b /= a.tiles_x;
  const int ty = b % a.tiles_y;
  const int np = b / a.tiles_y;
  const int oy0 = ty * CROWS, ox0 = tx * CCOLS;

  unsigned char* ring = lds;
  unsigned char* stile = lds + RING * STAGE_B;                         // [img][chunk][pixel][4 x 16 B swizzled]
  unsigned char* xch = stile + NIMG * NCH * CCHUNK;                    // [wave][m][lane][16 B] (NSQ > 0)
  float* bl = reinterpret_cast<float*>(xch + (NSQ > 0 ? 32768 : 0));   // biases [b1 | b3 | bs2]
  const unsigned ring_addr = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)ring;

  // ---------------------------------------------------------------- squeeze tile (both images, with halo): loads
  constexpr int NP = NCH * 4;                         // 16-byte pieces per pixel in LDS (zero padded)
  constexpr int SIT = (NIMG * CHP * NP + 511) / 512;  // pieces per thread
  i32x4 sv[SIT];
  {
    const int s_pieces = a.S * 2 / 16;
    const __amdgpu_buffer_rsrc_t rin = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.sq_in), 0, a.in_bytes, 0x00020000);
    constexpr unsigned OOB = 0xfffffff0u;
#pragma unroll
    for (int it = 0; it < SIT; ++it) {
      const int idx = it * 512 + (int)threadIdx.x;
      const int im = idx / (CHP * NP);
      const int rem = idx - im * (CHP * NP);
      const int P = rem / NP, q = rem - P * NP;
      const int r = P / (CCOLS + 2), c = P - r * (CCOLS + 2);
      const int iy = oy0 - 1 + r, ix = ox0 - 1 + c;
      const int n = np * 2 + im;
      const bool ok = idx < NIMG * CHP * NP && q < s_pieces && n < a.N && iy >= 0 && iy < a.H && ix >= 0 && ix < a.W;
      const unsigned off = ok ? (unsigned)((((n * a.H + iy) * a.W + ix) * a.S) * 2 + q * 16) : OOB;
      sv[it] = __builtin_amdgcn_raw_buffer_load_b128(rin, off, 0, 0);   // out of range = the zero padding
    }
  }

  // ---------------------------------------------------------------- the weight stream
  // A stage is 12 one-KiB pieces (3 slots x 4 fragments): waves 0-3 fetch pieces w and 8 + w, waves 4-7 piece w.
  // EVERY sync is followed by one refill (the stream buffer carries LOOK + 1 dummy stages behind the last real
  // one), so exactly LOOK - 1 stages are in flight behind the one being waited for: the wait count is a constant
  // per wave and the steady state needs no bookkeeping beyond two wrapping ring offsets.
  const unsigned voff1 = (unsigned)(wave * 1024 + lane * 16), voff2 = voff1 + 8192;
  const unsigned char* sp = a.stream;                                    // next stage to request
  const unsigned m0_lo = ring_addr + (unsigned)wave * 1024;
  unsigned m0n = m0_lo;                                                  // LDS address its pieces go to
  auto refill = [&]() {
    if constexpr (DBG >= 2) return;
    glds16(voff1, sp, m0n);
    if (h == 0) glds16(voff2, sp, m0n + 8192);
    sp += STAGE_B;
    m0n = m0n + STAGE_B == m0_lo + RING * STAGE_B ? m0_lo : m0n + STAGE_B;
  };
#pragma unroll
  for (int s = 0; s < LOOK; ++s) refill();
  CTL(1);

  // ---------------------------------------------------------------- squeeze tile + biases -> LDS
  {
#pragma unroll
    for (int it = 0; it < SIT; ++it) {
      const int idx = it * 512 + (int)threadIdx.x;
      const int im = idx / (CHP * NP);
      const int rem = idx - im * (CHP * NP);
      const int P = rem / NP, q = rem - P * NP;
      if (idx < NIMG * CHP * NP)
        *reinterpret_cast<i32x4*>(stile + (im * NCH + (q >> 2)) * CCHUNK + P * 64 + (((q & 3) ^ ((P >> 1) & 3)) << 4)) = sv[it];
    }
    const int nbias = a.E1 + a.E3 + a.S2;
    for (int i = threadIdx.x; i < nbias; i += 512)
      bl[i] = i < a.E1 ? a.b1[i] : (i < a.E1 + a.E3 ? a.b3[i - a.E1] : a.bs2[i - a.E1 - a.E3]);
  }
  __syncthreads();
  CTL(2);

  const int img = pg >> 1, r0 = (pg & 1) * 4;
  const unsigned char* simg = stile + img * NCH * CCHUNK;
  // B fragment of (chunk c, halo row rr of this wave = tile row r0 - 1 + rr, column shift dx)
  auto load_b = [&](int c, int rr, int dx) {
    const int P = (r0 + rr) * (CCOLS + 2) + j + dx;
    return *reinterpret_cast<const i32x4*>(simg + c * CCHUNK + P * 64 + ((g ^ ((P >> 1) & 3)) << 4));
  };
  unsigned cbo = 0, nbo = STAGE_B;   // ring offsets of the stage being consumed and of the next one
  auto lda = [&](unsigned bo, int f) {
    return *reinterpret_cast<const i32x4*>(ring + bo + f * 1024 + lane * 16);
  };
  int ks = 0, ep_ks = -100;   // (WY only) syncs so far, and their number at the time of the last concat-tensor stores
  // Makes the NEXT stage available: every wave's pieces of it have landed (own vmcnt, then the barrier), and
  // every wave is past the stage before the current one -- its ring buffer takes the refill that follows.
  // LOOK - 1 younger stages are in flight behind it; the 4 concat-tensor stores of a block epilogue are younger
  // than the awaited pieces for the next LOOK syncs (loads and stores retire in issue order).
  auto sync = [&](bool lds_writes) {
    bool st = false;
    if constexpr (WY) { st = (ks - ep_ks) < LOOK; ++ks; }
    if constexpr (DBG < 1) {
      if (h == 0) { if (st) vm_wait<2 * (LOOK - 1) + 4>(); else vm_wait<2 * (LOOK - 1)>(); }
      else { if (st) vm_wait<(LOOK - 1) + 4>(); else vm_wait<(LOOK - 1)>(); }
    }
    if (lds_writes) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if constexpr (DBG < 3) __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
  };
  auto next_buf = [&]() {
    cbo = nbo;
    nbo = nbo + STAGE_B == RING * STAGE_B ? 0 : nbo + STAGE_B;
  };

  f32x4 accs[TH > 0 ? TH : 1][4];
#pragma unroll
  for (int t = 0; t < (TH > 0 ? TH : 1); ++t)
#pragma unroll
    for (int m = 0; m < 4; ++m) accs[t][m] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int ctot = a.E1 + a.E3;
  const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc(a.y, 0, WY ? a.y_bytes : 0u, 0x00020000);
  const int n_img = np * 2 + img;
  const int ox = ox0 + j;
  const bool col_ok = ox < a.W && n_img < a.N;

  i32x4 an[2];      // this wave's two fragments (tiles 2h, 2h+1) of slot 0 of the stage about to be consumed
  sync(false);      // stage 0
  CTL(3);
  refill();
#pragma unroll
  for (int n = 0; n < 2; ++n) an[n] = lda(0, 2 * h + n);

  // Block epilogue: bias + ReLU + float16 rounding of this wave's 32 couts (concat channels cc0 + 32h .. +32) of its
  // 64 pixels; optional store; chain into the next squeeze (consumes one ring stage: Ws2 rows cc0 .. cc0+64).
  auto finish_block = [&](f32x4 (&acc)[4][2], int cc0) {
    i32x4 bf[4];
    {
      const f32x4 bias0 = *reinterpret_cast<const f32x4*>(bl + cc0 + h * 32 + g * 8);
      const f32x4 bias1 = *reinterpret_cast<const f32x4*>(bl + cc0 + h * 32 + g * 8 + 4);
#pragma unroll
      for (int m = 0; m < 4; ++m) bf[m] = pack8(relu4(acc[m][0] + bias0), relu4(acc[m][1] + bias1));
    }
    if (WY) {
#pragma unroll
      for (int m = 0; m < 4; ++m) {
        const int oy = oy0 + r0 + m;
        const bool ok = col_ok && oy < a.H;
        const unsigned off = ok ? (unsigned)((((n_img * a.H + oy) * a.W + ox) * ctot + cc0 + h * 32 + g * 8) * 2) : 0xfffffff0u;
        __builtin_amdgcn_raw_buffer_store_b128(bf[m], ry, off, 0, 0);   // out of range = dropped
      }
      ep_ks = ks;
    }
    if constexpr (NSQ > 0) {
      // chain stage: ring fragments f = u * NSQ + t (u = chunk = the pair member that produced it, t = squeeze tile);
      // this wave accumulates tiles h*TH .. (+TH) over u = 0, 1 in that order
      i32x4 fr[2][TH];
#pragma unroll
      for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int i = 0; i < TH; ++i) {
          const int t = min(h * TH + i, NSQ - 1);
          fr[u][i] = lda(cbo, u * NSQ + t);
        }
#pragma unroll
      for (int m = 0; m < 4; ++m)
        *reinterpret_cast<i32x4*>(xch + (wave * 4 + m) * 1024 + lane * 16) = bf[m];
      sync(true);
      // both K chunks of the block come back from the exchange area in canonical order (u = 0: the h = 0 wave's, u = 1: the
      // h = 1 wave's) -- this wave's own chunk included: four more 16-byte LDS reads instead of 64 lane-uniform selects
      // between the registers it still holds and its partner's (h is a run-time scalar)
      i32x4 bx[2][4];
#pragma unroll
      for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int m = 0; m < 4; ++m) bx[u][m] = *reinterpret_cast<const i32x4*>(xch + ((pg + 4 * u) * 4 + m) * 1024 + lane * 16);
#pragma unroll
      for (int n = 0; n < 2; ++n) an[n] = lda(nbo, 2 * h + n);
#pragma unroll
      for (int u = 0; u < 2; ++u) {
#pragma unroll
        for (int i = 0; i < TH; ++i) {
          if (NSQ % 2 == 0 || h * TH + i < NSQ) {        // (an even NSQ splits evenly over the pair: no test)
#pragma unroll
            for (int m = 0; m < 4; ++m) mma16<T>(accs[i][m], fr[u][i], bx[u][m]);
          }
          if (u == 0 && i == 0) refill();
        }
      }
      next_buf();
    }
  };

  // ---------------------------------------------------------------- expand1x1 blocks (centre tap only)
  {
    i32x4 b1f[NCH][4];
#pragma unroll
    for (int c = 0; c < NCH; ++c)
#pragma unroll
      for (int m = 0; m < 4; ++m) b1f[c][m] = load_b(c, m + 1, 1);
#pragma unroll 1
    for (int blk = 0; blk < a.nb1; ++blk) {
      f32x4 acc[4][2];
#pragma unroll
      for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int n = 0; n < 2; ++n) acc[m][n] = f32x4{0.f, 0.f, 0.f, 0.f};
      i32x4 ac[2];
#pragma unroll
      for (int n = 0; n < 2; ++n) ac[n] = an[n];
#pragma unroll
      for (int c = 0; c < NCH; ++c) {
        i32x4 nx[2];
        if (c + 1 < NCH) {
#pragma unroll
          for (int n = 0; n < 2; ++n) nx[n] = lda(cbo, (c + 1) * 4 + 2 * h + n);
        } else {
          sync(false);
#pragma unroll
          for (int n = 0; n < 2; ++n) nx[n] = lda(nbo, 2 * h + n);
        }
#pragma unroll
        for (int m = 0; m < 4; ++m) {
#pragma unroll
          for (int n = 0; n < 2; ++n) mma16<T>(acc[m][n], ac[n], b1f[c][m]);
          if (c + 1 == NCH && m == 2 * h) refill();
        }
#pragma unroll
        for (int n = 0; n < 2; ++n) ac[n] = nx[n];
      }
#pragma unroll
      for (int n = 0; n < 2; ++n) an[n] = ac[n];
      next_buf();
      finish_block(acc, blk * 64);
    }
  }

  CTL(4);
  // ---------------------------------------------------------------- expand3x3 blocks
  // The 18 B fragments of a K chunk (6 halo rows x 3 column shifts) serve its nine taps.  They are fetched while
  // the MFMAs run: rows 0,1 of the NEXT chunk during this chunk's last tap row (which reads rows 2..5), rows 2..4
  // behind the first MFMAs of the chunk (tap row 0 starts on rows 0,1), row 5 during tap row 0.
  i32x4 B[6][3];
#pragma unroll
  for (int rr = 0; rr < 2; ++rr)
#pragma unroll
    for (int dx = 0; dx < 3; ++dx) B[rr][dx] = load_b(0, rr, dx);
#pragma unroll 1
  for (int blk = 0; blk < a.nb3; ++blk) {
    f32x4 acc[4][2];
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
      for (int n = 0; n < 2; ++n) acc[m][n] = f32x4{0.f, 0.f, 0.f, 0.f};
    i32x4 ac[2];
#pragma unroll
    for (int n = 0; n < 2; ++n) ac[n] = an[n];
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
#pragma unroll
      for (int dy = 0; dy < 3; ++dy) {
        if (DBG >= 4 && blk > 0) {
        } else if (dy == 0) {
#pragma unroll
          for (int rr = 2; rr < 5; ++rr)
#pragma unroll
            for (int dx = 0; dx < 3; ++dx) B[rr][dx] = load_b(c, rr, dx);
        } else if (dy == 1) {
#pragma unroll
          for (int dx = 0; dx < 3; ++dx) B[5][dx] = load_b(c, 5, dx);
        } else {
#pragma unroll
          for (int rr = 0; rr < 2; ++rr)
#pragma unroll
            for (int dx = 0; dx < 3; ++dx) B[rr][dx] = load_b((c + 1) % NCH, rr, dx);
        }
#pragma unroll
        for (int dx = 0; dx < 3; ++dx) {
          i32x4 nx[2];
          if (DBG >= 5 && blk > 0) {
            nx[0] = ac[1]; nx[1] = ac[0];
            if (dx == 2) sync(false);
          } else if (dx < 2) {
#pragma unroll
            for (int n = 0; n < 2; ++n) nx[n] = lda(cbo, (dx + 1) * 4 + 2 * h + n);
          } else {
            sync(false);
#pragma unroll
            for (int n = 0; n < 2; ++n) nx[n] = lda(nbo, 2 * h + n);
          }
#pragma unroll
          for (int m = 0; m < 4; ++m) {
#pragma unroll
            for (int n = 0; n < 2; ++n) mma16<T>(acc[m][n], ac[n], B[m + dy][dx]);
            if (dx == 2 && m == 2 * h) refill();
          }
#pragma unroll
          for (int n = 0; n < 2; ++n) ac[n] = nx[n];
        }
        next_buf();
      }
    }
#pragma unroll
    for (int n = 0; n < 2; ++n) an[n] = ac[n];
    if constexpr (DBG >= 6) {
#pragma unroll
      for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int n = 0; n < 2; ++n) accs[0][m] += acc[m][n];   // (keeps the MFMAs alive)
      next_buf();
    } else {
      finish_block(acc, a.E1 + blk * 64);
    }
  }
  CTL(5);
  vm_wait<0>();   // the dummy stages have landed (nobody reads them) before this workgroup's LDS is handed on
  CTL(6);

  // ---------------------------------------------------------------- next squeeze: bias + ReLU -> sq_out
  if constexpr (NSQ > 0) {
    T* so = reinterpret_cast<T*>(a.sq_out);
    const float* bs = bl + a.E1 + a.E3;
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      const int oy = oy0 + r0 + m;
      if (col_ok && oy < a.H) {
        T* dst = so + ((size_t)(n_img * a.H + oy) * a.W + ox) * a.S2 + g * 4 * NSQ;
#pragma unroll
        for (int i = 0; i < TH; ++i) {
          const int t = h * TH + i;
          if (t < NSQ) {
            const f32x4 bias = *reinterpret_cast<const f32x4*>(bs + g * 4 * NSQ + t * 4);
            store4<T>(dst + t * 4, relu4(accs[i][m] + bias));
          }
        }
      }
    }
  }
#ifdef SQDET_CHAIN_TIMELINE
  CTL(7);
  if (threadIdx.x == 0 && blockIdx.x < 4096) {
#pragma unroll
    for (int k = 0; k < 8; ++k) g_chain_tl[(size_t)blockIdx.x * 8 + k] = ctl[k];
  }
#endif
}

// ---- weight stream: float32 HWIO kernels -> the ring stages in consumption order ----
// stage = 3 slots x 4 fragments x 64 lanes x 16 B.  Element (stage, slot, tile, lane = (i, g), e):
//   expand stage: cout = block*64 + (tile>>1)*32 + (i>>2)*8 + (tile&1)*4 + (i&3), cin = chunk*32 + g*8 + e
//     (expand1x1: slot = chunk; expand3x3 stage (chunk, dy): slot = dx)
//   chain stage of concat channels cc0..cc0+64: fragment f = slot*4 + tile = u*NSQ + t:
//     cout' = (i>>2)*4*NSQ + t*4 + (i&3) (the squeeze conv's own packing), cin' = cc0 + u*32 + g*8 + e
__global__ void chain_pack_kernel(const float* __restrict__ w1, const float* __restrict__ w3, const float* __restrict__ ws2,
                                  f16* __restrict__ out, int S, int E1, int E3, int S2, ChainGeom gm, size_t total) {
  for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
    size_t t = idx;
    const int e = t % 8; t /= 8;
    const int lane = t % 64; t /= 64;
    const int tile = t % 4; t /= 4;
    const int slot = t % 3; t /= 3;
    const int stage = (int)t;
    const int i = lane & 15, g = lane >> 4;
    int kind, blk, r;   // kind 0: expand1x1, 1: expand3x3 (r = chunk*3 + dy), 2: chain
    int cc0 = 0;
    if (stage < gm.base3) {
      blk = stage / (1 + gm.chain);
      r = stage - blk * (1 + gm.chain);
      kind = r == 0 ? 0 : 2;
      cc0 = blk * 64;
    } else {
      const int k2 = stage - gm.base3;
      blk = k2 / gm.per3;
      r = k2 - blk * gm.per3;
      kind = r < 3 * gm.nch ? 1 : 2;
      cc0 = E1 + blk * 64;
    }
    const float* src = kind == 0 ? w1 : (kind == 1 ? w3 : ws2);
    if (!src) continue;   // this part of the stream is not being (re)written
    float v = 0.f;
    if (kind == 2) {
      const int f = slot * 4 + tile;
      if (f < 2 * gm.nsq) {
        const int u = f / gm.nsq, tt = f - u * gm.nsq;
        const int co = (i >> 2) * 4 * gm.nsq + tt * 4 + (i & 3);
        const int ci = cc0 + u * 32 + g * 8 + e;
        v = ws2[(size_t)ci * S2 + co];
      }
    } else {
      const int co = blk * 64 + (tile >> 1) * 32 + (i >> 2) * 8 + (tile & 1) * 4 + (i & 3);
      if (kind == 0) {
        const int ci = slot * 32 + g * 8 + e;
        if (slot < gm.nch && ci < S) v = w1[(size_t)ci * E1 + co];
      } else {
        const int c = r / 3, dy = r - c * 3;
        const int ci = c * 32 + g * 8 + e;
        if (ci < S) v = w3[((size_t)(dy * 3 + slot) * S + ci) * E3 + co];
      }
    }
    out[idx] = (f16)v;
  }
}

// =====================================================================================================================
// Persistent form for the LARGE maps with one-chunk squeezes (fire2 .. fire5 of SqueezeDet: 94x311 / 47x156, squeeze 16 / 32,
// expand 64 / 128).  There a 256-pixel workgroup has ~100 MFMAs per wave to do: the ring kernel above spends its time
// filling the ring and passing barriers (measured: no faster than the streaming fused fire).  Here the weights -- 44-104
// KiB -- are RESIDENT in LDS (copied once per workgroup from the same packed stream), the workgroups are persistent
// (one per CU) and walk the map four 8x16 tiles at a time: wave w owns tile slot w>>1, rows 4*(w&1)..+4, and -- since
// nothing is shared but the weights -- ALL couts of its 64 pixels, so the chained squeeze needs no exchange.  The next
// four tiles' squeeze halos (46 KiB) are prefetched into registers while the current ones are computed; two barriers
// per four tiles.  Accumulation orders are the ring kernel's (canonical): bitwise the separate convs.
struct ChainSArgs {
  const void* sq_in;
  void* sq_out;
  const unsigned char* stream;
  const float *b1, *b3, *bs2;
  int N, H, W, S, E1, E3, S2;
  int tiles_x, tiles_y, ntiles, nquads;
  int nb1, nb3;
  unsigned in_bytes;
};

constexpr int CS_TILES = 4;                                   // tiles per workgroup step
// LDS row pitch of a squeeze tile: 24 pixels, not 18 (as in fire2.hip): with a pitch that is a multiple of 8 the swizzle
// term ((pixel >> 1) & 3) of a fragment read does not depend on the row, so the reads of one column shift share ONE
// address register + immediate row offsets
constexpr int CS_LW = 24;
constexpr int CS_TILE_B = (CROWS + 2) * CS_LW * 64;           // bytes of one tile slot (15360)

template <int NSQ>
__global__ __launch_bounds__(512, 2) void fire_chain_stream(ChainSArgs a) {
  using T = f16;
  constexpr int F1 = 4 + 2 * NSQ;                             // resident fragments per expand1x1 block (+ its chain)
  constexpr int F3 = 36 + 2 * NSQ;                            // ... per expand3x3 block
  constexpr int NP = 4;                                       // 16-byte pieces per pixel of the (zero padded) chunk
  constexpr int SIT = (CHP * NP + 127) / 128;                 // pieces per thread per step: a tile slot is fetched by its two waves
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int j = lane & 15, g = lane >> 4;
  const int nfrag = a.nb1 * F1 + a.nb3 * F3;
  unsigned char* wres = lds;                                  // [fragment][64 lanes][16 B]
  unsigned char* stile = lds + (size_t)nfrag * 1024;          // [tile slot][halo row][24 pixels][4 x 16 B swizzled]
  float* bl = reinterpret_cast<float*>(stile + CS_TILES * CS_TILE_B);

  // ---- one-time: weights (the valid fragments of the packed stream's stages) and biases -> LDS
  {
    const int base3s = a.nb1 * 2;                             // stream stage of the first expand3x3 block (NCH = 1)
    for (int f = wave; f < nfrag; f += 8) {
      int stage, fs;
      if (f < a.nb1 * F1) {
        const int blk = f / F1, r = f - blk * F1;
        stage = blk * 2 + (r < 4 ? 0 : 1);
        fs = r < 4 ? r : r - 4;
      } else {
        const int f2 = f - a.nb1 * F1;
        const int blk = f2 / F3, r = f2 - blk * F3;
        stage = base3s + blk * 4 + (r < 36 ? r / 12 : 3);
        fs = r < 36 ? r % 12 : r - 36;
      }
      reinterpret_cast<i32x4*>(wres)[(size_t)f * 64 + lane] =
          reinterpret_cast<const i32x4*>(a.stream + (size_t)stage * STAGE_B)[fs * 64 + lane];
    }
    const int nbias = a.E1 + a.E3 + a.S2;
    for (int i = threadIdx.x; i < nbias; i += 512)
      bl[i] = i < a.E1 ? a.b1[i] : (i < a.E1 + a.E3 ? a.b3[i - a.E1] : a.bs2[i - a.E1 - a.E3]);
  }

  // every XCD owns a contiguous band of tile quads (halo re-reads stay in one L2)
  const int xcd = blockIdx.x & 7, lid = blockIdx.x >> 3, nl = gridDim.x >> 3;   // gridDim.x is a multiple of 8
  const int per = (a.nquads + 7) >> 3;
  const int band_end = min(a.nquads, (xcd + 1) * per);
  int quad = xcd * per + lid;

  const __amdgpu_buffer_rsrc_t rin = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.sq_in), 0, a.in_bytes, 0x00020000);
  constexpr unsigned OOB = 0xfffffff0u;
  const int s_pieces = a.S * 2 / 16;
  const int ts = wave >> 1, r0 = (wave & 1) * 4;              // this wave's tile slot and first tile row
  const int tl = (int)(threadIdx.x & 127);                    // index among the two waves of the slot
  i32x4 sv[SIT];
  auto prefetch = [&](int q) {       // this slot's squeeze halo of quad q -> registers (out of range = the zero padding)
    int t = q * CS_TILES + ts;       // (wave-uniform tile decode)
    const bool tile_ok = q < band_end && t < a.ntiles;
    const int tx = t % a.tiles_x; t /= a.tiles_x;
    const int ty = t % a.tiles_y;
    const int n = t / a.tiles_y;
    const int pix0 = (n * a.H + ty * CROWS - 1) * a.W + tx * CCOLS - 1;
#pragma unroll
    for (int it = 0; it < SIT; ++it) {
      const int idx = it * 128 + tl;
      const int P = idx / NP, pc = idx - P * NP;
      const int r = P / (CCOLS + 2), c = P - r * (CCOLS + 2);
      const int iy = ty * CROWS - 1 + r, ix = tx * CCOLS - 1 + c;
      const bool ok = tile_ok && idx < CHP * NP && pc < s_pieces && iy >= 0 && iy < a.H && ix >= 0 && ix < a.W;
      const unsigned off = ok ? (unsigned)(((pix0 + r * a.W + c) * a.S) * 2 + pc * 16) : OOB;
      sv[it] = __builtin_amdgcn_raw_buffer_load_b128(rin, off, 0, 0);
    }
  };
  prefetch(quad);

  unsigned char* simg = stile + ts * CS_TILE_B;
  // B fragment (halo row r0 + rr, column shift dx): the swizzle term depends on the column only (pitch 24)
  const unsigned char* bbase[3];
#pragma unroll
  for (int dx = 0; dx < 3; ++dx) {
    const int P0 = r0 * CS_LW + j + dx;
    bbase[dx] = simg + P0 * 64 + ((g ^ ((P0 >> 1) & 3)) << 4);
  }
  auto load_b = [&](int rr, int dx) { return *reinterpret_cast<const i32x4*>(bbase[dx] + rr * (CS_LW * 64)); };
  auto wfrag = [&](int f) { return *reinterpret_cast<const i32x4*>(wres + (size_t)f * 1024 + lane * 16); };
  T* so = reinterpret_cast<T*>(a.sq_out);

  for (; quad < band_end; quad += nl) {
    // ---- the prefetched halo -> this slot's LDS tile
#pragma unroll
    for (int it = 0; it < SIT; ++it) {
      const int idx = it * 128 + tl;
      const int P = idx / NP, pc = idx - P * NP;
      const int r = P / (CCOLS + 2), c = P - r * (CCOLS + 2);
      const int PL = r * CS_LW + c;
      if (idx < CHP * NP) *reinterpret_cast<i32x4*>(simg + PL * 64 + ((pc ^ ((PL >> 1) & 3)) << 4)) = sv[it];
    }
    __syncthreads();
    prefetch(quad + nl);              // lands while this quad is computed
    int t = quad * CS_TILES + ts;
    const bool tile_ok = t < a.ntiles;
    const int tx = t % a.tiles_x; t /= a.tiles_x;
    const int ty = t % a.tiles_y;
    const int n_img = t / a.tiles_y;
    const int oy0 = ty * CROWS, ox = tx * CCOLS + j;
    const bool col_ok = tile_ok && ox < a.W;

    f32x4 accs[NSQ][4];
#pragma unroll
    for (int tq = 0; tq < NSQ; ++tq)
#pragma unroll
      for (int m = 0; m < 4; ++m) accs[tq][m] = f32x4{0.f, 0.f, 0.f, 0.f};
    // bias + ReLU + float16 rounding of a 64-cout block; its two 32-cout pairs are two K chunks of the next squeeze
    auto finish_block = [&](f32x4 (&acc)[4][4], int cc0, int fchain) {
      i32x4 bf[4][2];
#pragma unroll
      for (int p = 0; p < 2; ++p) {
        const f32x4 bias0 = *reinterpret_cast<const f32x4*>(bl + cc0 + p * 32 + g * 8);
        const f32x4 bias1 = *reinterpret_cast<const f32x4*>(bl + cc0 + p * 32 + g * 8 + 4);
#pragma unroll
        for (int m = 0; m < 4; ++m) bf[m][p] = pack8(relu4(acc[m][2 * p] + bias0), relu4(acc[m][2 * p + 1] + bias1));
      }
#pragma unroll
      for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int tq = 0; tq < NSQ; ++tq) {
          const i32x4 fr = wfrag(fchain + u * NSQ + tq);
#pragma unroll
          for (int m = 0; m < 4; ++m) mma16<T>(accs[tq][m], fr, bf[m][u]);
        }
    };
    // ---- expand1x1 blocks (centre tap)
    {
      i32x4 b1f[4];
#pragma unroll
      for (int m = 0; m < 4; ++m) b1f[m] = load_b(m + 1, 1);
#pragma unroll 1
      for (int blk = 0; blk < a.nb1; ++blk) {
        f32x4 acc[4][4];
        i32x4 af[4];
#pragma unroll
        for (int tt = 0; tt < 4; ++tt) af[tt] = wfrag(blk * F1 + tt);
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
          for (int tt = 0; tt < 4; ++tt) {
            acc[m][tt] = f32x4{0.f, 0.f, 0.f, 0.f};
            mma16<T>(acc[m][tt], af[tt], b1f[m]);
          }
        finish_block(acc, blk * 64, blk * F1 + 4);
      }
    }
    // ---- expand3x3 blocks
#pragma unroll 1
    for (int blk = 0; blk < a.nb3; ++blk) {
      f32x4 acc[4][4];
#pragma unroll
      for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int tt = 0; tt < 4; ++tt) acc[m][tt] = f32x4{0.f, 0.f, 0.f, 0.f};
      const int f0 = a.nb1 * F1 + blk * F3;
      // nine taps, the next tap's eight LDS fragments requested before the current tap's 16 MFMAs (two register sets;
      // the scheduling barriers keep hipcc from hoisting every tap's reads to the top, which spilled)
      i32x4 Bc[4], ac[4];
#pragma unroll
      for (int m = 0; m < 4; ++m) Bc[m] = load_b(m, 0);
#pragma unroll
      for (int tt = 0; tt < 4; ++tt) ac[tt] = wfrag(f0 + tt);
#pragma unroll
      for (int tap = 0; tap < 9; ++tap) {
        i32x4 Bn[4], an[4];
        if (tap + 1 < 9) {
          const int dy = (tap + 1) / 3, dx = (tap + 1) % 3;
#pragma unroll
          for (int m = 0; m < 4; ++m) Bn[m] = load_b(m + dy, dx);
#pragma unroll
          for (int tt = 0; tt < 4; ++tt) an[tt] = wfrag(f0 + (tap + 1) * 4 + tt);
        }
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
          for (int tt = 0; tt < 4; ++tt) mma16<T>(acc[m][tt], ac[tt], Bc[m]);
        __builtin_amdgcn_sched_barrier(0);
        if (tap + 1 < 9) {
#pragma unroll
          for (int m = 0; m < 4; ++m) Bc[m] = Bn[m];
#pragma unroll
          for (int tt = 0; tt < 4; ++tt) ac[tt] = an[tt];
        }
      }
      finish_block(acc, a.E1 + blk * 64, f0 + 36);
    }
    // ---- next squeeze: bias + ReLU -> sq_out
    {
      const float* bs = bl + a.E1 + a.E3;
#pragma unroll
      for (int m = 0; m < 4; ++m) {
        const int oy = oy0 + r0 + m;
        if (col_ok && oy < a.H) {
          T* dst = so + ((size_t)(n_img * a.H + oy) * a.W + ox) * a.S2 + g * 4 * NSQ;
#pragma unroll
          for (int tq = 0; tq < NSQ; ++tq) {
            const f32x4 bias = *reinterpret_cast<const f32x4*>(bs + g * 4 * NSQ + tq * 4);
            store4<T>(dst + tq * 4, relu4(accs[tq][m] + bias));
          }
        }
      }
    }
    __syncthreads();                  // every wave is done with the squeeze tiles before they are overwritten
  }
}

bool fire_chain_stream_shape(int s, int e1, int e3, int s2) {
  return s > 0 && s <= 32 && s % 8 == 0 && e1 % 64 == 0 && e3 % 64 == 0 && e1 > 0 && e3 > 0 && (s2 == 16 || s2 == 32 || s2 == 48) &&
         (size_t)((e1 / 64) * (4 + s2 / 8) + (e3 / 64) * (36 + s2 / 8)) * 1024 + CS_TILES * CS_TILE_B + (size_t)(e1 + e3 + s2) * 4 <= 160 * 1024;
}

template <int NSQ>
int launch_chain_stream(const ChainSArgs& a, hipStream_t st) {
  const size_t lds = (size_t)(a.nb1 * (4 + 2 * NSQ) + a.nb3 * (36 + 2 * NSQ)) * 1024 + CS_TILES * CS_TILE_B + (size_t)(a.E1 + a.E3 + a.S2) * 4;
  static PerDevice once;
  SQDET_CHECK_HIP(once.run([] { return hipFuncSetAttribute(reinterpret_cast<const void*>(&fire_chain_stream<NSQ>),
                                                           hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); }));
  int grid = cu_count();                                    // one persistent workgroup per CU
  if (grid > (a.nquads + 7) / 8 * 8) grid = (a.nquads + 7) / 8 * 8;
  hipLaunchKernelGGL((fire_chain_stream<NSQ>), dim3((unsigned)grid), dim3(512), lds, st, a);
  SQDET_CHECK_HIP(hipGetLastError());
  return SQDET_OK;
}

constexpr size_t chain_lds_bytes(int nch, int nsq, int ring, int nbias) {
  return (size_t)ring * STAGE_B + (size_t)NIMG * nch * CCHUNK + (nsq > 0 ? 32768 : 0) + (size_t)nbias * 4;
}
// deepest ring the 160 KiB allow (the squeeze tile of 3 chunks + the exchange area leave room for 4 stages only)
constexpr int chain_ring(int nch, int nsq) { return chain_lds_bytes(nch, nsq, 6, 768 + 96) <= 160 * 1024 ? 6 : 4; }

template <int NCH, int NSQ, bool WY>
int launch_chain(const ChainArgs& a, hipStream_t st) {
  constexpr int RG = chain_ring(NCH, NSQ);
  const size_t lds = chain_lds_bytes(NCH, NSQ, RG, a.E1 + a.E3 + a.S2);
  static PerDevice once;
  SQDET_CHECK_HIP(once.run([] { return hipFuncSetAttribute(reinterpret_cast<const void*>(&fire_chain<NCH, NSQ, WY, RG>),
                                                           hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); }));
  const int wgs = ((a.N + 1) / 2) * a.tiles_x * a.tiles_y;
  ChainArgs& am = const_cast<ChainArgs&>(a);
  am.per_xcd = (wgs + 7) / 8;
  const int riders_per_xcd = a.ride.nimg > 0 ? (a.ride.nriders + 7) / 8 : 0;
  const dim3 grid((unsigned)(8 * (am.per_xcd + riders_per_xcd)));
  static_assert(sizeof(FastLds<512>) <= RG * STAGE_B, "the rider's state lives in the ring's LDS");
#ifdef SQDET_CHAIN_DBG   // experiment builds only (tools/chainbench.py --dbg): cost ladder of the fire10 -> fire11 shape
  if constexpr (NCH == 3 && NSQ == 6 && !WY) {
    const int d = tune(TUNE_DBG);
    static bool dbg_attr = false;
    if (!dbg_attr) {
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&fire_chain<NCH, NSQ, WY, RG, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&fire_chain<NCH, NSQ, WY, RG, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&fire_chain<NCH, NSQ, WY, RG, 3>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&fire_chain<NCH, NSQ, WY, RG, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&fire_chain<NCH, NSQ, WY, RG, 5>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&fire_chain<NCH, NSQ, WY, RG, 6>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
      dbg_attr = true;
    }
    switch (d) {
      case 1: hipLaunchKernelGGL((fire_chain<NCH, NSQ, WY, RG, 1>), grid, dim3(512), lds, st, a); return SQDET_OK;
      case 2: hipLaunchKernelGGL((fire_chain<NCH, NSQ, WY, RG, 2>), grid, dim3(512), lds, st, a); return SQDET_OK;
      case 3: hipLaunchKernelGGL((fire_chain<NCH, NSQ, WY, RG, 3>), grid, dim3(512), lds, st, a); return SQDET_OK;
      case 4: hipLaunchKernelGGL((fire_chain<NCH, NSQ, WY, RG, 4>), grid, dim3(512), lds, st, a); return SQDET_OK;
      case 5: hipLaunchKernelGGL((fire_chain<NCH, NSQ, WY, RG, 5>), grid, dim3(512), lds, st, a); return SQDET_OK;
      case 6: hipLaunchKernelGGL((fire_chain<NCH, NSQ, WY, RG, 6>), grid, dim3(512), lds, st, a); return SQDET_OK;
      default: break;
    }
  }
#endif
  hipLaunchKernelGGL((fire_chain<NCH, NSQ, WY, RG>), grid, dim3(512), lds, st, a);
  SQDET_CHECK_HIP(hipGetLastError());
  return SQDET_OK;
}

template <int NCH, bool WY>
int dispatch_chain_nsq(const ChainArgs& a, int nsq, hipStream_t st) {
  // (instantiated: the squeeze widths SqueezeDet pairs with each squeeze depth class -- fire2..5: 16 / 32 / 48 behind a
  //  one-chunk squeeze; fire6..11: 48 / 64 / 96 behind two or three chunks)
  if (nsq == 0) { if constexpr (WY) return launch_chain<NCH, 0, true>(a, st); }
  if constexpr (NCH == 1) {
    switch (nsq) {
      case 1: return launch_chain<NCH, 1, WY>(a, st);
      case 2: return launch_chain<NCH, 2, WY>(a, st);
      case 3: return launch_chain<NCH, 3, WY>(a, st);
      default: break;
    }
  } else {
    switch (nsq) {
      case 3: return launch_chain<NCH, 3, WY>(a, st);
      case 4: return launch_chain<NCH, 4, WY>(a, st);
      case 6: return launch_chain<NCH, 6, WY>(a, st);
      default: break;
    }
  }
  set_error("fire_chain: unsupported next-squeeze width");
  return SQDET_EUNSUPPORTED;
}

}  // namespace

bool fire_chain_eligible(int s, int e1, int e3, int s2, int dtype) {
  if (dtype != SQDET_F16 || conv_algo() != 0) return false;
  if (s <= 0 || s % 8 != 0 || s > 96) return false;
  if (e1 <= 0 || e3 <= 0 || e1 % 64 != 0 || e3 % 64 != 0) return false;
  if (!(s2 == 0 || s2 == 16 || s2 == 32 || s2 == 48 || s2 == 64 || s2 == 96)) return false;
  const ChainGeom g = chain_geom(s, e1, e3, s2);
  if (g.nch < 1 || g.nch > 3) return false;
  if (e1 + e3 + s2 > 768 + 96) return false;   // (the ring depth is chosen for at most this many biases)
  return chain_lds_bytes(g.nch, g.nsq, chain_ring(g.nch, g.nsq), e1 + e3 + s2) <= 160 * 1024;
}

}  // namespace sqdet

using namespace sqdet;

extern "C" size_t sqdet_fire_chain_stream_bytes(int s1x1, int e1x1, int e3x3, int next_s1x1, int dtype) {
  if (!fire_chain_eligible(s1x1, e1x1, e3x3, next_s1x1, dtype)) return 0;
  return (size_t)(chain_geom(s1x1, e1x1, e3x3, next_s1x1).nstages + 5) * STAGE_B;   // + the dummy stages (up to LOOK + 1 = 5)
}

extern "C" int sqdet_fire_chain_pack(const float* w_e1_hwio, const float* w_e3_hwio, const float* w_next_s_hwio,
                                     void* stream_buf, int s1x1, int e1x1, int e3x3, int next_s1x1, int dtype,
                                     sqdet_stream_t stream) {
  SQDET_REQUIRE(stream_buf, "fire_chain_pack: null stream buffer");
  SQDET_UNSUPPORTED(!fire_chain_eligible(s1x1, e1x1, e3x3, next_s1x1, dtype), "fire_chain_pack: shape/dtype not covered");
  const ChainGeom g = chain_geom(s1x1, e1x1, e3x3, next_s1x1);
  const size_t total = (size_t)g.nstages * STAGE_B / 2;
  const int blocks = (int)((total + 255) / 256 > 4096 ? 4096 : (total + 255) / 256);
  hipLaunchKernelGGL(chain_pack_kernel, dim3(blocks), dim3(256), 0, as_stream(stream), w_e1_hwio, w_e3_hwio,
                     next_s1x1 > 0 ? w_next_s_hwio : nullptr, (f16*)stream_buf, s1x1, e1x1, e3x3, next_s1x1, g, total);
  SQDET_CHECK_HIP(hipGetLastError());
  return SQDET_OK;
}

extern "C" int sqdet_fire_chain_fwd(const void* sq_in, const void* stream_buf, const float* b_e1, const float* b_e3,
                                    const float* b_next_s, void* y, void* sq_out, int n, int h, int w, int s1x1,
                                    int e1x1, int e3x3, int next_s1x1, int dtype, sqdet_stream_t stream) {
  return sqdet::fire_chain_launch_ride(sq_in, stream_buf, b_e1, b_e3, b_next_s, y, sq_out, n, h, w, s1x1, e1x1, e3x3, next_s1x1, dtype,
                                       nullptr, as_stream(stream));
}

int sqdet::fire_chain_idle_cus(int n, int h, int w, int s1x1, int e1x1, int e3x3, int next_s1x1, int dtype) {
  if (!fire_chain_eligible(s1x1, e1x1, e3x3, next_s1x1, dtype)) return 0;
  const ChainGeom g = chain_geom(s1x1, e1x1, e3x3, next_s1x1);
  const long px = (long)n * h * w;
  if (g.nch == 1 && px > 100000) return 0;                       // (the persistent form fills every CU)
  const int wgs = ((n + 1) / 2) * ((w + CCOLS - 1) / CCOLS) * ((h + CROWS - 1) / CROWS);
  const int per = (wgs + 7) / 8;
  return per < 32 ? (32 - per) * 8 : 0;                          // one chain workgroup per CU (its LDS), 32 CUs per XCD
}

int sqdet::fire_chain_launch_ride(const void* sq_in, const void* stream_buf, const float* b_e1, const float* b_e3,
                                  const float* b_next_s, void* y, void* sq_out, int n, int h, int w, int s1x1,
                                  int e1x1, int e3x3, int next_s1x1, int dtype, const ChainRide* ride, hipStream_t st_in) {
  const sqdet_stream_t stream = reinterpret_cast<sqdet_stream_t>(st_in);
  SQDET_REQUIRE(sq_in && stream_buf && b_e1 && b_e3, "fire_chain_fwd: null pointer");
  SQDET_REQUIRE(n > 0 && h > 0 && w > 0, "fire_chain_fwd: bad dims");
  SQDET_REQUIRE((next_s1x1 > 0) == (sq_out != nullptr) && (next_s1x1 == 0 || b_next_s), "fire_chain_fwd: sq_out / next_s1x1 mismatch");
  SQDET_REQUIRE(y || sq_out, "fire_chain_fwd: no output");
  SQDET_UNSUPPORTED(!fire_chain_eligible(s1x1, e1x1, e3x3, next_s1x1, dtype), "fire_chain_fwd: shape/dtype not covered");
  const long px = (long)n * h * w;
  SQDET_UNSUPPORTED(px * (e1x1 + e3x3) * 2 >= (1L << 31), "fire_chain_fwd: tensor too large for 32-bit offsets");
  const ChainGeom g = chain_geom(s1x1, e1x1, e3x3, next_s1x1);
  ChainArgs a;
  a.sq_in = sq_in; a.sq_out = sq_out; a.y = y; a.stream = reinterpret_cast<const unsigned char*>(stream_buf);
  a.b1 = b_e1; a.b3 = b_e3; a.bs2 = b_next_s;
  a.N = n; a.H = h; a.W = w; a.S = s1x1; a.E1 = e1x1; a.E3 = e3x3; a.S2 = next_s1x1;
  a.tiles_x = (w + CCOLS - 1) / CCOLS; a.tiles_y = (h + CROWS - 1) / CROWS;
  a.nb1 = g.nb1; a.nb3 = g.nb3; a.nstages = g.nstages;
  a.in_bytes = (unsigned)(px * s1x1 * 2);
  a.out_bytes = (unsigned)(px * next_s1x1 * 2);
  a.y_bytes = (unsigned)(px * (e1x1 + e3x3) * 2);
  a.per_xcd = 0;
  if (ride) a.ride = *ride; else { a.ride = ChainRide{}; a.ride.nimg = 0; a.ride.nriders = 0; a.ride.img0 = 0; }
  hipStream_t st = as_stream(stream);
  // large maps with a one-chunk squeeze: the persistent, weights-resident form ("dbg" 30 keeps the ring kernel for
  // A/B, 31 takes the persistent form at any size -- tests)
  if (!y && g.nch == 1 && (px > 100000 || tune(TUNE_DBG) == 31) && fire_chain_stream_shape(s1x1, e1x1, e3x3, next_s1x1) && tune(TUNE_DBG) != 30) {
    SQDET_REQUIRE(a.ride.nimg == 0, "fire_chain: the persistent form carries no riders");
    ChainSArgs c;
    c.sq_in = sq_in; c.sq_out = sq_out; c.stream = a.stream; c.b1 = b_e1; c.b3 = b_e3; c.bs2 = b_next_s;
    c.N = n; c.H = h; c.W = w; c.S = s1x1; c.E1 = e1x1; c.E3 = e3x3; c.S2 = next_s1x1;
    c.tiles_x = a.tiles_x; c.tiles_y = a.tiles_y;
    const long nt = (long)n * a.tiles_x * a.tiles_y;
    c.ntiles = (int)nt; c.nquads = (int)((nt + CS_TILES - 1) / CS_TILES);
    c.nb1 = g.nb1; c.nb3 = g.nb3; c.in_bytes = a.in_bytes;
    switch (g.nsq) {
      case 1: return launch_chain_stream<1>(c, st);
      case 2: return launch_chain_stream<2>(c, st);
      case 3: return launch_chain_stream<3>(c, st);
      default: break;
    }
  }
  if (g.nch == 1) return y ? dispatch_chain_nsq<1, true>(a, g.nsq, st) : dispatch_chain_nsq<1, false>(a, g.nsq, st);
  if (g.nch == 2) return y ? dispatch_chain_nsq<2, true>(a, g.nsq, st) : dispatch_chain_nsq<2, false>(a, g.nsq, st);
  return y ? dispatch_chain_nsq<3, true>(a, g.nsq, st) : dispatch_chain_nsq<3, false>(a, g.nsq, st);
}

#ifdef SQDET_CHAIN_TIMELINE
extern "C" int sqdet_debug_chain_timeline(unsigned long long* host, int count) {
  return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(sqdet::g_chain_tl), sizeof(unsigned long long) * count);
}
#endif
