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

namespace sqdet {

namespace {

constexpr int CROWS = 8, CCOLS = 16;
constexpr int CHP = (CROWS + 2) * (CCOLS + 2);   // 180 halo pixels of one image's tile
constexpr int CCHUNK = CHP * 64;                 // bytes of one 64-byte K chunk of a tile
constexpr int RING = 6;                          // ring stages
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
  int dbg;   // experiments (sqdet_set_option "dbg"): 50 no vmcnt waits, 51 no weight stream at all, 52 no barriers either, 55 prologue only
};

// One 1-KiB piece of the weight stream straight into LDS (no registers).  Hidden from hipcc's wait-count pass on
// purpose: its completion is counted by hand (vm_wait below).  M0 = the wave-uniform LDS byte address.
__device__ __forceinline__ void glds16(const unsigned char* gsrc, unsigned lds_dst) {
  unsigned keep;
  asm volatile(
      "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
      : "=&s"(keep)
      : "v"(gsrc), "s"(lds_dst)
      : "memory");
}

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
// WY: the concat tensor is written.
template <int NCH, int NSQ, bool WY>
__global__ __launch_bounds__(256, 1) void fire_chain(ChainArgs a) {
  using T = f16;
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int j = lane & 15, g = lane >> 4;

  int b = (int)((blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3));   // XCD-banded order (gridDim.x % 8 == 0)
  const int npairs = (a.N + 1) >> 1;
  if (b >= npairs * a.tiles_x * a.tiles_y || a.dbg == 56) return;
  const int tx = b % a.tiles_x; b /= a.tiles_x;
  const int ty = b % a.tiles_y;
  const int np = b / a.tiles_y;
  const int oy0 = ty * CROWS, ox0 = tx * CCOLS;

  unsigned char* ring = lds;
  unsigned char* stile = lds + RING * STAGE_B;                      // [img][chunk][pixel][4 x 16 B swizzled]
  float* bl = reinterpret_cast<float*>(stile + NIMG * NCH * CCHUNK);   // biases [b1 | b3 | bs2]
  const unsigned ring_addr = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)ring;

  // ---------------------------------------------------------------- squeeze tile (both images, with halo): loads
  constexpr int NP = NCH * 4;                         // 16-byte pieces per pixel in LDS (zero padded)
  constexpr int SIT = (NIMG * CHP * NP + 255) / 256;  // pieces per thread
  i32x4 sv[SIT];
  {
    const int s_pieces = a.S * 2 / 16;
    const __amdgpu_buffer_rsrc_t rin = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.sq_in), 0, a.in_bytes, 0x00020000);
    constexpr unsigned OOB = 0xfffffff0u;
#pragma unroll
    for (int it = 0; it < SIT; ++it) {
      const int idx = it * 256 + (int)threadIdx.x;
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
  // (the stream carries three dummy stages behind the last real one: exactly three stages are in flight behind
  // the one being waited for at EVERY sync, so the wait count is a constant)
  const unsigned char* gl = a.stream + wave * 1024 + lane * 16;     // this lane's 16 bytes of every slot's quarter
  const int nissue = a.nstages + 3;
  int ks = 0;        // stages made available so far (= index of the next one to wait for)
  int ib = 0;        // ring buffer the next refill goes to
  int pend = -1;     // stage whose refill is due (issued a few MFMAs behind the barrier, a different few per wave)
  auto issue = [&](int stage) {
    if (a.dbg == 51 || a.dbg == 52) return;
    const unsigned char* src = gl + (size_t)stage * STAGE_B;
    const unsigned dst = ring_addr + (unsigned)ib * STAGE_B + (unsigned)wave * 1024;
    glds16(src, dst);
    glds16(src + 4096, dst + 4096);
    glds16(src + 8192, dst + 8192);
    ib = ib + 1 == RING ? 0 : ib + 1;
  };
  auto refill = [&]() {
    if (pend >= 0) { issue(pend); pend = -1; }
  };
#pragma unroll
  for (int s = 0; s < RING - 2; ++s) issue(s);   // (nissue >= 4 always)

  // ---------------------------------------------------------------- squeeze tile + biases -> LDS
  {
#pragma unroll
    for (int it = 0; it < SIT; ++it) {
      const int idx = it * 256 + (int)threadIdx.x;
      const int im = idx / (CHP * NP);
      const int rem = idx - im * (CHP * NP);
      const int P = rem / NP, q = rem - P * NP;
      if (idx < NIMG * CHP * NP)
        *reinterpret_cast<i32x4*>(stile + (im * NCH + (q >> 2)) * CCHUNK + P * 64 + (((q & 3) ^ ((P >> 1) & 3)) << 4)) = sv[it];
    }
    const int nbias = a.E1 + a.E3 + a.S2;
    for (int i = threadIdx.x; i < nbias; i += 256)
      bl[i] = i < a.E1 ? a.b1[i] : (i < a.E1 + a.E3 ? a.b3[i - a.E1] : a.bs2[i - a.E1 - a.E3]);
  }
  __syncthreads();

  const int img = wave >> 1, r0 = (wave & 1) * 4;
  const unsigned char* simg = stile + img * NCH * CCHUNK;
  // B fragment of (chunk c, halo row rr of this wave = tile row r0 - 1 + rr, column shift dx)
  auto load_b = [&](int c, int rr, int dx) {
    const int P = (r0 + rr) * (CCOLS + 2) + j + dx;
    return *reinterpret_cast<const i32x4*>(simg + c * CCHUNK + P * 64 + ((g ^ ((P >> 1) & 3)) << 4));
  };
  int cb = 0;    // ring buffer of the stage being consumed
  auto lda = [&](int buf, int f) {
    return *reinterpret_cast<const i32x4*>(ring + buf * STAGE_B + f * 1024 + lane * 16);
  };
  int ep_ks = -100;   // ks at the time of the last concat-tensor stores (WY only)
  // Makes the NEXT stage available: every wave's pieces of it have landed (own vmcnt, then the barrier), and
  // every wave is past stage ks-2 -- its ring buffer takes the refill (issued by refill(), a few MFMAs later).
  // Three younger stages (9 pieces) are in flight behind it; the 8 concat-tensor stores of a block epilogue are
  // younger than stage ks's pieces for the next four syncs (loads and stores retire in issue order).
  auto sync = [&]() {
    if (a.dbg < 50 || a.dbg > 52) {
      if (WY && (ks - ep_ks) <= 3) vm_wait<17>();
      else vm_wait<9>();
    }
    if (a.dbg != 52) __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    const int nk = ks + RING - 2;
    pend = nk < nissue ? nk : -1;
    ++ks;
  };
  auto next_buf = [&]() { cb = cb + 1 == RING ? 0 : cb + 1; };

  f32x4 accs[NSQ > 0 ? NSQ : 1][4];
#pragma unroll
  for (int t = 0; t < (NSQ > 0 ? NSQ : 1); ++t)
#pragma unroll
    for (int m = 0; m < 4; ++m) accs[t][m] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int ctot = a.E1 + a.E3;
  const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc(a.y, 0, WY ? a.y_bytes : 0u, 0x00020000);
  const int n_img = np * 2 + img;
  const int ox = ox0 + j;
  const bool col_ok = ox < a.W && n_img < a.N;

  if (a.dbg == 55) { vm_wait<0>(); return; }
  i32x4 an[4];      // first four fragments of the stage about to be consumed
  sync();           // stage 0
  refill();
#pragma unroll
  for (int t = 0; t < 4; ++t) an[t] = lda(0, t);

  // Block epilogue: bias + ReLU + float16 rounding of the 64 couts (concat channels cc0 .. cc0+64) of this wave's
  // 64 pixels; optional store; chain into the next squeeze (consumes one ring stage: Ws2 rows cc0 .. cc0+64).
  auto finish_block = [&](f32x4 (&acc)[4][4], int cc0) {
    i32x4 bf[4][2];
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      const f32x4 bias0 = *reinterpret_cast<const f32x4*>(bl + cc0 + p * 32 + g * 8);
      const f32x4 bias1 = *reinterpret_cast<const f32x4*>(bl + cc0 + p * 32 + g * 8 + 4);
#pragma unroll
      for (int m = 0; m < 4; ++m) bf[m][p] = pack8(relu4(acc[m][2 * p] + bias0), relu4(acc[m][2 * p + 1] + bias1));
    }
    if (WY) {
#pragma unroll
      for (int m = 0; m < 4; ++m) {
        const int oy = oy0 + r0 + m;
        const bool ok = col_ok && oy < a.H;
#pragma unroll
        for (int p = 0; p < 2; ++p) {
          const unsigned off = ok ? (unsigned)((((n_img * a.H + oy) * a.W + ox) * ctot + cc0 + p * 32 + g * 8) * 2) : 0xfffffff0u;
          __builtin_amdgcn_raw_buffer_store_b128(bf[m][p], ry, off, 0, 0);   // out of range = dropped
        }
      }
      ep_ks = ks;
    }
    if constexpr (NSQ > 0) {
      // chain stage: fragments f = u * NSQ + t (u = chunk of the pair, t = squeeze tile), an = fragments 0..3
      constexpr int NF = 2 * NSQ;
      i32x4 fr[NF];
#pragma unroll
      for (int f = 0; f < 4; ++f) fr[f] = an[f];
#pragma unroll
      for (int f = 4; f < NF; ++f) fr[f] = lda(cb, f);
      constexpr int SPLIT = NF > 8 ? 8 : (NF > 4 ? 4 : 0);   // the last group of MFMAs runs behind the next sync
#pragma unroll
      for (int f = 0; f < SPLIT; ++f) {
        const int u = f / NSQ, t = f - u * NSQ;
#pragma unroll
        for (int m = 0; m < 4; ++m) mma16<T>(accs[t][m], fr[f], bf[m][u]);
      }
      sync();
      const int nb = cb + 1 == RING ? 0 : cb + 1;
#pragma unroll
      for (int t = 0; t < 4; ++t) an[t] = lda(nb, t);
#pragma unroll
      for (int f = SPLIT; f < NF; ++f) {
        const int u = f / NSQ, t = f - u * NSQ;
#pragma unroll
        for (int m = 0; m < 4; ++m) mma16<T>(accs[t][m], fr[f], bf[m][u]);
        if (f - SPLIT == wave) refill();
      }
      refill();
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
      f32x4 acc[4][4];
#pragma unroll
      for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int t = 0; t < 4; ++t) acc[m][t] = f32x4{0.f, 0.f, 0.f, 0.f};
      i32x4 ac[4];
#pragma unroll
      for (int t = 0; t < 4; ++t) ac[t] = an[t];
#pragma unroll
      for (int c = 0; c < NCH; ++c) {
        i32x4 nx[4];
        if (c + 1 < NCH) {
#pragma unroll
          for (int t = 0; t < 4; ++t) nx[t] = lda(cb, (c + 1) * 4 + t);
        } else {
          sync();
          const int nb = cb + 1 == RING ? 0 : cb + 1;
#pragma unroll
          for (int t = 0; t < 4; ++t) nx[t] = lda(nb, t);
        }
#pragma unroll
        for (int m = 0; m < 4; ++m) {
#pragma unroll
          for (int t = 0; t < 4; ++t) mma16<T>(acc[m][t], ac[t], b1f[c][m]);
          if (c + 1 == NCH && m == wave) refill();
        }
#pragma unroll
        for (int t = 0; t < 4; ++t) ac[t] = nx[t];
      }
#pragma unroll
      for (int t = 0; t < 4; ++t) an[t] = ac[t];
      next_buf();
      finish_block(acc, blk * 64);
    }
  }

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
    f32x4 acc[4][4];
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
      for (int t = 0; t < 4; ++t) acc[m][t] = f32x4{0.f, 0.f, 0.f, 0.f};
    i32x4 ac[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) ac[t] = an[t];
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
#pragma unroll
      for (int dy = 0; dy < 3; ++dy) {
        if (dy == 0) {
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
          i32x4 nx[4];
          if (dx < 2) {
#pragma unroll
            for (int t = 0; t < 4; ++t) nx[t] = lda(cb, (dx + 1) * 4 + t);
          } else {
            sync();
            const int nb = cb + 1 == RING ? 0 : cb + 1;
#pragma unroll
            for (int t = 0; t < 4; ++t) nx[t] = lda(nb, t);
          }
#pragma unroll
          for (int m = 0; m < 4; ++m) {
#pragma unroll
            for (int t = 0; t < 4; ++t) mma16<T>(acc[m][t], ac[t], B[m + dy][dx]);
            if (dx == 2 && m == wave) refill();
          }
#pragma unroll
          for (int t = 0; t < 4; ++t) ac[t] = nx[t];
        }
        next_buf();
      }
    }
#pragma unroll
    for (int t = 0; t < 4; ++t) an[t] = ac[t];
    finish_block(acc, a.E1 + blk * 64);
  }
  vm_wait<0>();   // the dummy stages have landed (nobody reads them) before this wave's LDS can be handed on

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
        for (int t = 0; t < NSQ; ++t) {
          const f32x4 bias = *reinterpret_cast<const f32x4*>(bs + g * 4 * NSQ + t * 4);
          store4<T>(dst + t * 4, relu4(accs[t][m] + bias));
        }
      }
    }
  }
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

template <int NCH, int NSQ, bool WY>
int launch_chain(const ChainArgs& a, hipStream_t st) {
  const size_t lds = (size_t)RING * STAGE_B + (size_t)NIMG * NCH * CCHUNK + (size_t)(a.E1 + a.E3 + a.S2) * 4;
  static bool attr_done = false;
  if (!attr_done) {
    SQDET_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&fire_chain<NCH, NSQ, WY>),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    attr_done = true;
  }
  const int wgs = ((a.N + 1) / 2) * a.tiles_x * a.tiles_y;
  hipLaunchKernelGGL((fire_chain<NCH, NSQ, WY>), dim3((unsigned)((wgs + 7) / 8 * 8)), dim3(256), lds, st, a);
  SQDET_CHECK_HIP(hipGetLastError());
  return SQDET_OK;
}

template <int NCH, bool WY>
int dispatch_chain_nsq(const ChainArgs& a, int nsq, hipStream_t st) {
  switch (nsq) {
    case 0: if (WY) return launch_chain<NCH, 0, true>(a, st); break;
    case 3: return launch_chain<NCH, 3, WY>(a, st);
    case 4: return launch_chain<NCH, 4, WY>(a, st);
    case 6: return launch_chain<NCH, 6, WY>(a, st);
    default: break;
  }
  set_error("fire_chain: unsupported next-squeeze width");
  return SQDET_EUNSUPPORTED;
}

}  // namespace

bool fire_chain_eligible(int s, int e1, int e3, int s2, int dtype) {
  if (dtype != SQDET_F16 || conv_algo() != 0) return false;
  if (s <= 0 || s % 8 != 0 || s > 96) return false;
  if (e1 <= 0 || e3 <= 0 || e1 % 64 != 0 || e3 % 64 != 0) return false;
  if (!(s2 == 0 || s2 == 48 || s2 == 64 || s2 == 96)) return false;
  const ChainGeom g = chain_geom(s, e1, e3, s2);
  if (g.nch < 2 || g.nch > 3) return false;
  const size_t lds = (size_t)RING * STAGE_B + (size_t)NIMG * g.nch * CCHUNK + (size_t)(e1 + e3 + s2) * 4;
  return lds <= 160 * 1024;
}

}  // namespace sqdet

using namespace sqdet;

extern "C" size_t sqdet_fire_chain_stream_bytes(int s1x1, int e1x1, int e3x3, int next_s1x1, int dtype) {
  if (!fire_chain_eligible(s1x1, e1x1, e3x3, next_s1x1, dtype)) return 0;
  return (size_t)(chain_geom(s1x1, e1x1, e3x3, next_s1x1).nstages + 3) * STAGE_B;   // + the dummy stages
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
  a.dbg = tune(TUNE_DBG);
  hipStream_t st = as_stream(stream);
  if (g.nch == 2) return y ? dispatch_chain_nsq<2, true>(a, g.nsq, st) : dispatch_chain_nsq<2, false>(a, g.nsq, st);
  return y ? dispatch_chain_nsq<3, true>(a, g.nsq, st) : dispatch_chain_nsq<3, false>(a, g.nsq, st);
}
