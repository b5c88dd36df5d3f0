// The ConvDet head (3x3 / SAME, 768 -> 72 couts = 5 cout tiles, K = 6912; reference src/nets/squeezeDet.py:76-79,
// src/nn_skeleton.py:471-563): conv3x3_tile's 8 x 16 output tile and LDS layout (conv3x3_tile.h) with K split over the
// four waves, as a PERSISTENT kernel, in a translation unit of its own.
//
//   * Every wave owns all 5 cout tiles (40 accumulators) and a quarter of K: the input is staged in stages of 4 K-chunks
//     (46 KB), wave w walks chunk 4*stage + w, 9 taps each -- one 5-KiB weight step per 40 MFMAs, every weight byte used
//     once per workgroup, so each step's fragments come from L2.  The 9 taps are unrolled over THREE register sets named
//     statically (step s uses set s % 3; 9 % 3 == 0, so the names line up across the stage loop and nothing is ever
//     copied), two steps are in flight, and the first two of the next stage are issued before this stage ends.
//     (requires nchunk % 4 == 0: every wave has a chunk in every stage)
//   * That is ~230 VGPRs beside the 160 accumulators: one workgroup per CU (one wave per SIMD).  This file is compiled
//     with hipcc's default register form -- accumulators in AGPRs, the 256 VGPRs for everything else; with
//     -amdgpu-mfma-vgpr-form=1 (the build's choice for every other MFMA kernel) the compiler parks 36 values in AGPRs and
//     moves them back and forth inside the tap loop: 143 v_accvgpr_* + 29 extra s_nop per 360 MFMAs (conv12 at batch 32:
//     70.3 us against 69.1, at batch 1: 27.5 against 25.3).
//   * Nobody else hides this workgroup's input staging, so the NEXT stage's 46 KB are fetched into registers (12 x 16 B
//     per thread, raw buffer loads: out-of-image offsets have bit 31 set and return the zero padding) while the current
//     stage computes -- two pieces per tap over taps 0..5, not one burst behind the barrier: the CU's L1 already moves
//     20 KB of weight fragments per tap (~60 % of its fill rate together with the input), and a 12-deep burst from all
//     four waves stalls the weight stream queued behind it (69 -> 66 us) -- and only the LDS stores sit between the two
//     barriers of a stage hand-over.
//   * tools/convdet_timing.py (s_memtime per segment): the tap loop is 63 % of a tile's time, the cold start (address
//     arithmetic, first 46 KB from HBM, first weights) 14 %, the stage hand-overs 12 %, the reduction 11 %; the SQ counters
//     (SQ_VALU_MFMA_BUSY_CYCLES = 16 cycles x every MFMA, against GRBM_GUI_ACTIVE) put the matrix pipe at 41 % of the
//     kernel, i.e. ~65-70 % inside the tap loop (s_memtime ticks slower than the shader clock).  A float16 workgroup
//     is therefore persistent (grid = one per CU, XCD-banded tile order): the last stage of a tile prefetches the first
//     stage of the workgroup's NEXT tile (and its first two weight steps), which stays in registers across the reduction
//     (same box, batch 32: 63.4 -> 61.4 us; one tile per workgroup, batch 1: 24.0 -> 24.6 us).  Register pressure decides
//     everything here: any value that lives across the tap loop beside the 228 VGPRs it needs is spilled to scratch, and
//     a spill in the prologue or the reduction costs 1-3 us per tile (measured: +11 us at batch 32 with 16 spilled
//     registers) -- the staging offsets are a per-stage scalar base + 12 per-thread constants + one validity bit mask per
//     tile, the bias is loaded per tile in the reduction, and hoisting out of the tile loop is blocked where it would
//     keep 64-bit addresses alive.  float32 (5x the MFMA time per tile, same staging) keeps one tile per workgroup.
//   * The 4 K-partials are summed by a reduce-scatter through LDS (wave o owns tile rows 2o, 2o+1; fixed order
//     ((w0+w1)+w2)+w3): results are bitwise those of any other split of the same K order.
// Tried and dropped (all bitwise-equal): wave-private double-buffered staging without barriers (64-byte line halves per
// wave: twice the L1 requests, 63.5 us against 62), cooperative double buffering with one barrier per stage and the LDS
// stores under taps 6..8 (62.0 us against 62.5 at batch 32, slower at batch 1 and 8), weights three steps ahead around a
// one-tap input burst with a single rolling B set (67.6 us).  Reading the same bytes as [stage][pixel][256 B] planes
// instead of NHWC rows: -1.2 us (not worth a private layout).  Without any input loads the kernel takes 55 us, with the
// input read from a 4-KB region 57 us: what remains is the CU's L1 traffic (20 KB of weight fragments per tap + 46 KB of
// input per stage = ~60 % of its fill rate) under an HBM-latency input stream.  Weights three steps ahead in the same three
// register sets (MFMAs fragment-major, a fragment's registers reloaded as soon as its 8 MFMAs are issued): 67.4 us against
// 61.2 on the same box.  float32: issuing the four K = 4 MFMAs of a
// fragment pair kk-outermost (independent accumulators back to back) is 8 % SLOWER (551 us against 510).
#include "conv3x3_tile.h"
#include "postproc.h"

namespace sqdet {

// -DSQDET_FIRE_TIMING (experiments only): per-wave s_memtime totals of the kernel's segments (tools/convdet_timing.py)
#ifdef SQDET_FIRE_TIMING
__device__ unsigned long long g_cd_timing[2048 * 8];
#define CT_MARK(k) do { const unsigned long long now_ = __builtin_amdgcn_s_memtime(); ct_acc[k] += now_ - ct_last; ct_last = now_; } while (0)
#else
#define CT_MARK(k) do {} while (0)
#endif

constexpr int CD_SLOT = 2 * 5 * 1024;                 // reduce-scatter: one (owner, source) slot = 2 rows x 5 tiles
constexpr int CD_LDS = 12 * CD_SLOT;                  // 122880 B (> the 46080 B of a staged input stage)
// SCORE form: behind the reduction slots, per wave, the float16-rounded preds channels [0, 40) of the wave's 2 x 16 pixels
// (class logits 0..26, confidences 27..35; 80 B per pixel) -- the exchange area of the score pass
constexpr int CD_SC_PIX = 80, CD_SC_WAVE = 2 * 16 * CD_SC_PIX, CD_SC_LDS = 4 * CD_SC_WAVE;

// PERS = false (float32: the persistent form does not fit the register file without spilling inside the tap loop): one tile
// per workgroup, nothing is prefetched across tiles
// SCORE (float16, Cout = 9 * (3 + 5) = 72: the reference's head): the epilogue also computes interpret_output's det_probs for the
// tile's 128 x 9 anchors from the float16-ROUNDED preds it stores -- score_from_logits (postproc.h), the very function the
// stand-alone score kernel and the filter kernel use, so the picks stay bit-exact -- and writes them as float32: the
// post-processing that follows is the 32 filter workgroups only (no second pass over preds, no 2100-workgroup launch on the
// side stream competing with the next forward's stem).  Per wave: its 2 rows x 16 pixels x 9 anchors = 288 scores, 4.5 per
// lane; a lane's anchor needs values of lane groups 0 and 1 of its pixel (couts 0..19 / 20..39), hence the LDS bounce.
template <typename T, bool PERS, bool SCORE = false>
__global__ __launch_bounds__(256) void convdet_kernel(TileArgs a, int ntiles, int per_xcd) {
  static_assert(!SCORE || sizeof(T) == 2, "the score epilogue is float16 only");
  constexpr int MT = 8, NTW = 5;
  constexpr int NSV = (HP * 16 + 255) / 256;   // 16-byte pieces per thread per stage = 12 (the last one partial)
  static_assert(NSV == 12, "two pieces per tap over six taps");
#ifdef SQDET_FIRE_TIMING
  unsigned long long ct_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, ct_last = __builtin_amdgcn_s_memtime();
#endif
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int j = lane & 15, g = lane >> 4;

  // XCD-aware tile order: workgroup b runs on XCD b % 8 (own L2 each); every XCD walks a contiguous band of tiles, so
  // the halos shared by neighbouring tiles are fetched into ONE L2.  Slot s of an XCD takes tiles s, s + nslot, ...
  const int xcd = (int)(blockIdx.x & 7), nslot = (int)(gridDim.x >> 3);
  int tl = (int)(blockIdx.x >> 3);
  auto tile_ok = [&](int t) { return t < per_xcd && xcd * per_xcd + t < ntiles; };
  if (!tile_ok(tl)) return;
  auto decode = [&](int t, int& n, int& oy0, int& ox0) {
    int b = xcd * per_xcd + t;
    const int tx = b % a.tiles_x; b /= a.tiles_x;
    const int ty = b % a.tiles_y;
    n = b / a.tiles_y; oy0 = ty * TROWS; ox0 = tx * TCOLS;
  };
  const int nstages = a.nchunk >> 2;

  f32x4 acc[MT][NTW];
  auto zero_acc = [&]() {
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int t = 0; t < NTW; ++t) acc[m][t] = f32x4{0.f, 0.f, 0.f, 0.f};
  };
  zero_acc();

  // ---- input staging: thread = (16-byte piece sq of the stage's 16, halo pixels sP0 + 16u)
  const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.c.x), 0, a.x_bytes, 0x00020000);
  const int sq = threadIdx.x & 15, sP0 = threadIdx.x >> 4;
  // byte offset of piece u (halo pixel P = sP0 + 16u -> halo row r, column cc) = a per-(tile, stage) scalar base + rel[u];
  // which of a thread's 12 pieces lie inside the image is one bit mask per tile (bit 31 of a buffer offset = zeros)
  const int row_bytes = a.pieces * 16;
  unsigned rel[NSV];
#pragma unroll
  for (int u = 0; u < NSV; ++u) {
    const int P = sP0 + 16 * u;
    const int r = P / (TCOLS + 2), cc = P - r * (TCOLS + 2);
    rel[u] = (unsigned)((r * a.c.W + cc) * row_bytes + sq * 16);
  }
  auto tile_base = [&](int n, int oy0, int ox0) { return (unsigned)(((n * a.c.H + oy0 - 1) * a.c.W + ox0 - 1) * row_bytes); };
  auto tile_mask = [&](int oy0, int ox0, bool valid) {
    unsigned m = 0;
    int p0 = sP0;
    asm volatile("" : "+v"(p0));   // r, cc are recomputed here, not kept in 24 registers across the stage loop
#pragma unroll
    for (int u = 0; u < NSV; ++u) {
      const int P = p0 + 16 * u;
      const int r = (int)(__umul24((unsigned)P, 57u) >> 10);   // P / 18 for P < 192
      const int cc = P - r * (TCOLS + 2);
      const int ok = (int)(u < NSV - 1 || P < HP) & (int)((unsigned)(oy0 - 1 + r) < (unsigned)a.c.H) &
                     (int)((unsigned)(ox0 - 1 + cc) < (unsigned)a.c.W);
      m |= (unsigned)ok << u;
    }
    return valid ? m : 0u;
  };
  auto piece_off = [&](int u, unsigned base, unsigned mask) { return (mask >> u) & 1u ? base + rel[u] : 0x80000000u; };
  // ((P + 16u) >> 1) & 3 == (P >> 1) & 3: the XOR slot is the same for all of a thread's pixels
  unsigned char* const sdst = lds + (sq >> 2) * CHUNK_BYTES + sP0 * 64 + (((sq & 3) ^ ((sP0 >> 1) & 3)) << 4);
  i32x4 sv[NSV];

  // ---- weights: fragment t of step (tap t9, chunk 4*stage + wave)
  const i32x4* wbase = reinterpret_cast<const i32x4*>(a.c.wp) + lane;
  auto wstep = [&](int stage, int t9) { return wbase + (size_t)(t9 * a.nchunk + stage * 4 + wave) * (NTW * 64); };
  i32x4 wf[3][NTW];

  int cn, coy0, cox0;   // the tile being computed
  decode(tl, cn, coy0, cox0);
  unsigned mask_cur = tile_mask(coy0, cox0, true);
  {
    const unsigned base = tile_base(cn, coy0, cox0);
#pragma unroll
    for (int u = 0; u < NSV; ++u) sv[u] = __builtin_amdgcn_raw_buffer_load_b128(rx, piece_off(u, base, mask_cur), 0, 0);
  }
#pragma unroll
  for (int p = 0; p < 2; ++p) {
    const i32x4* wp = wstep(0, p);
#pragma unroll
    for (int t = 0; t < NTW; ++t) wf[p][t] = wp[t * 64];
  }

  T* const y = reinterpret_cast<T*>(a.c.y);
  const int cb0 = (lane >> 4) * 4 * NTW;   // epilogue: lane = pixel (row m, col j); couts g*20 + t*4 .. +4

  bool first = true;
  int nn = 0, noy0 = 0, nox0 = 0;   // the workgroup's next tile
  bool has_next = PERS && tile_ok(tl + nslot);
  if (has_next) decode(tl + nslot, nn, noy0, nox0);
  unsigned mask_next = PERS ? tile_mask(noy0, nox0, has_next) : 0u;
  for (;;) {
#pragma unroll 1
    for (int stage = 0; stage < nstages; ++stage) {
      CT_MARK(0);
      if (!first) __syncthreads();   // every wave is done reading the previous stage (or the previous tile's reduction slots)
      first = false;
      CT_MARK(1);
#pragma unroll
      for (int u = 0; u < NSV; ++u)
        if (u < NSV - 1 || sP0 + 16 * u < HP) *reinterpret_cast<i32x4*>(sdst + u * 1024) = sv[u];   // sP0 <= 15
      CT_MARK(2);
      __syncthreads();
      CT_MARK(3);
      // what taps 0..5 prefetch: the next stage of this tile, or the first stage of the next tile (or nothing)
      const bool last = stage == nstages - 1;
      const unsigned pbase = last ? tile_base(nn, noy0, nox0) : tile_base(cn, coy0, cox0) + (unsigned)(stage + 1) * 256u;
      const unsigned pmask = last ? mask_next : mask_cur;
      const int nstage = last ? 0 : stage + 1;               // whose first two weight steps follow tap 8
      const unsigned char* lchunk = lds + wave * CHUNK_BYTES;
      // B fragments of tap t9 + 1 are read from LDS under the MFMAs of tap t9 (two statically named sets)
      i32x4 bfs[2][MT];
      auto bread = [&](int t9, i32x4 (&bf)[MT]) {
        const int dy = t9 / 3, dx = t9 - dy * 3;
        const int Pb = dy * (TCOLS + 2) + j + dx;
        const int h0 = Pb >> 1;
#pragma unroll
        for (int m = 0; m < MT; ++m)
          bf[m] = *reinterpret_cast<const i32x4*>(lchunk + (Pb + (TCOLS + 2) * m) * 64 + ((g ^ ((h0 + m) & 3)) << 4));
      };
      bread(0, bfs[0]);
      CT_MARK(4);
#pragma unroll
      for (int t9 = 0; t9 < 9; ++t9) {
        __builtin_amdgcn_sched_barrier(0);   // steps stay in order: no later tap's reads hoisted, no load sunk
        {
          const i32x4* wp = t9 + 2 < 9 ? wstep(stage, t9 + 2) : wstep(nstage, t9 + 2 - 9);
#pragma unroll
          for (int t = 0; t < NTW; ++t) wf[(t9 + 2) % 3][t] = wp[t * 64];
        }
        if (t9 < 6) {
#pragma unroll
          for (int u = 2 * t9; u < 2 * t9 + 2; ++u) sv[u] = __builtin_amdgcn_raw_buffer_load_b128(rx, piece_off(u, pbase, pmask), 0, 0);
        }
        if (t9 + 1 < 9) bread(t9 + 1, bfs[(t9 + 1) & 1]);
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
          for (int t = 0; t < NTW; ++t) mma16<T>(acc[m][t], wf[t9 % 3][t], bfs[t9 & 1][m]);
        // issue order inside the step: one memory instruction, then two or three MFMAs (the issues + their address
        // arithmetic cost ~200 cycles per step when they all sat ahead of the 40 MFMAs)
        const bool spread = t9 < 6;   // two more VMEM issues in this step
#pragma unroll
        for (int k = 0; k < NTW + (spread ? 2 : 0); ++k) {
          __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);   // VMEM read
          __builtin_amdgcn_sched_group_barrier(0x008, 3, 0);   // MFMA
        }
#pragma unroll
        for (int k = 0; k < (t9 + 1 < 9 ? MT : 0); ++k) {
          __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);   // DS read
          if (spread) __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
          else __builtin_amdgcn_sched_group_barrier(0x008, 3, 0);
        }
      }
      CT_MARK(5);
    }

    // ---- deterministic sum of the 4 K-partial accumulators, ((w0+w1)+w2)+w3, as a reduce-scatter through LDS: wave o
    // owns tile rows 2o, 2o+1; every wave writes the 30 accumulators it does not own (slot [owner][source][10 KiB],
    // 120 KiB), one barrier, and every wave sums its own 10 in ascending source order and stores them.  (Taking
    // turns on one 40-KiB buffer -- wave 0 writes, 1..3 add -- was 21 % of a workgroup's life.)
    constexpr int MO = MT / 4;   // rows per owner
    const int ox = cox0 + j;
    __syncthreads();  // all waves are done reading the input tile (the buffer is reused)
    int l16 = lane * 16;
    asm volatile("" : "+v"(l16));   // the slot addresses are recomputed per tile: hoisted out of the tile loop they were spilled
    int lz = lane;
    asm volatile("" : "+v"(lz));    // (SCORE: the same for the exchange-area addresses)
    auto slot_of = [&](int owner, int src) { return lds + (owner * 3 + (src < owner ? src : src - 1)) * CD_SLOT + l16; };
    auto scatter = [&](auto oc) {   // this wave's partials of owner oc's rows
      constexpr int o = decltype(oc)::value;
      if (wave == o) return;
      unsigned char* p = slot_of(o, wave);
#pragma unroll
      for (int mm = 0; mm < MO; ++mm)
#pragma unroll
        for (int t = 0; t < NTW; ++t) *reinterpret_cast<f32x4*>(p + (mm * NTW + t) * 1024) = acc[o * MO + mm][t];
    };
    scatter(std::integral_constant<int, 0>{});
    scatter(std::integral_constant<int, 1>{});
    scatter(std::integral_constant<int, 2>{});
    scatter(std::integral_constant<int, 3>{});
    // the tile after the next one: its mask is computed here, where the wave would only wait for the barrier
    const int tl_nn = tl + 2 * nslot;
    const bool has_nn = PERS && tile_ok(tl_nn);
    int n2 = 0, n2oy0 = 0, n2ox0 = 0;
    if (has_nn) decode(tl_nn, n2, n2oy0, n2ox0);
    const unsigned mask_nn = PERS ? tile_mask(n2oy0, n2ox0, has_nn) : 0u;
    f32x4 bias[NTW];   // loaded per tile: 20 registers that must not live across the tap loop
    int nt_valid = 0;  // whole 4-cout pieces beyond Cout are skipped
    int cb = cb0;
    asm volatile("" : "+v"(cb));   // nothing derived from it (bias / output pointers) is hoisted out of the tile loop
#pragma unroll
    for (int t = 0; t < NTW; ++t) {
      const bool ok = cb + t * 4 < a.c.Cout;
      bias[t] = ok ? *reinterpret_cast<const f32x4*>(a.c.bias + cb + t * 4) : f32x4{0.f, 0.f, 0.f, 0.f};
      nt_valid += ok ? 1 : 0;
    }
    __syncthreads();
    auto gather = [&](auto oc) {
      constexpr int o = decltype(oc)::value;
      if (wave != o || ox >= a.c.W) return;
#pragma unroll
      for (int mm = 0; mm < MO; ++mm) {
        const int oy = coy0 + o * MO + mm;
        if (oy >= a.c.H) break;
        T* dst = y + (((size_t)cn * a.c.H + oy) * a.c.W + ox) * a.c.y_cstride + a.c.y_coffset + cb;
        f32x4 v[NTW];
#pragma unroll
        for (int t = 0; t < NTW; ++t) {
          f32x4 s = o == 0 ? acc[o * MO + mm][t] : *reinterpret_cast<const f32x4*>(slot_of(o, 0) + (mm * NTW + t) * 1024);
#pragma unroll
          for (int src = 1; src < 4; ++src)
            s += src == o ? acc[o * MO + mm][t] : *reinterpret_cast<const f32x4*>(slot_of(o, src) + (mm * NTW + t) * 1024);
          v[t] = s + bias[t];
          if (a.c.relu) {
            v[t][0] = fmaxf(v[t][0], 0.f); v[t][1] = fmaxf(v[t][1], 0.f);
            v[t][2] = fmaxf(v[t][2], 0.f); v[t][3] = fmaxf(v[t][3], 0.f);
          }
        }
        store_couts<T, NTW>(dst, v, nt_valid);
        if constexpr (SCORE) {
          // (address from the per-tile laundered lane id: hoisted out of the tile loop it was spilled, and a scratch
          // reload in the reduction waits for the whole in-order vmcnt queue, i.e. for the next tile's prefetched input)
          const int j2 = lz & 15, g2 = lz >> 4;
          if (g2 < 2) {    // channels [20g, 20g + 20) of pixel (row mm, column j), rounded as stored
            unsigned char* sp = lds + CD_LDS + o * CD_SC_WAVE + (mm * 16 + j2) * CD_SC_PIX + g2 * 40;
#pragma unroll
            for (int t = 0; t < NTW; ++t) {
              const f16x4 h = {(f16)v[t][0], (f16)v[t][1], (f16)v[t][2], (f16)v[t][3]};
              *reinterpret_cast<f16x4*>(sp + t * 8) = h;
            }
          }
        }
      }
    };
    gather(std::integral_constant<int, 0>{});
    gather(std::integral_constant<int, 1>{});
    gather(std::integral_constant<int, 2>{});
    gather(std::integral_constant<int, 3>{});
    if constexpr (SCORE) {
      // this wave's LDS writes above are complete before its reads below (same wave: one in-order LDS queue; the wait makes
      // the data visible to the OTHER lanes' reads)
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      const unsigned char* sw = lds + CD_LDS + wave * CD_SC_WAVE;
      const int cells = a.c.H * a.c.W;
      int l2 = lz;                      // nothing of the 5 iterations' index arithmetic is hoisted out of the tile loop
      // (unrolled: five independent ~150-instruction dependency chains interleave; rolled, the pass was latency-bound and cost
      // the launch 5 us)
#pragma unroll
      for (int it = 0; it < 5; ++it) {
        const int idx0 = it * 64 + l2;                      // (row mm, pixel px, anchor k) = idx / 144, (idx % 144) / 9, idx % 9
        const int idx = idx0 < 288 ? idx0 : 287;            // computed unconditionally (no branch between the five chains),
        const int mm = idx >= 144 ? 1 : 0;                  // stored under the predicate
        const int r = idx - mm * 144;
        const int px = (int)(__umul24((unsigned)r, 57u) >> 9);   // r / 9 for r < 144
        const int k = r - px * 9;
        const int oy = coy0 + wave * MO + mm, oxx = cox0 + px;
        const f16* hp = reinterpret_cast<const f16*>(sw + (mm * 16 + px) * CD_SC_PIX);
        const float lg[3] = {(float)hp[3 * k], (float)hp[3 * k + 1], (float)hp[3 * k + 2]};
        int bc;
        const float sc = score_from_logits(lg, 3, (float)hp[27 + k], &bc);
        if (idx0 < 288 && oy < a.c.H && oxx < a.c.W) a.c.scores[((size_t)cn * cells + (size_t)oy * a.c.W + oxx) * 9 + k] = sc;
      }
    }
    CT_MARK(6);
    if (!PERS || !has_next) break;
    tl += nslot; cn = nn; coy0 = noy0; cox0 = nox0; mask_cur = mask_next;
    has_next = has_nn; nn = n2; noy0 = n2oy0; nox0 = n2ox0; mask_next = mask_nn;
    zero_acc();
  }
#ifdef SQDET_FIRE_TIMING
  if (lane == 0 && blockIdx.x * 4 + wave < 2048)
    for (int k = 0; k < 8; ++k) g_cd_timing[(blockIdx.x * 4 + wave) * 8 + k] = ct_acc[k];
#endif
}


// ---------------------------------------------------------------------------------------------------------------------------
// DMA form (round 4, float16, persistent): the same tile, K split, weight stream and reduction order, with the INPUT staging
// rebuilt around LDS-DMA:
//   * two stage buffers of 4 x 12 KiB (chunk pitch 192 pixels: whole 1-KiB DMA blocks); stage s computes from buffer s & 1 while
//     the 48 blocks of stage s + 1 -- `buffer_load_dwordx4 ... lds`, 16 pixels x 64 B each, wave w the pixel blocks 3w..3w+2 of all
//     four chunks (a pixel's 256 contiguous bytes are requested by one wave in consecutive instructions) -- land in the other one,
//     two blocks per tap over taps 0..5 as the register prefetch was.  No staging registers (-48 VGPRs at one wave per SIMD), no
//     ds_write pass, ONE barrier per stage instead of two.  The slot swizzle is applied on the source side (lane l requests the
//     piece whose slot it fills); out-of-image pixels and the 12 pitch pixels are out-of-range offsets = zeros.
//   * the DMA queue needs no wait of its own: the weight loads issued behind a stage's last DMA (tap 6: step 8's fragments) are
//     waited for by the compiler at tap 8, and `vmcnt` retires in order -- an explicit vmcnt(10) in front of the stage barrier
//     states it.
//   * the reduction runs in TWO rounds (row 0 then row 1 of every owner: 12 x 5 KiB = 60 KiB) inside buffer 1 + 12 KiB, so the next
//     tile's first stage can land in buffer 0 while the partials are summed.  Same order ((w0+w1)+w2)+w3: bitwise the same preds.
constexpr int CDD_CHUNK = 12288;                       // chunk pitch: 192 pixels x 64 B
constexpr int CDD_BUF = 4 * CDD_CHUNK;                 // 48 KiB
constexpr int CDD_SLOT = 5 * 1024;                     // reduction: one (owner, source) slot of ONE row = 5 tiles
constexpr int CDD_RED = CDD_BUF;                       // reduction area [48 KiB, 108 KiB)
constexpr int CDD_SC = CDD_RED + 12 * CDD_SLOT;        // score exchange area behind it
constexpr int CDD_LDS = CDD_SC;                        // (+ CD_SC_LDS in the SCORE form)

__device__ __forceinline__ void cd_dma16(unsigned voff, const i32x4& rsrc, unsigned lds_dst) {
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, 0 offen lds" ::"v"(voff), "s"(rsrc), "s"(lds_dst) : "memory", "m0");
}

template <bool SCORE>
__global__ __launch_bounds__(256) void convdet_dma_kernel(TileArgs a, int ntiles, int per_xcd) {
  using T = f16;
  constexpr int MT = 8, NTW = 5;
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  const unsigned lds_addr = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)lds;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int j = lane & 15, g = lane >> 4;

  const int xcd = (int)(blockIdx.x & 7), nslot = (int)(gridDim.x >> 3);
  int tl = (int)(blockIdx.x >> 3);
  auto tile_ok = [&](int t) { return t < per_xcd && xcd * per_xcd + t < ntiles; };
  if (!tile_ok(tl)) return;
  auto decode = [&](int t, int& n, int& oy0, int& ox0) {
    int b = xcd * per_xcd + t;
    const int tx = b % a.tiles_x; b /= a.tiles_x;
    const int ty = b % a.tiles_y;
    n = b / a.tiles_y; oy0 = ty * TROWS; ox0 = tx * TCOLS;
  };
  const int nstages = a.nchunk >> 2;                   // (even: checked by the launcher)

  f32x4 acc[MT][NTW];
  auto zero_acc = [&]() {
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int t = 0; t < NTW; ++t) acc[m][t] = f32x4{0.f, 0.f, 0.f, 0.f};
  };
  zero_acc();

  // ---- input staging by DMA: this wave's pixel blocks 3w + i (i = 0..2), lane = (pixel 16*(3w+i) + l/4, slot l & 3)
  const unsigned long long xaddr = (unsigned long long)(uintptr_t)a.c.x;
  const i32x4 rx = {(int)(unsigned)xaddr, (int)(unsigned)((xaddr >> 32) & 0xffffu), (int)a.x_bytes, 0x00020000};
  const int row_bytes = a.pieces * 16;
  unsigned rel[3];
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const int P = (3 * wave + i) * 16 + (lane >> 2);
    const int r = P / (TCOLS + 2), cc = P - r * (TCOLS + 2);
    const int piece = (lane & 3) ^ ((P >> 1) & 3);
    rel[i] = P < HP ? (unsigned)((r * a.c.W + cc) * row_bytes + piece * 16) : 0x80000000u;
  }
  auto tile_base = [&](int n, int oy0, int ox0) { return (unsigned)(((n * a.c.H + oy0 - 1) * a.c.W + ox0 - 1) * row_bytes); };
  auto tile_mask = [&](int oy0, int ox0, bool valid) {   // bit i: pixel of block 3w + i lies inside the image
    unsigned m = 0;
    int l4 = lane >> 2;
    asm volatile("" : "+v"(l4));
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      const int P = (3 * wave + i) * 16 + l4;
      const int r = (int)(__umul24((unsigned)P, 57u) >> 10);   // P / 18 for P < 192
      const int cc = P - r * (TCOLS + 2);
      const int ok = (int)(P < HP) & (int)((unsigned)(oy0 - 1 + r) < (unsigned)a.c.H) & (int)((unsigned)(ox0 - 1 + cc) < (unsigned)a.c.W);
      m |= (unsigned)ok << i;
    }
    return valid ? m : 0u;
  };
  // block (chunk c, pixel block i) of a stage whose first byte is `base` (tile base + stage * 256), into buffer `bufo`
  auto dma_block = [&](int c, int i, unsigned base, unsigned mask, unsigned bufo) {
    const unsigned off = (mask >> i) & 1u ? base + rel[i] + (unsigned)(c * 64) : 0x80000000u;
    cd_dma16(off, rx, lds_addr + bufo + (unsigned)(c * CDD_CHUNK + (3 * wave + i) * 1024));
  };

  const i32x4* wbase = reinterpret_cast<const i32x4*>(a.c.wp) + lane;
  auto wstep = [&](int stage, int t9) { return wbase + (size_t)(t9 * a.nchunk + stage * 4 + wave) * (NTW * 64); };
  i32x4 wf[3][NTW];

  int cn, coy0, cox0;
  decode(tl, cn, coy0, cox0);
  unsigned mask_cur = tile_mask(coy0, cox0, true);
  {
    const unsigned base = tile_base(cn, coy0, cox0);
#pragma unroll
    for (int k = 0; k < 12; ++k) dma_block(k & 3, k >> 2, base, mask_cur, 0u);
  }
#pragma unroll
  for (int p = 0; p < 2; ++p) {
    const i32x4* wp = wstep(0, p);
#pragma unroll
    for (int t = 0; t < NTW; ++t) wf[p][t] = wp[t * 64];
  }
  T* const y = reinterpret_cast<T*>(a.c.y);
  const int cb0 = (lane >> 4) * 4 * NTW;

  int nn = 0, noy0 = 0, nox0 = 0;
  bool has_next = tile_ok(tl + nslot);
  if (has_next) decode(tl + nslot, nn, noy0, nox0);
  unsigned mask_next = tile_mask(noy0, nox0, has_next);
  for (;;) {
    // The tile's first stage (and its first two weight steps) have landed -- as a wait the COMPILER sees (the builtin, not asm): it
    // also tells hipcc's wait-count pass that nothing of the previous tile's epilogue (bias loads, preds / score stores) is pending
    // when the stage loop is entered.  With an asm wait the loop header inherited those from the tile loop's back edge and the pass
    // put `s_waitcnt vmcnt(0)` behind every stage's barrier: each of a tile's six stages began by draining the weight stream (the
    // two steps requested at taps 7 and 8 of the stage before) -- the "stage hand-overs" of the file header.
    __builtin_amdgcn_s_waitcnt(0x0f70);                 // vmcnt(0), expcnt / lgkmcnt untouched
#pragma unroll 1
    for (int stage = 0; stage < nstages; ++stage) {
      // everybody's blocks of this stage have landed (see the file header); everybody is done reading the other buffer
      asm volatile("s_waitcnt vmcnt(10)\n\ts_barrier" ::: "memory");
      const bool last = stage == nstages - 1;
      const unsigned pbase = last ? tile_base(nn, noy0, nox0) : tile_base(cn, coy0, cox0) + (unsigned)(stage + 1) * 256u;
      const unsigned pmask = last ? mask_next : mask_cur;
      const unsigned pbuf = (unsigned)(((stage + 1) & 1) * CDD_BUF);
      const int nstage = last ? 0 : stage + 1;
      const unsigned char* lchunk = lds + (stage & 1) * CDD_BUF + wave * CDD_CHUNK;
      i32x4 bfs[2][MT];
      auto bread = [&](int t9, i32x4 (&bf)[MT]) {
        const int dy = t9 / 3, dx = t9 - dy * 3;
        const int Pb = dy * (TCOLS + 2) + j + dx;
        const int h0 = Pb >> 1;
#pragma unroll
        for (int m = 0; m < MT; ++m)
          bf[m] = *reinterpret_cast<const i32x4*>(lchunk + (Pb + (TCOLS + 2) * m) * 64 + ((g ^ ((h0 + m) & 3)) << 4));
      };
      bread(0, bfs[0]);
#pragma unroll
      for (int t9 = 0; t9 < 9; ++t9) {
        __builtin_amdgcn_sched_barrier(0);
        if (t9 < 6) {   // two DMA blocks per tap: chunks (2 t9) & 3, (2 t9 + 1) & 3 of pixel block t9 / 2 -- unconditional (no next
                        // tile: its mask is 0, every offset out of range, zeros land in the idle buffer): the tap loop stays one block
          dma_block((2 * t9) & 3, (2 * t9) >> 2, pbase, pmask, pbuf);
          dma_block((2 * t9 + 1) & 3, (2 * t9 + 1) >> 2, pbase, pmask, pbuf);
        }
        {
          const i32x4* wp = t9 + 2 < 9 ? wstep(stage, t9 + 2) : wstep(nstage, t9 + 2 - 9);
#pragma unroll
          for (int t = 0; t < NTW; ++t) wf[(t9 + 2) % 3][t] = wp[t * 64];
        }
        if (t9 + 1 < 9) bread(t9 + 1, bfs[(t9 + 1) & 1]);
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
          for (int t = 0; t < NTW; ++t) mma16<T>(acc[m][t], wf[t9 % 3][t], bfs[t9 & 1][m]);
#pragma unroll
        for (int k = 0; k < NTW; ++k) {
          __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);   // VMEM read
          __builtin_amdgcn_sched_group_barrier(0x008, 3, 0);   // MFMA
        }
#pragma unroll
        for (int k = 0; k < (t9 + 1 < 9 ? MT : 0); ++k) {
          __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);   // DS read
          __builtin_amdgcn_sched_group_barrier(0x008, 3, 0);
        }
      }
    }

    // ---- deterministic sum of the 4 K-partials, ((w0+w1)+w2)+w3, as a reduce-scatter through LDS in TWO rounds (row mm = 0, 1
    // of every owner): slot [owner][source] of 5 KiB, 60 KiB inside buffer 1 (+ 12 KiB): buffer 0 is taking the next tile's first stage
    constexpr int MO = MT / 4;
    const int ox = cox0 + j;
    int l16 = lane * 16;
    asm volatile("" : "+v"(l16));
    int lz = lane;
    asm volatile("" : "+v"(lz));
    auto slot_of = [&](int owner, int src) { return lds + CDD_RED + (owner * 3 + (src < owner ? src : src - 1)) * CDD_SLOT + l16; };
    const int tl_nn = tl + 2 * nslot;
    const bool has_nn = tile_ok(tl_nn);
    int n2 = 0, n2oy0 = 0, n2ox0 = 0;
    if (has_nn) decode(tl_nn, n2, n2oy0, n2ox0);
    const unsigned mask_nn = tile_mask(n2oy0, n2ox0, has_nn);
    f32x4 bias[NTW];
    int nt_valid = 0;
    int cb = cb0;
    asm volatile("" : "+v"(cb));
#pragma unroll
    for (int t = 0; t < NTW; ++t) {
      const bool ok = cb + t * 4 < a.c.Cout;
      bias[t] = ok ? *reinterpret_cast<const f32x4*>(a.c.bias + cb + t * 4) : f32x4{0.f, 0.f, 0.f, 0.f};
      nt_valid += ok ? 1 : 0;
    }
#pragma unroll
    for (int mm = 0; mm < MO; ++mm) {
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");   // all waves are done reading buffer 1 (round 0) / the previous round's slots
      auto scatter = [&](auto oc) {
        constexpr int o = decltype(oc)::value;
        if (wave == o) return;
        unsigned char* p = slot_of(o, wave);
#pragma unroll
        for (int t = 0; t < NTW; ++t) *reinterpret_cast<f32x4*>(p + t * 1024) = acc[o * MO + mm][t];
      };
      scatter(std::integral_constant<int, 0>{});
      scatter(std::integral_constant<int, 1>{});
      scatter(std::integral_constant<int, 2>{});
      scatter(std::integral_constant<int, 3>{});
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
      auto gather = [&](auto oc) {
        constexpr int o = decltype(oc)::value;
        if (wave != o || ox >= a.c.W) return;
        const int oy = coy0 + o * MO + mm;
        if (oy >= a.c.H) return;
        T* dst = y + (((size_t)cn * a.c.H + oy) * a.c.W + ox) * a.c.y_cstride + a.c.y_coffset + cb;
        f32x4 v[NTW];
#pragma unroll
        for (int t = 0; t < NTW; ++t) {
          f32x4 s_ = o == 0 ? acc[o * MO + mm][t] : *reinterpret_cast<const f32x4*>(slot_of(o, 0) + t * 1024);
#pragma unroll
          for (int src = 1; src < 4; ++src)
            s_ += src == o ? acc[o * MO + mm][t] : *reinterpret_cast<const f32x4*>(slot_of(o, src) + t * 1024);
          v[t] = s_ + bias[t];
          if (a.c.relu) {
            v[t][0] = fmaxf(v[t][0], 0.f); v[t][1] = fmaxf(v[t][1], 0.f);
            v[t][2] = fmaxf(v[t][2], 0.f); v[t][3] = fmaxf(v[t][3], 0.f);
          }
        }
        store_couts<T, NTW>(dst, v, nt_valid);
        if constexpr (SCORE) {
          const int j2 = lz & 15, g2 = lz >> 4;
          if (g2 < 2) {
            unsigned char* sp = lds + CDD_SC + o * CD_SC_WAVE + (mm * 16 + j2) * CD_SC_PIX + g2 * 40;
#pragma unroll
            for (int t = 0; t < NTW; ++t) {
              const f16x4 h = {(f16)v[t][0], (f16)v[t][1], (f16)v[t][2], (f16)v[t][3]};
              *reinterpret_cast<f16x4*>(sp + t * 8) = h;
            }
          }
        }
      };
      gather(std::integral_constant<int, 0>{});
      gather(std::integral_constant<int, 1>{});
      gather(std::integral_constant<int, 2>{});
      gather(std::integral_constant<int, 3>{});
    }
    if constexpr (SCORE) {
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      const unsigned char* sw = lds + CDD_SC + wave * CD_SC_WAVE;
      const int cells = a.c.H * a.c.W;
      int l2 = lz;
#pragma unroll
      for (int it = 0; it < 5; ++it) {
        const int idx0 = it * 64 + l2;
        const int idx = idx0 < 288 ? idx0 : 287;
        const int mm = idx >= 144 ? 1 : 0;
        const int r = idx - mm * 144;
        const int px = (int)(__umul24((unsigned)r, 57u) >> 9);
        const int k = r - px * 9;
        const int oy = coy0 + wave * MO + mm, oxx = cox0 + px;
        const f16* hp = reinterpret_cast<const f16*>(sw + (mm * 16 + px) * CD_SC_PIX);
        const float lg[3] = {(float)hp[3 * k], (float)hp[3 * k + 1], (float)hp[3 * k + 2]};
        int bc;
        const float sc = score_from_logits(lg, 3, (float)hp[27 + k], &bc);
        if (idx0 < 288 && oy < a.c.H && oxx < a.c.W) a.c.scores[((size_t)cn * cells + (size_t)oy * a.c.W + oxx) * 9 + k] = sc;
      }
    }
    if (!has_next) break;
    tl += nslot; cn = nn; coy0 = noy0; cox0 = nox0; mask_cur = mask_next;
    has_next = has_nn; nn = n2; noy0 = n2oy0; nox0 = n2ox0; mask_next = mask_nn;
    zero_acc();
  }
}

template <bool SCORE>
static void convdet_dma_launch(const TileArgs& a, hipStream_t st) {
  static PerDevice once;
  (void)once.run([] { return hipFuncSetAttribute(reinterpret_cast<const void*>(&convdet_dma_kernel<SCORE>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); });
  const int ntiles = a.c.N * a.tiles_x * a.tiles_y;
  const int per_xcd = (ntiles + 7) / 8;
  const int cus8 = cu_count() / 8;                            // persistent: one workgroup per CU
  const int slots = per_xcd < cus8 ? per_xcd : cus8;
  hipLaunchKernelGGL((convdet_dma_kernel<SCORE>), dim3((unsigned)(slots * 8)), dim3(256), CDD_LDS + (SCORE ? CD_SC_LDS : 0), st, a, ntiles, per_xcd);
}

template <typename T, bool PERS, bool SCORE = false>
static void convdet_launch(const TileArgs& a, hipStream_t st) {
  static PerDevice once;        // > 64 KiB of dynamic LDS has to be allowed once per kernel and device
  (void)once.run([] { return hipFuncSetAttribute(reinterpret_cast<const void*>(&convdet_kernel<T, PERS, SCORE>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); });
  const int ntiles = a.c.N * a.tiles_x * a.tiles_y;
  const int per_xcd = (ntiles + 7) / 8;
  const int cus8 = cu_count() / 8;
  const int slots = (!PERS || per_xcd < cus8) ? per_xcd : cus8;   // persistent: one workgroup per CU (32 CUs per XCD)
  hipLaunchKernelGGL((convdet_kernel<T, PERS, SCORE>), dim3((unsigned)(slots * 8)), dim3(256), CD_LDS + (SCORE ? CD_SC_LDS : 0), st, a, ntiles, per_xcd);
}

bool convdet_score_supported(int cout, int apg, int classes, int dtype) {
  return dtype == SQDET_F16 && apg == 9 && classes == 3 && cout == apg * (classes + 5);
}

int convdet_tile_launch(const TileArgs& a, int dtype, hipStream_t st) {
  if (a.c.scores) {
    if (!convdet_score_supported(a.c.Cout, a.c.score_apg, a.c.score_classes, dtype) || a.c.relu) {
      set_error("convdet: the score epilogue needs float16, 9 anchors x (3 classes + 5) = 72 couts, no ReLU");
      return SQDET_EUNSUPPORTED;
    }
    // the DMA-staged form is the default since round 4 ("dbg" 80: the register-prefetch form).  Stand-alone the two are level
    // (69.4 against 70.4 us on one box); inside the 32-image step the DMA form measures 0.4836 against 0.4935 ms (three alternating
    // runs; the chip also holds ~2 % more clock under it: `box` in the bench line)
    if (tune(TUNE_DBG) != 80 && (a.nchunk >> 2) % 2 == 0) convdet_dma_launch<true>(a, st);
    else convdet_launch<f16, true, true>(a, st);
    return SQDET_OK;
  }
  if (dtype == SQDET_F16 && tune(TUNE_DBG) != 80 && (a.nchunk >> 2) % 2 == 0) convdet_dma_launch<false>(a, st);
  else if (dtype == SQDET_F16) convdet_launch<f16, true>(a, st);
  else convdet_launch<float, false>(a, st);
  return SQDET_OK;
}

}  // namespace sqdet

#ifdef SQDET_FIRE_TIMING
extern "C" int sqdet_debug_convdet_timing(unsigned long long* host, int count) {
  return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(sqdet::g_cd_timing), sizeof(unsigned long long) * count);
}
#endif
