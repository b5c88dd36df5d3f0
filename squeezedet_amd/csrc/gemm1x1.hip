// 1x1 convolution (any stride) with deep K as a workgroup-level GEMM tile (the squeeze / expand1x1 layers of SqueezeDet+,
// every 1x1 of ResNet50's bottleneck blocks incl. the residual-accumulate ones; reference src/nn_skeleton.py:471-563 via
// nets/squeezeDetPlus.py:81-106 and nets/resnet50_convDet.py:134-169).
//
// The generic conv_direct lets every wave fetch BOTH operands itself (8 KB per 16 MFMAs: the four SIMDs ask the CU's
// 64 B/clk L1 for about twice what it delivers); conv1x1_stream keeps the weights in registers and only fits K <= 4
// chunks.  Here a 256-thread workgroup owns 16*MB consecutive NHWC pixels (a 1x1 conv has no halo: "pixel" = row of the
// [P, Cin] activation matrix) and up to 4*NTW cout tiles:
//   * the activations of the tile are staged into LDS in stages of SC 64-byte K-chunks, laid out [chunk][pixel][4 x 16 B]
//     with the 16-byte slot of lane group g stored at g ^ ((pixel>>1)&3) (conv3x3.hip's bank-conflict-free layout), so
//     each activation byte crosses L2 -> CU once per workgroup and is read from LDS by the four waves;
//   * a wave owns MBW pixel blocks x NTW cout tiles (MBW*NTW accumulators) -- one 1-KiB weight fragment from L1 feeds MBW
//     MFMAs, one LDS B fragment NTW of them: (NTW + MBW) KiB of operands per MBW*NTW MFMAs.  The four waves are laid out
//     WR along the pixel blocks x 4/WR along the cout slices: WR = 1 when the conv has >= 4 slices (a slice = NTW cout
//     tiles), 2 or 4 when it has fewer (ResNet50's 256 -> 64 reductions have ONE: without the pixel split three of the
//     four waves only helped staging);
//   * weight fragments come straight from global one K-chunk ahead (register double buffer), across the staging barriers;
//   * epilogue: bias (+ the residual already in y: ResNet's branch2c) + ReLU, 8*NTW contiguous bytes per lane.
// Accumulation order = ascending K chunks, as in every other conv kernel here (bitwise-equal to conv_direct).
#include "conv_common.h"

namespace sqdet {
namespace {

struct G1Args {
  ConvArgs c;
  int nt_pack;       // tiles per packed cout group
  int slices;        // wave slices = ngroups * (nt_pack / NTW)
  int grid_y;        // workgroups per pixel tile = ceil(slices / (4 / WR))
  int ptiles;        // pixel tiles
  int pieces;        // 16-byte pieces per pixel = Cin * sizeof(T) / 16
  int stage_chunks;  // K-chunks resident in LDS at a time
};

template <typename T, int MBW, int NTW, int WR>
__global__ __launch_bounds__(256) void conv1x1_tile(G1Args a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  constexpr int MB = MBW * WR;           // pixel blocks per tile
  constexpr int WS = 4 / WR;             // cout slices per workgroup
  constexpr int TP = 16 * MB;            // pixels per tile
  constexpr int CH = TP * 64;            // bytes of one K-chunk of the tile
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int j = lane & 15, g = lane >> 4;
  // XCD-aware order: workgroup L runs on XCD L % 8; every XCD gets a contiguous band of pixel tiles, and the grid_y
  // workgroups that share a pixel tile (different couts) are neighbours on the same XCD (the tile is fetched into ONE L2)
  const int per_xcd = (a.ptiles + 7) / 8;
  const int idx = (int)(blockIdx.x >> 3);
  const int tl = idx / a.grid_y, ysl = idx - tl * a.grid_y;
  const int tile = (int)(blockIdx.x & 7) * per_xcd + tl;
  if (tl >= per_xcd || tile >= a.ptiles) return;
  const int p0 = tile * TP;

  const int wr = wave % WR;              // this wave's pixel blocks: wr*MBW ..
  const int slice = ysl * WS + wave / WR;
  const bool active = slice < a.slices;
  const int spg = a.nt_pack / NTW;       // slices per packed group
  const int group = active ? slice / spg : 0;
  const int n0 = active ? (slice - group * spg) * NTW : 0;

  f32x4 acc[MBW][NTW];
#pragma unroll
  for (int m = 0; m < MBW; ++m)
#pragma unroll
    for (int t = 0; t < NTW; ++t) acc[m][t] = f32x4{0.f, 0.f, 0.f, 0.f};

  const i32x4* wbase = reinterpret_cast<const i32x4*>(a.c.wp) + ((size_t)group * a.c.steps * a.nt_pack + n0) * 64 + lane;
  auto wptr = [&](int c) { return wbase + (size_t)c * a.nt_pack * 64; };
  i32x4 af[NTW], afn[NTW];
  if (active) {
#pragma unroll
    for (int t = 0; t < NTW; ++t) af[t] = wptr(0)[t * 64];
  }

  // the input may be a channel slice [x_coffset, x_coffset + Cin) of rows x_cstride channels wide (backward-data of a fire
  // module reads the expand1x1 half of dY)
  const unsigned char* x = reinterpret_cast<const unsigned char*>(a.c.x) + (size_t)a.c.x_coffset * sizeof(T);
  const int row_bytes = a.c.x_cstride * (int)sizeof(T);
  const int nchunk = a.c.nchunk;
  for (int c0 = 0; c0 < nchunk; c0 += a.stage_chunks) {
    const int nload = nchunk - c0 < a.stage_chunks ? nchunk - c0 : a.stage_chunks;
    if (c0 > 0) __syncthreads();           // everyone is done reading the previous stage
    {
      // stage nload chunks of the tile: batches of 4 loads in flight per thread before the LDS stores
      const int ppc = nload * 4, total = TP * ppc;
      for (int base = threadIdx.x; base < total; base += 256 * 4) {
        i32x4 v[4];
        int off[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int id = base + u * 256;
          const int P = id / ppc, q = id - P * ppc;
          const int gq = c0 * 4 + q;
          v[u] = i32x4{0, 0, 0, 0};
          if (id < total && gq < a.pieces && p0 + P < a.c.P) {
            size_t sp = (size_t)(p0 + P);                 // source pixel of output pixel p0 + P
            if (a.c.stride != 1) {                        // strided 1x1 (ResNet50's res3a / res4a reductions): every stride-th pixel
              const int hw = a.c.Ho * a.c.Wo;
              const int n = (p0 + P) / hw, r = (p0 + P) - n * hw;
              const int oy = r / a.c.Wo, ox = r - oy * a.c.Wo;
              sp = ((size_t)n * a.c.H + (size_t)oy * a.c.stride) * a.c.W + (size_t)ox * a.c.stride;
            }
            v[u] = *reinterpret_cast<const i32x4*>(x + sp * row_bytes + gq * 16);
          }
          off[u] = id < total ? (q >> 2) * CH + P * 64 + (((q & 3) ^ ((P >> 1) & 3)) << 4) : -1;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
          if (off[u] >= 0) *reinterpret_cast<i32x4*>(lds + off[u]) = v[u];
      }
    }
    __syncthreads();
    if (!active) continue;
#pragma unroll 1
    for (int cl = 0; cl < nload; ++cl) {
      const int c = c0 + cl;
      if (c + 1 < nchunk) {
        const i32x4* wp = wptr(c + 1);
#pragma unroll
        for (int t = 0; t < NTW; ++t) afn[t] = wp[t * 64];
      }
      // pixel (wr*MBW + m)*16 + j: (P>>1)&3 == (j>>1)&3
      const unsigned char* lc = lds + cl * CH + (wr * MBW * 16 + j) * 64 + ((g ^ ((j >> 1) & 3)) << 4);
      i32x4 bf[MBW];
#pragma unroll
      for (int m = 0; m < MBW; ++m) bf[m] = *reinterpret_cast<const i32x4*>(lc + m * 16 * 64);
#pragma unroll
      for (int m = 0; m < MBW; ++m)
#pragma unroll
        for (int t = 0; t < NTW; ++t) mma16<T>(acc[m][t], af[t], bf[m]);
      if (c + 1 < nchunk) {
#pragma unroll
        for (int t = 0; t < NTW; ++t) af[t] = afn[t];
      }
    }
  }
  if (!active) return;

  // epilogue: lane = pixel (block m, column j); couts group*16*NTp + g*4*NTp + n0*4 .. : 4*NTW consecutive
  T* y = reinterpret_cast<T*>(a.c.y);
  const int cb = group * 16 * a.nt_pack + g * 4 * a.nt_pack + n0 * 4;
  f32x4 bias[NTW];
  int nt_valid = 0;   // Cout is a multiple of 4: whole 4-cout pieces beyond Cout are skipped
#pragma unroll
  for (int t = 0; t < NTW; ++t) {
    const bool ok = cb + t * 4 < a.c.Cout;
    bias[t] = ok && a.c.bias ? *reinterpret_cast<const f32x4*>(a.c.bias + cb + t * 4) : f32x4{0.f, 0.f, 0.f, 0.f};
    nt_valid += ok ? 1 : 0;
  }
#pragma unroll
  for (int m = 0; m < MBW; ++m) {
    const int p = p0 + (wr * MBW + m) * 16 + j;
    if (p >= a.c.P) continue;
    T* dst = y + (size_t)p * a.c.y_cstride + a.c.y_coffset + cb;
    f32x4 v[NTW];
#pragma unroll
    for (int t = 0; t < NTW; ++t) {
      v[t] = acc[m][t] + bias[t];
      if (a.c.accum && t < nt_valid) {
        v[t][0] += (float)dst[t * 4 + 0]; v[t][1] += (float)dst[t * 4 + 1];
        v[t][2] += (float)dst[t * 4 + 2]; v[t][3] += (float)dst[t * 4 + 3];
      }
      if (a.c.relu) {
        v[t][0] = fmaxf(v[t][0], 0.f); v[t][1] = fmaxf(v[t][1], 0.f);
        v[t][2] = fmaxf(v[t][2], 0.f); v[t][3] = fmaxf(v[t][3], 0.f);
      }
      if (a.c.relu_of && t < nt_valid) {   // ReLU backward of the layer below (see ConvArgs)
        const T* r = reinterpret_cast<const T*>(a.c.relu_of) + (dst - y) + t * 4;
        v[t][0] = (float)r[0] > 0.f ? v[t][0] : 0.f; v[t][1] = (float)r[1] > 0.f ? v[t][1] : 0.f;
        v[t][2] = (float)r[2] > 0.f ? v[t][2] : 0.f; v[t][3] = (float)r[3] > 0.f ? v[t][3] : 0.f;
      }
    }
    store_couts<T, NTW>(dst, v, nt_valid);
  }
}

template <typename T, int MBW, int NTW, int WR>
void launch_g1(const G1Args& a, size_t lds, hipStream_t st) {
  const int per_xcd = (a.ptiles + 7) / 8;
  const dim3 grid((unsigned)(per_xcd * a.grid_y * 8));
  hipLaunchKernelGGL((conv1x1_tile<T, MBW, NTW, WR>), grid, dim3(256), lds, st, a);
}

template <typename T, int MBW, int WR>
bool dispatch_ntw(const G1Args& a, int ntw, size_t lds, hipStream_t st) {
  switch (ntw) {
    case 1: if constexpr (WR == 4) { launch_g1<T, MBW, 1, WR>(a, lds, st); return true; } return false;
    case 2: if constexpr (WR == 4) { launch_g1<T, MBW, 2, WR>(a, lds, st); return true; } return false;
    case 3: launch_g1<T, MBW, 3, WR>(a, lds, st); return true;
    case 4: launch_g1<T, MBW, 4, WR>(a, lds, st); return true;
    case 5: if constexpr (MBW <= 4) { launch_g1<T, MBW, 5, WR>(a, lds, st); return true; } return false;
    default: return false;
  }
}

template <typename T>
bool dispatch_g1(const G1Args& a, int mbw, int wr, int ntw, size_t lds, hipStream_t st) {
  if (wr == 1) return mbw == 8 ? dispatch_ntw<T, 8, 1>(a, ntw, lds, st) : dispatch_ntw<T, 4, 1>(a, ntw, lds, st);
  if (wr == 2) return mbw == 4 ? dispatch_ntw<T, 4, 2>(a, ntw, lds, st) : dispatch_ntw<T, 2, 2>(a, ntw, lds, st);
  return mbw == 4 ? dispatch_ntw<T, 4, 4>(a, ntw, lds, st) : dispatch_ntw<T, 2, 4>(a, ntw, lds, st);
}


// ---------------------------------------------------------------------------------------------------------------------------
// conv1x1_pipe (round 6): the same decomposition -- a wave owns MBW pixel blocks x NTW cout tiles, the four waves are laid out
// WR x 4/WR -- as a software PIPELINE.  conv1x1_tile alternates "fetch a 4-chunk stage, store it to LDS, barrier, compute": a
// workgroup's loads and MFMAs never overlap, and where a launch has only one workgroup per CU (the 13-15 k-pixel maps of ResNet50
// res4 / SqueezeDet+ fire9-11 at the benchmark batches) every stage is an exposed memory round trip -- 26-37 us for launches whose
// bytes move in 8 and whose MFMAs issue in 4 (profiles/r05_conv1x1_shapes_ab.txt).  Here
//   * the activation tile arrives by LDS-DMA (`buffer_load_dwordx4 ... lds`: no registers, no LDS stores) in a ring of NS one-chunk
//     slots, chunk c + NS - 1 requested while chunk c computes; out-of-range offsets (pixels past the end, channel padding of the last
//     chunk, chunks past the last one) land as zeros, so there is no branch in the loop;
//   * the weight fragments of chunk c + NS - 1 are requested at the same point into the register set chunk c - 1 has left: NS sets
//     named statically over an NS-times unrolled trip, no copies (chunks past the last one re-read the last chunk's weights and
//     multiply zeros: K is walked in whole trips);
//   * ONE counted wait + one bare barrier per chunk: `s_waitcnt vmcnt((NS - 2) * (Q + NTW))` leaves the two younger chunks' requests
//     in flight (vmcnt is one in-order counter over the DMA pieces and the fragment loads), the barrier publishes the other waves'
//     pieces of the slot and retires everybody's reads of the slot that is refilled behind it.  For that count to be EXACT every
//     memory instruction of the loop is inline asm: with compiler-issued fragment loads hipcc's wait-count pass -- which cannot see
//     the DMA pieces queued between them -- put `vmcnt((NS - 2) NTW)` in front of a chunk's MFMAs, i.e. waited for most of chunk
//     c + 1 as well (first version of this kernel: 720 cycles per 16-MFMA chunk where a launch has one workgroup per CU).  The wait
//     takes the chunk's fragment registers as in/out operands, so nothing that reads them can be scheduled above it;
//     tests/test_static_waits.py checks that no other instruction touches a fragment register inside the loop;
//   * the epilogue has no memory round trip: the bias is loaded before the loop, and the residual (`y +=`: ResNet50's branch2c,
//     the backward-data sums) and the ReLU mask (`relu_of`) of the wave's tile come in by LDS-DMA before the first chunk is
//     requested -- wave-private LDS, no registers, oldest in the queue (EPI form);
//   * <= 256 registers: two workgroups per CU where the launch has them, so one's ramp and epilogue run under the other's MFMAs.
// Accumulation order = ascending K chunks (+ 0 for the padding chunks): bitwise conv1x1_tile's and conv_direct's results.
struct P1Args {
  ConvArgs c;
  int nt_pack, slices, grid_y, ptiles, pieces;
  int nchunk_pad;      // K chunks rounded up to whole NS-chunk trips
  unsigned x_bytes;    // extent of x (buffer resource: offsets beyond it read zeros)
  unsigned y_bytes;    // extent of y / relu_of
  int res_off, mask_off;   // EPI: LDS byte offsets of the residual / mask areas (-1: not staged)
  int dbg;                 // -DSQDET_G1_EXP builds only: "dbg" 61 no activation traffic, 62 one weight chunk, 69 both, 63 K walked from a
                           // per-tile start (what bounds the K loop: G1_DBG=.. tools/g1_kslope.py); -DSQDET_G1_TIMELINE: 71 = cycles per
                           // step region (tools/g1_timeline.py)
};

// -DSQDET_G1_TIMELINE (experiments only, tools/g1_timeline.py): seven s_memrealtime (100 MHz) stamps per workgroup -- entry, first NS - 1
// chunks requested, chunk 0 landed, K loop done, queue drained, last store issued, stores retired -- written once at the very end
#ifdef SQDET_G1_TIMELINE
__device__ unsigned long long g_g1_tl[8192 * 8];
#define GTL(k) do { gtl[k] = wall_clock64(); } while (0)
#else
#define GTL(k) do {} while (0)
#endif

__device__ __forceinline__ void g1_dma16(unsigned voff, const i32x4& rsrc, unsigned lds_dst) {
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, 0 offen lds" ::"v"(voff), "s"(rsrc), "s"(lds_dst) : "memory", "m0");
}
// a weight fragment (wave-uniform base + lane offset + immediate), hidden from hipcc's wait-count pass like the DMA pieces (see the
// header): the register is valid behind g1_wait
template <int IMM>
__device__ __forceinline__ void g1_wload(i32x4& dst, unsigned voff, const unsigned char* sbase) {
  asm volatile("global_load_dwordx4 %0, %1, %2 offset:%3" : "=v"(dst) : "v"(voff), "s"(sbase), "n"(IMM) : "memory");
}
// `s_waitcnt vmcnt(N)` + barrier; the chunk's fragment registers pass THROUGH it (nothing that reads them moves above the wait)
template <int N, int NTW>
__device__ __forceinline__ void g1_wait(i32x4 (&w)[NTW]) {
  static_assert(N < 64 && NTW >= 2 && NTW <= 5, "");
  if constexpr (NTW == 2)
    asm volatile("s_waitcnt vmcnt(%2)\n\ts_barrier" : "+v"(w[0]), "+v"(w[1]) : "n"(N) : "memory");
  else if constexpr (NTW == 3)
    asm volatile("s_waitcnt vmcnt(%3)\n\ts_barrier" : "+v"(w[0]), "+v"(w[1]), "+v"(w[2]) : "n"(N) : "memory");
  else if constexpr (NTW == 4)
    asm volatile("s_waitcnt vmcnt(%4)\n\ts_barrier" : "+v"(w[0]), "+v"(w[1]), "+v"(w[2]), "+v"(w[3]) : "n"(N) : "memory");
  else
    asm volatile("s_waitcnt vmcnt(%5)\n\ts_barrier" : "+v"(w[0]), "+v"(w[1]), "+v"(w[2]), "+v"(w[3]), "+v"(w[4]) : "n"(N) : "memory");
}

template <typename T, int MBW, int NTW, int WR, int NS, bool EPI>
__global__ __launch_bounds__(256, 2) void conv1x1_pipe(P1Args a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  constexpr int MB = MBW * WR;           // pixel blocks per tile
  constexpr int WS = 4 / WR;             // cout slices per workgroup
  constexpr int TP = 16 * MB;            // pixels per tile
  constexpr int CH = TP * 64;            // bytes of one K-chunk of the tile = one ring slot
  constexpr int Q = MB / 4;              // DMA pieces (16 pixels x 64 B) per wave per chunk
  constexpr int RPM = sizeof(T) == 2 ? (NTW + 1) / 2 : NTW;   // EPI: 16-byte pieces per lane and pixel block (4 NTW couts)
  static_assert(MB % 4 == 0, "every wave fetches the same number of pieces (one wait count for all)");
  constexpr unsigned OOB = 0x80000000u;
  const unsigned lds_addr = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)lds;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int j = lane & 15, g = lane >> 4;
#ifdef SQDET_G1_TIMELINE
  unsigned long long gtl[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#endif
  GTL(0);
  // Every kernel argument the prologue needs is fetched HERE, in one batch behind one wait: left alone hipcc requests them where
  // they are first used -- seven dependent scalar-load round trips (0.2-0.5 us each on a busy chip) in front of the first request.
  asm volatile("" ::"s"(a.ptiles), "s"(a.grid_y), "s"(a.slices), "s"(a.nt_pack), "s"(a.pieces), "s"(a.nchunk_pad), "s"(a.x_bytes),
               "s"(a.y_bytes), "s"(a.res_off), "s"(a.mask_off), "s"(a.c.nchunk), "s"(a.c.steps), "s"(a.c.P), "s"(a.c.Cout),
               "s"(a.c.stride), "s"(a.c.x_cstride), "s"(a.c.x_coffset), "s"(a.c.y_cstride), "s"(a.c.y_coffset));
  asm volatile("" ::"s"(a.c.x), "s"(a.c.wp), "s"(a.c.bias), "s"(a.c.y), "s"(a.c.relu_of), "s"(a.c.relu), "s"(a.c.res));
#ifdef SQDET_G1_TIMELINE
  const unsigned long long clk0 = clock64();   // shader cycles: gtl[7] = cycles between here and the last stamp (effective clock)
#endif
  // XCD-aware order (as conv1x1_tile): workgroup L runs on XCD L % 8; an XCD owns a contiguous band of pixel tiles, the grid_y
  // workgroups of one pixel tile are neighbours on the same XCD
  const int per_xcd = (a.ptiles + 7) / 8;
  const int idx = (int)(blockIdx.x >> 3);
  const int tl = idx / a.grid_y, ysl = idx - tl * a.grid_y;
  const int tile = (int)(blockIdx.x & 7) * per_xcd + tl;
  if (tl >= per_xcd || tile >= a.ptiles) return;
  const int p0 = tile * TP;

  const int wr = wave % WR;
  const int slice_raw = ysl * WS + wave / WR;
  const bool active = slice_raw < a.slices;
  const int slice = active ? slice_raw : a.slices - 1;   // (a wave without a slice still fetches its pieces and keeps the barriers)
  const int spg = a.nt_pack / NTW;
  const int group = slice / spg;
  const int n0 = (slice - group * spg) * NTW;
  const int nchunk = a.c.nchunk;
  const int esz = (int)sizeof(T);
  const int cb = group * 16 * a.nt_pack + g * 4 * a.nt_pack + n0 * 4;   // this lane's 4 NTW consecutive couts

  // ---- epilogue operands first (oldest in the queue): the tile's residual / mask -> wave-private LDS, lane-linear
  const unsigned yrow = (unsigned)(a.c.y_cstride * esz);
  const unsigned ycol = (unsigned)((a.c.y_coffset + cb) * esz);
  const int pe0 = p0 + wr * MBW * 16 + j;                        // this lane's pixel of block m: pe0 + 16 m
  if constexpr (EPI) {
    const unsigned long long yaddr = (unsigned long long)(uintptr_t)(a.c.res ? a.c.res : a.c.y), maddr = (unsigned long long)(uintptr_t)a.c.relu_of;
    const i32x4 ry = {(int)(unsigned)yaddr, (int)(unsigned)((yaddr >> 32) & 0xffffu), (int)a.y_bytes, 0x00020000};
    const i32x4 rm = {(int)(unsigned)maddr, (int)(unsigned)((maddr >> 32) & 0xffffu), (int)a.y_bytes, 0x00020000};
#pragma unroll
    for (int m = 0; m < MBW; ++m) {
      const int p = pe0 + 16 * m;
      const unsigned yo = p < a.c.P && active ? (unsigned)p * yrow + ycol : OOB;
#pragma unroll
      for (int r = 0; r < RPM; ++r) {
        // (pieces past Cout read the neighbour's couts or zeros: never used)
        if (a.res_off >= 0) g1_dma16(yo + (unsigned)(r * 16), ry, lds_addr + (unsigned)(a.res_off + ((wave * MBW + m) * RPM + r) * 1024));
        if (a.mask_off >= 0) g1_dma16(yo + (unsigned)(r * 16), rm, lds_addr + (unsigned)(a.mask_off + ((wave * MBW + m) * RPM + r) * 1024));
      }
    }
  }
  // the bias as raw buffer loads (no bias / couts past Cout: out of range, zeros) -- nothing here waits for them
  f32x4 bias[NTW];
  int nt_valid = 0;   // Cout is a multiple of 4: whole 4-cout pieces beyond Cout are skipped
  {
    const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.c.bias), 0, a.c.bias ? a.c.Cout * 4 : 0, 0x00020000);
#pragma unroll
    for (int t = 0; t < NTW; ++t) {
      bias[t] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rb, (unsigned)(cb + t * 4) * 4u, 0, 0));
      nt_valid += cb + t * 4 < a.c.Cout ? 1 : 0;
    }
  }

  // ---- this lane's DMA sources: piece i = pixel block wave + 4 i of the tile; lane = (pixel l >> 2, LDS slot l & 3)
  const unsigned long long xaddr = (unsigned long long)(uintptr_t)a.c.x;
  const i32x4 rx = {(int)(unsigned)xaddr, (int)(unsigned)((xaddr >> 32) & 0xffffu), (int)a.x_bytes, 0x00020000};
  const int row_bytes = a.c.x_cstride * esz;
  const int piece = (lane & 3) ^ ((lane >> 3) & 3);              // the slot's XOR swizzle: (P >> 1) & 3 == (l >> 3) & 3
  const int nvalid = (a.pieces - piece + 3) >> 2;                // chunks c < nvalid hold this lane's piece
  unsigned xoff[Q];
#pragma unroll
  for (int i = 0; i < Q; ++i) {
    const int p = p0 + (wave + 4 * i) * 16 + (lane >> 2);
    unsigned sp = (unsigned)p;
    if (a.c.stride != 1) {                                       // strided 1x1: every stride-th pixel of the input map
      const int hw = a.c.Ho * a.c.Wo;
      const int n = p / hw, r = p - n * hw;
      const int oy = r / a.c.Wo, ox = r - oy * a.c.Wo;
      sp = (unsigned)((n * a.c.H + oy * a.c.stride) * a.c.W + ox * a.c.stride);
    }
    xoff[i] = p < a.c.P ? sp * (unsigned)row_bytes + (unsigned)(a.c.x_coffset * esz + piece * 16) : OOB;
  }
  // weight fragment t of chunk c: wave-uniform base + lane * 16 (+ 4096 for the fifth tile: the immediate ends at 4095)
  const unsigned char* wgrp = reinterpret_cast<const unsigned char*>(a.c.wp) + ((size_t)group * a.c.steps * a.nt_pack + n0) * 1024;
  const unsigned wlane = (unsigned)lane * 16u, wlane4 = wlane + 4096u;
  const unsigned wstep = (unsigned)a.nt_pack * 1024u;

  i32x4 wf[NS][NTW];
  // Requests walk the chunks in order, one chunk per step: running scalars instead of per-request address arithmetic (the K loop of a
  // lone workgroup is ISSUE-bound: a dozen extra scalar instructions per step cost it 300 cycles, tools/g1_kslope.py on an -DSQDET_G1_EXP
  // build).  wnext = the fragments of the next chunk to request (it stays on the last chunk: chunks past it multiply zeros),
  // xk / xkb = index and byte offset of the next chunk's pieces.
  const unsigned char* wnext = wgrp;
  int wk = 0, xk = 0;
  unsigned xkb = 0;
  // the tile's pieces of the next chunk into ring slot `slot` (static)
  auto request_x = [&](int slot) {
#ifdef SQDET_G1_EXP
    const bool off = a.dbg == 61 || a.dbg == 69;
#else
    constexpr bool off = false;
#endif
#pragma unroll
    for (int i = 0; i < Q; ++i)
      g1_dma16(xk < nvalid && !off ? xoff[i] + xkb : OOB, rx, lds_addr + (unsigned)(slot * CH + (wave + 4 * i) * 1024));
    xk += 1;
    xkb += 64u;
  };
  // the weight fragments of the next chunk into register set `set` (static)
  auto request_w = [&](int set) {
    const unsigned char* sb = wnext;
    g1_wload<0>(wf[set][0], wlane, sb);
    g1_wload<1024>(wf[set][1], wlane, sb);
    if constexpr (NTW > 2) g1_wload<2048>(wf[set][2], wlane, sb);
    if constexpr (NTW > 3) g1_wload<3072>(wf[set][3], wlane, sb);
    if constexpr (NTW > 4) g1_wload<0>(wf[set][4], wlane4, sb);
    wk += 1;
#ifdef SQDET_G1_EXP
    if (a.dbg == 62 || a.dbg == 69) return;
#endif
    wnext = wk < nchunk ? wnext + wstep : wnext;
  };
  // One step of the pipeline computes chunk c while chunk c + 1 is made available and chunk c + NS - 1 is requested:
  //     LDS reads of chunk c's LATER pixel blocks (m >= H)
  //     MFMAs of the chunk's first H pixel blocks  +  the fragment loads of chunk c + NS - 1 (into the set chunk c - 1 has left)
  //     wait + barrier for chunk c + 1             (covered by the MFMAs just issued)
  //     LDS reads of chunk c + 1's first H pixel blocks (their latency runs under the MFMAs that follow)
  //     MFMAs of the later pixel blocks            +  the DMA pieces of chunk c + NS - 1 (into the slot chunk c - 1 has left: every wave
  //                                                   has issued chunk c's first MFMAs, i.e. retired its reads of chunks < c)
  // i.e. every LDS read is issued half a chunk of MFMAs before its use (MBW + H fragments live), and a chunk's requests are in flight
  // for NS - 2 steps.  vmcnt retires in order, so the pieces cannot usefully run further ahead than the fragments: a wait for chunk
  // k + 1's fragments is a wait for every piece requested before them.  Depth is bought with register sets (3 / 4 / 6; 8 measured level stand-alone and - 2 % inside the steps).
  // (first version: wait, requests, LDS reads, MFMAs per chunk -- hipcc kept two B fragments live and exposed the LDS latency twice per
  //  chunk.)
  // The queue in front of the loop is the steady state's: per step [fragments x NTW][wait][pieces x Q], the wait for chunk k + 1 leaves
  // the NS - 3 younger chunks' requests and the fragments just requested in flight.
  constexpr int NWAIT = (NS - 3) * (Q + NTW) + NTW;
  constexpr int H = (MBW + 1) / 2;
#pragma unroll
  for (int u = 0; u < NS - 1; ++u) {
    request_w(u);
    request_x(u);
  }
  GTL(1);

  f32x4 acc[MBW][NTW];
#pragma unroll
  for (int m = 0; m < MBW; ++m)
#pragma unroll
    for (int t = 0; t < NTW; ++t) acc[m][t] = f32x4{0.f, 0.f, 0.f, 0.f};

  const unsigned char* lrd = lds + (wr * MBW * 16 + j) * 64 + ((g ^ ((j >> 1) & 3)) << 4);
  g1_wait<(NS - 2) * (Q + NTW), NTW>(wf[0]);            // chunk 0 has landed
  GTL(2);
  i32x4 bfa[H];                          // B fragments of the computing chunk's first H pixel blocks
#pragma unroll
  for (int m = 0; m < H; ++m) bfa[m] = *reinterpret_cast<const i32x4*>(lrd + m * 16 * 64);
#pragma unroll 1
  for (int c0 = 0; c0 < a.nchunk_pad; c0 += NS) {
#pragma unroll
    for (int u = 0; u < NS; ++u) {
      __builtin_amdgcn_sched_barrier(0);
      i32x4 bfb[MBW - H];
#pragma unroll
      for (int m = H; m < MBW; ++m) bfb[m - H] = *reinterpret_cast<const i32x4*>(lrd + u * CH + m * 16 * 64);
      request_w((u + NS - 1) % NS);
#pragma unroll
      for (int m = 0; m < H; ++m)
#pragma unroll
        for (int t = 0; t < NTW; ++t) mma16<T>(acc[m][t], wf[u][t], bfa[m]);
      __builtin_amdgcn_sched_barrier(0);
      g1_wait<NWAIT, NTW>(wf[(u + 1) % NS]);       // chunk c + 1: mine by the wait, everybody's by the barrier
#pragma unroll
      for (int m = 0; m < H; ++m) bfa[m] = *reinterpret_cast<const i32x4*>(lrd + ((u + 1) % NS) * CH + m * 16 * 64);
      __builtin_amdgcn_sched_barrier(0);
      request_x((u + NS - 1) % NS);
#pragma unroll
      for (int m = H; m < MBW; ++m)
#pragma unroll
        for (int t = 0; t < NTW; ++t) mma16<T>(acc[m][t], wf[u][t], bfb[m - H]);
    }
  }
  GTL(3);
  // (the trailing requests -- zeros into free slots, fragments nobody uses -- retire before the wave ends)
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  GTL(4);
  if (!active) return;

  // ---- epilogue: lane = pixel (block m, column j), 4 NTW consecutive couts; raw buffer stores (pixels past the end: dropped)
  const __amdgpu_buffer_rsrc_t ryb = __builtin_amdgcn_make_buffer_rsrc(a.c.y, 0, a.y_bytes, 0x00020000);
  constexpr unsigned DROP = 0xfffffff0u;
  // 16-byte stores of float16 tile pairs need whole tile pairs and 16-byte aligned row segments (wave-uniform)
  const bool wide = sizeof(T) == 2 && (NTW & 1) == 0 && nt_valid == NTW &&
                    ((reinterpret_cast<uintptr_t>(a.c.y) + (size_t)(a.c.y_coffset + group * 16 * a.nt_pack + n0 * 4) * esz) & 15) == 0 &&
                    (yrow & 15) == 0 && ((4 * a.nt_pack * esz) & 15) == 0;
  const unsigned char* lres = lds + (a.res_off >= 0 ? a.res_off : 0) + wave * MBW * RPM * 1024 + lane * 16;
  const unsigned char* lmask = lds + (a.mask_off >= 0 ? a.mask_off : 0) + wave * MBW * RPM * 1024 + lane * 16;
#pragma unroll
  for (int m = 0; m < MBW; ++m) {
    const int p = pe0 + 16 * m;
    const unsigned yo = p < a.c.P ? (unsigned)p * yrow + ycol : DROP;
    f32x4 v[NTW];
#pragma unroll
    for (int t = 0; t < NTW; ++t) {
      v[t] = acc[m][t] + bias[t];
      if constexpr (EPI) {
        if (a.res_off >= 0) {
          f32x4 r;
          if constexpr (sizeof(T) == 2) {
            const f16x8 h = *reinterpret_cast<const f16x8*>(lres + (m * RPM + (t >> 1)) * 1024);
            r = f32x4{(float)h[(t & 1) * 4 + 0], (float)h[(t & 1) * 4 + 1], (float)h[(t & 1) * 4 + 2], (float)h[(t & 1) * 4 + 3]};
          } else {
            r = *reinterpret_cast<const f32x4*>(lres + (m * RPM + t) * 1024);
          }
          v[t][0] += r[0]; v[t][1] += r[1]; v[t][2] += r[2]; v[t][3] += r[3];
        }
      }
      if (a.c.relu) {
        v[t][0] = fmaxf(v[t][0], 0.f); v[t][1] = fmaxf(v[t][1], 0.f);
        v[t][2] = fmaxf(v[t][2], 0.f); v[t][3] = fmaxf(v[t][3], 0.f);
      }
      if constexpr (EPI) {
        if (a.mask_off >= 0) {             // ReLU backward of the layer below (see ConvArgs)
          f32x4 r;
          if constexpr (sizeof(T) == 2) {
            const f16x8 h = *reinterpret_cast<const f16x8*>(lmask + (m * RPM + (t >> 1)) * 1024);
            r = f32x4{(float)h[(t & 1) * 4 + 0], (float)h[(t & 1) * 4 + 1], (float)h[(t & 1) * 4 + 2], (float)h[(t & 1) * 4 + 3]};
          } else {
            r = *reinterpret_cast<const f32x4*>(lmask + (m * RPM + t) * 1024);
          }
          v[t][0] = r[0] > 0.f ? v[t][0] : 0.f; v[t][1] = r[1] > 0.f ? v[t][1] : 0.f;
          v[t][2] = r[2] > 0.f ? v[t][2] : 0.f; v[t][3] = r[3] > 0.f ? v[t][3] : 0.f;
        }
      }
    }
    if (wide) {
#pragma unroll
      for (int t = 0; t + 1 < NTW; t += 2) {
        const f16x8 h = {(f16)v[t][0], (f16)v[t][1], (f16)v[t][2], (f16)v[t][3],
                         (f16)v[t + 1][0], (f16)v[t + 1][1], (f16)v[t + 1][2], (f16)v[t + 1][3]};
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(i32x4, h), ryb, yo != DROP ? yo + (unsigned)(t * 4 * esz) : DROP, 0, 0);
      }
    } else {
#pragma unroll
      for (int t = 0; t < NTW; ++t) {
        const unsigned off = (yo != DROP && t < nt_valid) ? yo + (unsigned)(t * 4 * esz) : DROP;
        if constexpr (sizeof(T) == 2) {
          const f16x4 h = {(f16)v[t][0], (f16)v[t][1], (f16)v[t][2], (f16)v[t][3]};
          __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(i32x2, h), ryb, off, 0, 0);
        } else {
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(i32x4, v[t]), ryb, off, 0, 0);
        }
      }
    }
  }
#ifdef SQDET_G1_TIMELINE
  GTL(5);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  GTL(6);
  gtl[7] = clock64() - clk0;
  if (a.dbg == 71) { gtl[1] = rg[0]; gtl[2] = rg[1]; gtl[3] = rg[2]; }
  if (threadIdx.x == 0 && blockIdx.x < 8192) {
#pragma unroll
    for (int k = 0; k < 8; ++k) g_g1_tl[(size_t)blockIdx.x * 8 + k] = gtl[k];
  }
#endif
}

template <typename T, int MBW, int NTW, int WR, int NS, bool EPI>
void launch_p1(P1Args& a, hipStream_t st) {
  const int per_xcd = (a.ptiles + 7) / 8;
  const dim3 grid((unsigned)(per_xcd * a.grid_y * 8));
  constexpr int RPM = sizeof(T) == 2 ? (NTW + 1) / 2 : NTW;
  size_t lds = (size_t)NS * 16 * MBW * WR * 64;
  const bool want_res = a.res_off >= 0, want_mask = a.mask_off >= 0;
  a.res_off = a.mask_off = -1;
  if (EPI && want_res) { a.res_off = (int)lds; lds += (size_t)4 * MBW * RPM * 1024; }
  if (EPI && want_mask) { a.mask_off = (int)lds; lds += (size_t)4 * MBW * RPM * 1024; }
  auto kern = &conv1x1_pipe<T, MBW, NTW, WR, NS, EPI>;
  if (lds > 64 * 1024) {
    static PerDevice once;
    (void)once.run([&] { return hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); });
  }
  hipLaunchKernelGGL(kern, grid, dim3(256), lds, st, a);
}

// (the six-set form exists for 64- and 128-pixel tiles with <= 4 pixel blocks and <= 4 cout tiles per wave: <= 256 registers,
//  <= 48 KiB of ring)
template <typename T, int MBW, int WR, int NS, bool EPI>
bool dispatch_p1_ntw(P1Args& a, int ntw, hipStream_t st) {
  constexpr bool deep = NS > 4;
  if constexpr (deep && (MBW > 4 || MBW * WR > 8)) return false;
  else {
    switch (ntw) {
      case 2: launch_p1<T, MBW, 2, WR, NS, EPI>(a, st); return true;
      case 3: launch_p1<T, MBW, 3, WR, NS, EPI>(a, st); return true;
      case 4: launch_p1<T, MBW, 4, WR, NS, EPI>(a, st); return true;
      case 5: if constexpr (MBW <= 4 && !deep) { launch_p1<T, MBW, 5, WR, NS, EPI>(a, st); return true; } return false;
      default: return false;
    }
  }
}

template <typename T, int NS, bool EPI>
bool dispatch_p1_geo(P1Args& a, int mbw, int wr, int ntw, hipStream_t st) {
  if (wr == 1) return mbw == 8 ? dispatch_p1_ntw<T, 8, 1, NS, EPI>(a, ntw, st) : dispatch_p1_ntw<T, 4, 1, NS, EPI>(a, ntw, st);
  if (wr == 2) return mbw == 4 ? dispatch_p1_ntw<T, 4, 2, NS, EPI>(a, ntw, st) : dispatch_p1_ntw<T, 2, 2, NS, EPI>(a, ntw, st);
  return mbw == 4 ? dispatch_p1_ntw<T, 4, 4, NS, EPI>(a, ntw, st) : dispatch_p1_ntw<T, 2, 4, NS, EPI>(a, ntw, st);
}

template <typename T, bool EPI>
bool dispatch_p1_ns(P1Args& a, int mbw, int wr, int ntw, int ns, hipStream_t st) {
  switch (ns) {
    case 3: return dispatch_p1_geo<T, 3, EPI>(a, mbw, wr, ntw, st);
    case 4: return dispatch_p1_geo<T, 4, EPI>(a, mbw, wr, ntw, st);
    case 6: return dispatch_p1_geo<T, 6, EPI>(a, mbw, wr, ntw, st);
    default: return false;
  }
}

template <typename T>
bool dispatch_p1(P1Args& a, int mbw, int wr, int ntw, int ns, bool epi, hipStream_t st) {
  return epi ? dispatch_p1_ns<T, true>(a, mbw, wr, ntw, ns, st) : dispatch_p1_ns<T, false>(a, mbw, wr, ntw, ns, st);
}

}  // namespace

// Eligibility + configuration.  *handled = false means "use the generic kernel".  Serves plain and residual-accumulate
// (accum) 1x1 / stride-1 convolutions whose input is a whole tensor (no channel slice) with Cin a multiple of the lane chunk.
int conv1x1_tile_launch(const ConvArgs& c, const ConvGeom& g, int dtype, hipStream_t st, bool* handled) {
  *handled = false;
  if (conv_algo() != 0) return SQDET_OK;
  if (c.k != 1 || c.stride < 1 || c.pt != 0 || c.pl != 0 || g.gather) return SQDET_OK;
  const int esz = dtype == SQDET_F16 ? 2 : 4;
  if ((c.Cin * esz) % 16 != 0 || (c.x_cstride * esz) % 16 != 0 || (c.x_coffset * esz) % 16 != 0) return SQDET_OK;
  if (g.nt < 3 && g.ngroups != 1) return SQDET_OK;          // 1- and 2-tile groups: only the one-slice layout (Cout <= 32)
  G1Args a;
  a.c = c;
  a.nt_pack = g.nt;
  a.pieces = c.Cin * esz / 16;
  const int ntw = g.nt <= 5 ? g.nt : g.nt / 2;            // 6-tile groups: two waves per group (accumulator budget)
  a.slices = g.ngroups * (g.nt / ntw);
  const int wr = a.slices >= 4 ? 1 : a.slices >= 2 ? 2 : 4;   // waves along the pixel blocks
  a.grid_y = (a.slices + 4 / wr - 1) / (4 / wr);
  // pixel blocks per wave: large tiles (128 pixels per wave column, 256 for the one-slice layout) unless that leaves the
  // chip under-filled (the 24x78 maps at batch 8 are 15 k pixels)
  int mbw = wr == 1 ? 8 : 4;
  auto wgs = [&](int m) { return (long)((c.P + 16 * m * wr - 1) / (16 * m * wr)) * a.grid_y; };
  if (wgs(mbw) < 768 || (wr == 1 && ntw == 5)) mbw /= 2;
  a.ptiles = (c.P + 16 * mbw * wr - 1) / (16 * mbw * wr);
  // The pipelined form (conv1x1_pipe) unless "dbg" 57 (conv1x1_tile for everything: A/B) or a tensor is beyond the 2 GiB of a buffer
  // offset.  Its layout -- measured inside the ResNet50 training step and the SqueezeDet+ serving step (profiles/r06_ab_g1_layout.txt),
  // where the inputs come from the Infinity Cache, not on cold stand-alone launches:
  //   * TWO cout tiles per wave wherever the packed group splits evenly (twice the workgroups, lighter waves: + 2-3 % per step);
  //   * 128-pixel tiles (8 pixel blocks per wave) in the four-slice layout while that leaves >= 200 workgroups (half the weight
  //     traffic per MFMA: + 3 % on SqueezeDet+), 64-pixel tiles otherwise and in the residual / mask forms beyond 64 KiB of LDS;
  //   * four register sets where K is a multiple of four chunks, else six, else three (eight were level stand-alone, - 2 % in-step).
  {
    const size_t xb = (size_t)c.N * c.H * c.W * c.x_cstride * esz, yb = (size_t)c.P * c.y_cstride * esz;
    if (xb < (1ull << 31) && yb < (1ull << 31) && tune(TUNE_DBG) != 57) {
      const bool epi = c.accum || c.relu_of;
      int pn = g.nt % 2 == 0 ? 2 : g.nt;
      if (tune(TUNE_G1_NTW) > 0 && g.nt % tune(TUNE_G1_NTW) == 0) pn = tune(TUNE_G1_NTW);
      const int pslices = g.ngroups * (g.nt / pn);
      int pw = pslices >= 4 ? 1 : pslices >= 2 ? 2 : 4;
      if (tune(TUNE_G1_WR) > 0) pw = tune(TUNE_G1_WR);
      const int pgy = (pslices + 4 / pw - 1) / (4 / pw);
      auto pwgs = [&](int m) { return (long)((c.P + 16 * m * pw - 1) / (16 * m * pw)) * pgy; };
      int pm = pw == 1 ? (pwgs(8) >= 200 && pn <= 4 ? 8 : 4) : (pwgs(4) >= 768 ? 4 : 2);
      if (tune(TUNE_G1_MBW) > 0) pm = tune(TUNE_G1_MBW);
      if (pw == 1 && pm == 2) pm = 4;
      if ((pw != 1 || pn > 4) && pm == 8) pm = 4;
      int ns = g.nchunk % 4 == 0 ? 4 : g.nchunk % 6 == 0 ? 6 : g.nchunk % 3 == 0 ? 3 : ((g.nchunk + 3) / 4 * 4 <= (g.nchunk + 2) / 3 * 3 ? 4 : 3);
      if (tune(TUNE_G1_NS) > 0) ns = tune(TUNE_G1_NS);
      if (ns != 3 && ns != 4 && ns != 6) ns = 4;
      if (ns == 6 && (pm > 4 || pm * pw > 8 || pn > 4)) ns = 3;                      // (the six-set form: <= 256 registers, <= 48 KiB of ring)
      const int rpm = esz == 2 ? (pn + 1) / 2 : pn;
      auto lds_need = [&](int m) { return (size_t)ns * 16 * m * pw * 64 + (size_t)((c.accum ? 1 : 0) + (c.relu_of ? 1 : 0)) * 4 * m * rpm * 1024; };
      if (epi && pm == 8 && lds_need(8) > 64 * 1024) pm = 4;
      if (lds_need(pm) <= 160 * 1024) {
        P1Args p;
        p.c = c; p.nt_pack = g.nt; p.slices = pslices; p.pieces = a.pieces;
        p.grid_y = pgy;
        p.ptiles = (c.P + 16 * pm * pw - 1) / (16 * pm * pw);
        p.x_bytes = (unsigned)xb; p.y_bytes = (unsigned)yb;
        p.dbg = tune(TUNE_DBG);
        p.res_off = c.accum ? 0 : -1; p.mask_off = c.relu_of ? 0 : -1;     // (launch_p1 turns the requests into LDS offsets)
        p.nchunk_pad = (g.nchunk + ns - 1) / ns * ns;
        const bool okp = dtype == SQDET_F16 ? dispatch_p1<f16>(p, pm, pw, pn, ns, epi, st) : dispatch_p1<float>(p, pm, pw, pn, ns, epi, st);
        if (okp) {
          SQDET_CHECK_HIP(hipGetLastError());
          *handled = true;
          return SQDET_OK;
        }
      }
    }
  }
  if (c.res) return SQDET_OK;            // (conv1x1_tile adds into y itself: the caller copies the residual first)
  a.stage_chunks = g.nchunk < 4 ? g.nchunk : 4;
  const size_t lds = (size_t)a.stage_chunks * 16 * mbw * wr * 64;   // <= 64 KiB
  const bool ok = dtype == SQDET_F16 ? dispatch_g1<f16>(a, mbw, wr, ntw, lds, st) : dispatch_g1<float>(a, mbw, wr, ntw, lds, st);
  if (!ok) return SQDET_OK;
  SQDET_CHECK_HIP(hipGetLastError());
  *handled = true;
  return SQDET_OK;
}

}  // namespace sqdet

#ifdef SQDET_G1_TIMELINE
extern "C" int sqdet_debug_g1_timeline(unsigned long long* host, int count) {
  return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(sqdet::g_g1_tl), sizeof(unsigned long long) * count);
}
#endif
