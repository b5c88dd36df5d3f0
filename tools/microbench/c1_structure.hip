// Microbenchmark: loop structure of a streaming 1x1 conv with K <= 32 and 64 couts per wave (conv1x1_stream<f16, 1, 4, MT, PERM> --
// fire2/3 expand1x1: 16 -> 64 channels on 935 k pixels, fire4/5 expand1x1: 32 -> 128 on 235 k), arithmetic included, results unchecked:
//   MT     16-pixel blocks per step (2, 4, 8)
//   NSETS  register sets of B fragments = prefetch distance + 1 (2, 3, 4): the loads of step i + NSETS - 1 are issued at the top of step i
//   WPS    waves per SIMD the kernel is compiled for (register budget 512 / WPS) -- the launch fills exactly that
//   RELU16 ReLU as v_pk_max_f16 behind the conversion (2 instead of 4 VALU per tile)
// Fresh memory every launch (regions rotate through 3 GiB).
//   hipcc --offload-arch=gfx950 -O3 -mllvm -amdgpu-mfma-vgpr-form=1 -o /tmp/c1_structure tools/microbench/c1_structure.hip && /tmp/c1_structure
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <type_traits>

typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef int i32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16;
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

struct Args {
  const void* x; void* y; const void* w; const float* bias;
  int P, Cin, Cout, ngroups, ntiles, nstreams;
  unsigned x_bytes, y_bytes;
};

__device__ __forceinline__ unsigned pkmax0(unsigned a) {
  unsigned r;
  asm("v_pk_max_f16 %0, %1, 0" : "=v"(r) : "v"(a));
  return r;
}

template <int MT, int NSETS, int WPS, bool RELU16>
__global__ __launch_bounds__(256, WPS) void k(Args a) {
  constexpr int NT = 4;
  const int lane = threadIdx.x & 63, j = lane & 15, g = lane >> 4;
  const int gw = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int group = gw % a.ngroups, stream = gw / a.ngroups;
  if (stream >= a.nstreams) return;
  i32x4 af[NT];
  f32x4 bias[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    af[t] = reinterpret_cast<const i32x4*>(a.w)[(group * NT + t) * 64 + lane];
    bias[t] = *reinterpret_cast<const f32x4*>(a.bias + group * 64 + 32 * (t >> 1) + 8 * g + 4 * (t & 1));
  }
  const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.x), 0, a.x_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc(a.y, 0, a.y_bytes, 0x00020000);
  constexpr unsigned OOB = 0xfffffff0u;
  const unsigned xrow = a.Cin * 2, yrow = a.Cout * 2;
  const unsigned kb = g * 8 < a.Cin ? g * 16 : OOB;
  const unsigned yb = group * 128 + g * 16;
  auto load_tile = [&](int tile, i32x4 (&bf)[MT]) {
#pragma unroll
    for (int m = 0; m < MT; ++m) {
      const int p = (tile * MT + m) * 16 + j;
      bf[m] = __builtin_amdgcn_raw_buffer_load_b128(rx, (p < a.P && kb != OOB) ? (unsigned)p * xrow + kb : OOB, 0, 0);
    }
  };
  auto dummy_stores = [&]() {
#pragma unroll
    for (int i = 0; i < MT * 2; ++i) __builtin_amdgcn_raw_buffer_store_b128(i32x4{0, 0, 0, 0}, ry, OOB, 0, 0);
  };
  auto step = [&](int tile, i32x4 (&cur)[MT], i32x4 (&nxt)[MT]) {
    load_tile(tile + (NSETS - 1) * a.nstreams, nxt);
    __builtin_amdgcn_sched_barrier(0);
    f32x4 acc[MT][NT];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int t = 0; t < NT; ++t)
        acc[m][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, af[t]), __builtin_bit_cast(f16x8, cur[m]), f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
#pragma unroll
    for (int m = 0; m < MT; ++m) {
      const int p = (tile * MT + m) * 16 + j;
      const unsigned po = p < a.P ? (unsigned)p * yrow + yb : OOB;
#pragma unroll
      for (int t = 0; t < NT; t += 2) {
        f32x4 v0 = acc[m][t] + bias[t], v1 = acc[m][t + 1] + bias[t + 1];
        i32x4 o;
        if (RELU16) {
          const f16x8 h = {(f16)v0[0], (f16)v0[1], (f16)v0[2], (f16)v0[3], (f16)v1[0], (f16)v1[1], (f16)v1[2], (f16)v1[3]};
          o = __builtin_bit_cast(i32x4, h);
#pragma unroll
          for (int e = 0; e < 4; ++e) o[e] = (int)pkmax0((unsigned)o[e]);
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e) { v0[e] = fmaxf(v0[e], 0.f); v1[e] = fmaxf(v1[e], 0.f); }
          const f16x8 h = {(f16)v0[0], (f16)v0[1], (f16)v0[2], (f16)v0[3], (f16)v1[0], (f16)v1[1], (f16)v1[2], (f16)v1[3]};
          o = __builtin_bit_cast(i32x4, h);
        }
        __builtin_amdgcn_raw_buffer_store_b128(o, ry, po != OOB ? po + (t >> 1) * 64 : OOB, 0, 0);
      }
    }
  };
  i32x4 b[NSETS][MT];
  int tile = stream;
#pragma unroll
  for (int s = 0; s < NSETS - 1; ++s) { load_tile(tile + s * a.nstreams, b[s]); dummy_stores(); }
  for (; tile < a.ntiles; tile += NSETS * a.nstreams) {
#pragma unroll
    for (int u = 0; u < NSETS; ++u) step(tile + u * a.nstreams, b[u], b[(u + NSETS - 1) % NSETS]);
  }
}

// "split": x and y of a launch live in allocations of their OWN (12 inputs, 3 outputs, hipMalloc each -- what a framework's tensors are)
// instead of 1-GiB regions of one 3-GiB allocation
static bool g_split = false;
static char* g_xs[12];
static char* g_ys[3];

template <int MT, int NSETS, int WPS, bool RELU16>
void run(char* pool, int P, int Cin, int Cout) {
  const size_t region = 1ull << 30;
  Args a;
  a.P = P; a.Cin = Cin; a.Cout = Cout; a.ngroups = Cout / 64;
  a.ntiles = (P + 16 * MT - 1) / (16 * MT);
  int per_cu = 0;
  CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, reinterpret_cast<const void*>(&k<MT, NSETS, WPS, RELU16>), 256, 0));
  int waves = 256 * per_cu * 4;
  a.nstreams = waves / a.ngroups;
  if (a.nstreams > a.ntiles) a.nstreams = a.ntiles;
  waves = a.nstreams * a.ngroups;
  a.x_bytes = (unsigned)((size_t)P * Cin * 2); a.y_bytes = (unsigned)((size_t)P * Cout * 2);
  a.w = pool + 3 * region - (1 << 20); a.bias = reinterpret_cast<const float*>(pool + 3 * region - (2 << 20));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  float best = 1e9f, sum = 0.f;
  const int reps = 9;
  for (int i = 0; i < reps + 2; ++i) {
    // (pool mode: inputs from three 512 MiB regions that are only ever read, outputs into three others -- an earlier version read what the
    //  launch before had written, i.e. from the Infinity Cache, and flattered every variant by ~10 us)
    a.x = g_split ? g_xs[i % 12] : pool + (size_t)(i % 3) * (region / 2);
    a.y = g_split ? g_ys[i % 3] : pool + (3ull << 29) + (size_t)(i % 3) * (region / 2);
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL((k<MT, NSETS, WPS, RELU16>), dim3((waves + 3) / 4), dim3(256), 0, 0, a);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    if (i >= 2) { best = ms < best ? ms : best; sum += ms; }
  }
  const double bytes = (double)P * (Cin + Cout) * 2;
  printf("Cin %3d Cout %3d  MT %d  NSETS %d  WPS %d  relu16 %d  blocks/CU %d  waves %5d  best %7.2f us  mean %7.2f us  %5.0f GB/s  %.3f of 8 TB/s\n", Cin, Cout,
         MT, NSETS, WPS, (int)RELU16, per_cu, waves, best * 1e3, sum / reps * 1e3, bytes / best / 1e6, bytes / best / 1e6 / 8000.0);
}

template <int MT, int NSETS, int WPS>
void both(char* pool, int P, int Cin, int Cout) {
  run<MT, NSETS, WPS, false>(pool, P, Cin, Cout);
  run<MT, NSETS, WPS, true>(pool, P, Cin, Cout);
}

// float16 values in [0.5, 2) with random mantissas (argument "random"): activations that look like data, not like a memset
__global__ void fill_random(unsigned* p, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    unsigned long long z = 0x9E3779B97F4A7C15ull * (i + 1);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    const unsigned r = (unsigned)(z >> 32);
    p[i] = (0x3800u | (r & 0x07ffu)) | ((0x3800u | ((r >> 16) & 0x07ffu)) << 16);
  }
}

int main(int argc, char** argv) {
  char* pool;
  CK(hipMalloc(&pool, 3ull << 30));
  CK(hipMemset(pool, 0, 3ull << 30));
  const bool random = argc > 1 && argv[1][0] == 'r';
  if (random) hipLaunchKernelGGL(fill_random, dim3(8192), dim3(256), 0, 0, reinterpret_cast<unsigned*>(pool), (3ull << 30) / 4);
  g_split = argc > 2 && argv[2][0] == 's';
  if (g_split) {
    for (int i = 0; i < 12; ++i) {
      CK(hipMalloc(&g_xs[i], 30u << 20));
      if (random) hipLaunchKernelGGL(fill_random, dim3(2048), dim3(256), 0, 0, reinterpret_cast<unsigned*>(g_xs[i]), (size_t)(30u << 20) / 4);
      else CK(hipMemset(g_xs[i], 0, 30u << 20));
    }
    for (int i = 0; i < 3; ++i) { CK(hipMalloc(&g_ys[i], 120u << 20)); CK(hipMemset(g_ys[i], 0, 120u << 20)); }
  }
  printf("pool contents: %s; %s\n", random ? "random float16 in [0.5, 2)" : "zeros", g_split ? "x / y in allocations of their own" : "x / y = regions of one 3 GiB allocation");
  CK(hipDeviceSynchronize());
  const int shapes[2][3] = {{32 * 94 * 311, 16, 64}, {32 * 47 * 156, 32, 128}};
  for (auto& s : shapes) {
    both<4, 2, 3>(pool, s[0], s[1], s[2]);
    both<4, 3, 3>(pool, s[0], s[1], s[2]);
    both<4, 4, 2>(pool, s[0], s[1], s[2]);
    both<2, 2, 4>(pool, s[0], s[1], s[2]);
    both<2, 4, 4>(pool, s[0], s[1], s[2]);
    both<8, 2, 2>(pool, s[0], s[1], s[2]);
    printf("\n");
  }
  return 0;
}
