// Microbenchmark: does the LANE -> ADDRESS mapping of a conv epilogue's stores matter once the data is the same?
// The stream of fire2/expand1x1 (batch 32: 935 k pixels, 32 B read and 128 B written per pixel, 150 MB per launch), with the loop
// structure of conv1x1_stream (persistent waves, 64-pixel steps, the next step's loads issued before this step's stores) and no
// arithmetic, fresh memory every launch (regions rotate through 3 GiB):
//   shape 0  MFMA D layout, conv1x1_stream's PERM stores: lane (j = l & 15, g = l >> 4) writes pixel j, bytes [64 s + 16 g, +16) of its
//            128-byte row in store s -- a store instruction covers 16 pixels x 64 B, 128 B apart, ADJACENT LANES 128 B APART
//   shape 1  linear: lane l writes bytes [16 (l & 7), +16) of pixel 8 s + (l >> 3): a store instruction covers 1 KiB contiguous,
//            adjacent lanes adjacent
//   shape 2  shape 0's segments with adjacent lanes adjacent: lane l writes pixel (l >> 2), bytes [64 s + 16 (l & 3), +16)
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/store_shape tools/microbench/store_shape.hip && /tmp/store_shape
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

typedef int i32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

template <int SHAPE, int MT>
__global__ __launch_bounds__(256) void k(const char* __restrict__ x, char* __restrict__ y, int ntiles, int nwaves) {
  const int lane = threadIdx.x & 63, j = lane & 15, g = lane >> 4;
  const int wave = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (wave >= nwaves) return;
  auto load_tile = [&](int t, i32x4 (&v)[MT]) {
#pragma unroll
    for (int m = 0; m < MT; ++m) {
      const size_t p = ((size_t)t * MT + m) * 16 + j;
      v[m] = g < 2 ? *reinterpret_cast<const i32x4*>(x + p * 32 + g * 16) : i32x4{0, 0, 0, 0};
    }
  };
  i32x4 cur[MT], nxt[MT];
  int t = wave;
  if (t < ntiles) load_tile(t, cur);
  for (; t < ntiles; t += nwaves) {
    if (t + nwaves < ntiles) load_tile(t + nwaves, nxt);
#pragma unroll
    for (int m = 0; m < MT; ++m) {
      const size_t p0 = ((size_t)t * MT + m) * 16;
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        size_t off;
        if (SHAPE == 0) off = (p0 + j) * 128 + s * 64 + g * 16;
        else if (SHAPE == 1) off = (p0 + s * 8 + (lane >> 3)) * 128 + (lane & 7) * 16;
        else off = (p0 + (lane >> 2)) * 128 + s * 64 + (lane & 3) * 16;
        i32x4 o = cur[m];
        o[1] += s;
        *reinterpret_cast<i32x4*>(y + off) = o;
      }
    }
#pragma unroll
    for (int m = 0; m < MT; ++m) cur[m] = nxt[m];
  }
}

template <int SHAPE, int MT>
void run(char* pool, size_t npix, int nwaves) {
  const size_t region = 1ull << 30;
  hipEvent_t a, b;
  CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  const int ntiles = (int)(npix / (16 * MT));
  float best = 1e9f, sum = 0.f;
  const int reps = 9;
  for (int i = 0; i < reps + 2; ++i) {
    const char* s = pool + (size_t)(i % 3) * (region / 2);            // sources: three 512 MiB regions that are only ever read
    char* d = pool + (3ull << 29) + (size_t)(i % 3) * (region / 2);    // destinations: three others
    CK(hipEventRecord(a));
    hipLaunchKernelGGL((k<SHAPE, MT>), dim3((nwaves + 3) / 4), dim3(256), 0, 0, s, d, ntiles, nwaves);
    CK(hipEventRecord(b));
    CK(hipEventSynchronize(b));
    float ms;
    CK(hipEventElapsedTime(&ms, a, b));
    if (i >= 2) { best = ms < best ? ms : best; sum += ms; }
  }
  const double bytes = (double)ntiles * 16 * MT * 160;
  printf("shape %d  MT %d  waves %6d  %6.1f MB  best %7.2f us  mean %7.2f us  %6.0f GB/s (best)\n", SHAPE, MT, nwaves, bytes / 1e6, best * 1e3,
         sum / reps * 1e3, bytes / best / 1e6);
}

int main() {
  char* pool;
  CK(hipMalloc(&pool, 3ull << 30));
  CK(hipMemset(pool, 1, 3ull << 30));
  CK(hipDeviceSynchronize());
  const size_t npix = 32ull * 94 * 311;
  for (int waves : {4096, 8192, 16384}) {
    run<0, 4>(pool, npix, waves);
    run<1, 4>(pool, npix, waves);
    run<2, 4>(pool, npix, waves);
    run<0, 1>(pool, npix, waves);
    run<1, 1>(pool, npix, waves);
    run<2, 1>(pool, npix, waves);
  }
  return 0;
}
