// Microbenchmark: HBM write bandwidth of the NHWC store patterns the conv epilogues can produce.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/store_patterns tools/microbench/store_patterns.hip && /tmp/store_patterns
// A wave owns 16 pixels (lane&15) x CH channels; lane group g = lane>>4.
//   pattern 0: lane g owns 16 consecutive channels -> two 16 B stores at [g*32, g*32+16) and +16   (current layout, NT=4)
//   pattern 1: store p covers channels p*32 + g*8  -> each store writes 64 contiguous bytes per pixel
//   pattern 2: fully linear: lane l of the wave writes 16 B at l*16 (+1 KB per store)               (upper bound)
//   pattern 3: like 0 but one 32 B (dwordx8-equivalent: two adjacent x4) -> same as 0, order swapped (sanity)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

typedef int i32x4 __attribute__((ext_vector_type(4)));

template <int PAT>
__global__ __launch_bounds__(256) void k(i32x4* y, int npix, int row_bytes, int seg_off, int seg_bytes) {
  const int lane = threadIdx.x & 63, j = lane & 15, g = lane >> 4;
  const long wave = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  const long p = wave * 16 + j;
  if (wave * 16 >= npix) return;
  char* base = reinterpret_cast<char*>(y);
  const i32x4 v = {lane, (int)wave, 3, 4};
  const int nst = seg_bytes / 64;   // 16 B stores per lane to cover seg_bytes per pixel with 4 lane groups
  if (PAT == 2) {
    char* wb = base + wave * 16 * row_bytes;   // the wave's 16 pixel rows as one linear range (only valid when seg == row)
    for (int s = 0; s < nst * 1; ++s) *reinterpret_cast<i32x4*>(wb + ((long)s * 64 + lane) * 16) = v;
    return;
  }
  if (p >= npix) return;
  char* px = base + p * row_bytes + seg_off;
  for (int s = 0; s < nst; ++s) {
    int off;
    if (PAT == 0) off = g * (nst * 16) + s * 16;   // lane group owns nst*16 contiguous bytes
    else off = s * 64 + g * 16;                    // each store: 64 contiguous bytes per pixel
    *reinterpret_cast<i32x4*>(px + off) = v;
  }
}

// pattern 4: a 4-wave workgroup owns 16 pixels x 256 B; wave w writes the 32-byte slices [w*32, +32) and
// [128 + w*32, +32) of every pixel as 8-byte stores (one 16-cout MFMA tile per wave, 4 couts per lane).
__global__ __launch_bounds__(256) void k8(char* y, int npix) {
  const int lane = threadIdx.x & 63, j = lane & 15, g = lane >> 4, w = threadIdx.x >> 6;
  typedef int i32x2 __attribute__((ext_vector_type(2)));
  const i32x2 v = {lane, w};
  for (int r = 0; r < 8; ++r) {   // 8 tile rows -> 8 groups of 16 pixels
    const long p = ((long)blockIdx.x * 8 + r) * 16 + j;
    if (p >= npix) return;
    char* px = y + p * 256;
    *reinterpret_cast<i32x2*>(px + w * 32 + g * 8) = v;
    *reinterpret_cast<i32x2*>(px + 128 + w * 32 + g * 8) = v;
  }
}

float run8(char* y, int npix) {
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  const int blocks = (npix / 16 + 7) / 8;
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(k8, dim3(blocks), dim3(256), 0, 0, y, npix);
  hipEventRecord(a);
  const int it = 20;
  for (int i = 0; i < it; ++i) hipLaunchKernelGGL(k8, dim3(blocks), dim3(256), 0, 0, y, npix);
  hipEventRecord(b);
  hipEventSynchronize(b);
  float ms;
  hipEventElapsedTime(&ms, a, b);
  return ms / it;
}

template <int PAT>
float run(i32x4* y, int npix, int row_bytes, int seg_off, int seg_bytes) {
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  const int blocks = (npix / 16 + 3) / 4;
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(k<PAT>, dim3(blocks), dim3(256), 0, 0, y, npix, row_bytes, seg_off, seg_bytes);
  hipEventRecord(a);
  const int it = 20;
  for (int i = 0; i < it; ++i) hipLaunchKernelGGL(k<PAT>, dim3(blocks), dim3(256), 0, 0, y, npix, row_bytes, seg_off, seg_bytes);
  hipEventRecord(b);
  hipEventSynchronize(b);
  float ms;
  hipEventElapsedTime(&ms, a, b);
  return ms / it;
}

int main() {
  const int npix = 32 * 94 * 311 / 16 * 16;
  struct { int row, off, seg; const char* what; } cases[] = {
      {256, 0, 256, "128 ch fp16, whole row"},
      {256, 0, 128, "first 64 of 128 ch (expand1x1 half of a concat row)"},
      {256, 128, 128, "second 64 of 128 ch (expand3x3 half)"},
      {128, 0, 128, "64 ch fp16, whole row"},
      {512, 0, 512, "256 ch fp16, whole row"},
  };
  i32x4* y;
  hipMalloc(&y, (size_t)npix * 512 + 4096);
  for (auto& c : cases) {
    const double bytes = (double)npix * c.seg;
    const float t0 = run<0>(y, npix, c.row, c.off, c.seg);
    const float t1 = run<1>(y, npix, c.row, c.off, c.seg);
    const float t2 = c.seg == c.row ? run<2>(y, npix, c.row, c.off, c.seg) : 0.f;
    printf("%-55s %6.1f MB | lane-owns-run %7.1f us %5.0f GB/s | 64B-per-store %7.1f us %5.0f GB/s | linear %7.1f us %5.0f GB/s\n", c.what,
           bytes / 1e6, t0 * 1e3, bytes / t0 / 1e6, t1 * 1e3, bytes / t1 / 1e6, t2 * 1e3, t2 > 0 ? bytes / t2 / 1e6 : 0.0);
  }
  {
    const float t = run8(reinterpret_cast<char*>(y), npix);
    printf("%-55s %6.1f MB | 8-byte stores, 32 B slices per wave %7.1f us %5.0f GB/s\n", "128 ch fp16, one 16-cout tile per wave",
           npix * 256.0 / 1e6, t * 1e3, npix * 256.0 / t / 1e6);
  }
  return 0;
}
