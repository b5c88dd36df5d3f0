// Microbenchmark: the same copy kernel on buffers that are (a) allocations of their own (hipMalloc per buffer: what a framework's
// tensors are) or (b) sub-ranges of ONE large allocation, at several buffer sizes and sub-range alignments.  Three source / destination
// pairs rotate (nothing survives in the 256 MiB Infinity Cache for the larger sizes).
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/alloc_granularity tools/microbench/alloc_granularity.hip && /tmp/alloc_granularity
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

typedef int i32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

__global__ __launch_bounds__(256) void copyk(const i32x4* __restrict__ s, i32x4* __restrict__ d, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    i32x4 v = s[i];
    v[0] += 1;
    d[i] = v;
  }
}

float time_pairs(char* src[3], char* dst[3], size_t bytes) {
  hipEvent_t a, b;
  CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  float best = 1e9f;
  for (int i = 0; i < 11; ++i) {
    CK(hipEventRecord(a));
    hipLaunchKernelGGL(copyk, dim3(8192), dim3(256), 0, 0, (const i32x4*)src[i % 3], (i32x4*)dst[i % 3], bytes / 16);
    CK(hipEventRecord(b));
    CK(hipEventSynchronize(b));
    float ms;
    CK(hipEventElapsedTime(&ms, a, b));
    if (i >= 2 && ms < best) best = ms;
  }
  return best;
}

int main() {
  const size_t MB = 1u << 20;
  const size_t sizes[] = {30 * MB, 64 * MB, 75 * MB, 120 * MB, 128 * MB, 256 * MB};
  char* pool;
  CK(hipMalloc(&pool, 3ull << 30));
  CK(hipMemset(pool, 1, 3ull << 30));
  printf("pool base %% 1 GiB = %zu MiB\n", (size_t)((uintptr_t)pool % (1ull << 30)) / MB);
  for (size_t sz : sizes) {
    // (a) own allocations
    char *s[3], *d[3];
    for (int i = 0; i < 3; ++i) { CK(hipMalloc(&s[i], sz)); CK(hipMalloc(&d[i], sz)); CK(hipMemset(s[i], 1, sz)); CK(hipMemset(d[i], 1, sz)); }
    CK(hipDeviceSynchronize());
    const float ta = time_pairs(s, d, sz);
    printf("size %4zu MiB  own allocations (bases %% 2 MiB: %zu KiB, %% 64 MiB: %zu MiB): %7.2f us = %5.0f GB/s", sz / MB,
           (size_t)((uintptr_t)s[0] % (2 * MB)) / 1024, (size_t)((uintptr_t)s[0] % (64 * MB)) / MB, ta * 1e3, 2.0 * sz / ta / 1e6);
    for (int i = 0; i < 3; ++i) { CK(hipFree(s[i])); CK(hipFree(d[i])); }
    // (b) sub-ranges of the pool, packed back to back (offsets = multiples of the size), and (c) with every sub-range shifted by 4 KiB + 256 B
    for (int i = 0; i < 3; ++i) { s[i] = pool + (size_t)(2 * i) * sz; d[i] = pool + (size_t)(2 * i + 1) * sz; }
    const float tb = time_pairs(s, d, sz);
    for (int i = 0; i < 3; ++i) { s[i] += 4096 + 256; d[i] += 4096 + 256; }
    const float tc = time_pairs(s, d, sz - 8192);
    printf("   | pool sub-ranges: %7.2f us = %5.0f GB/s   | shifted by 4352 B: %7.2f us = %5.0f GB/s\n", tb * 1e3, 2.0 * sz / tb / 1e6, tc * 1e3,
           2.0 * (sz - 8192) / tc / 1e6);
  }
  // many small allocations first (a fragmented heap), then the same test at 120 MiB
  {
    char* junk[64];
    for (int i = 0; i < 64; ++i) CK(hipMalloc(&junk[i], (size_t)(3 + (i % 5)) * MB));
    const size_t sz = 120 * MB;
    char *s[3], *d[3];
    for (int i = 0; i < 3; ++i) { CK(hipMalloc(&s[i], sz)); CK(hipMalloc(&d[i], sz)); CK(hipMemset(s[i], 1, sz)); }
    CK(hipDeviceSynchronize());
    const float ta = time_pairs(s, d, sz);
    printf("size  120 MiB  own allocations behind 64 small ones: %7.2f us = %5.0f GB/s\n", ta * 1e3, 2.0 * sz / ta / 1e6);
  }
  return 0;
}
