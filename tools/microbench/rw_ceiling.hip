// Microbenchmark: what the memory system sustains for streams of a given READ : WRITE mix, on buffers far larger than the 256 MiB
// Infinity Cache (every launch touches fresh memory: three 512 MiB source regions and three destination regions of a 3 GiB pool, rotating).  The fire modules' stand-alone
// 1x1 convs are such streams -- squeeze1x1 reads 4-8x what it writes, expand1x1 writes 4x what it reads -- and the table in
// profiles/rNN_fire_1x1_standalone.txt prices them all against ONE 8 TB/s figure.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/rw_ceiling tools/microbench/rw_ceiling.hip && /tmp/rw_ceiling
// Kernels: every lane moves 16 bytes per access, a wave 1 KiB of contiguous memory per instruction, grid-stride.
//   R reads per W writes: (1,0) read-only (sum into a register, one store per thread at the end), (0,1) fill, (1,1) copy,
//   (4,1) squeeze-like, (1,4) expand-like, (1,2), (2,1).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

typedef int i32x4 __attribute__((ext_vector_type(4)));

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

// each "unit" = R reads + W writes of 16 bytes per lane; units are dealt to threads grid-stride
template <int R, int W>
__global__ __launch_bounds__(256) void rw(const i32x4* __restrict__ src, i32x4* __restrict__ dst, size_t units, i32x4* sink) {
  const size_t nthreads = (size_t)gridDim.x * blockDim.x;
  i32x4 acc = {0, 0, 0, 0};
  for (size_t u = (size_t)blockIdx.x * blockDim.x + threadIdx.x; u < units; u += nthreads) {
    i32x4 v[R > 0 ? R : 1];
#pragma unroll
    for (int r = 0; r < R; ++r) v[r] = src[(size_t)r * units + u];       // R planes of `units` vectors: each plane linear
    i32x4 sum = {(int)u, 2, 3, 4};
#pragma unroll
    for (int r = 0; r < R; ++r) sum += v[r];           // (every read feeds every write: none of them is dead)
    acc += sum;
#pragma unroll
    for (int w = 0; w < W; ++w) {
      i32x4 o = sum;
      o[1] += w;
      dst[(size_t)w * units + u] = o;
    }
  }
  if (W == 0 && acc[0] == 0x12345678) sink[threadIdx.x] = acc;            // (keeps the reads alive)
}

template <int R, int W>
void run(char* pool, size_t pool_bytes, size_t bytes_per_launch, int blocks, i32x4* sink) {
  const size_t units = bytes_per_launch / ((size_t)(R + W) * 16);
  const size_t rbytes = units * 16 * R, wbytes = units * 16 * W;
  // SIX 512 MiB regions: launches read from regions 0..2 and write to regions 3..5, both rotating -- a source is never something an
  // earlier launch wrote (an earlier version read what the launch before had written: those "HBM reads" came from the Infinity Cache)
  const size_t region = 512ull << 20;
  (void)pool_bytes;
  hipEvent_t a, b;
  CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  float best = 1e9f, sum = 0.f;
  const int reps = 9;
  for (int i = 0; i < reps + 2; ++i) {
    // source and destination in DIFFERENT 1 GiB regions, both rotating
    const char* s = pool + (size_t)(i % 3) * region;
    char* d = pool + (size_t)(3 + i % 3) * region;
    CK(hipEventRecord(a));
    hipLaunchKernelGGL((rw<R, W>), dim3(blocks), dim3(256), 0, 0, (const i32x4*)s, (i32x4*)d, units, sink);
    CK(hipEventRecord(b));
    CK(hipEventSynchronize(b));
    float ms;
    CK(hipEventElapsedTime(&ms, a, b));
    if (i >= 2) { best = ms < best ? ms : best; sum += ms; }
  }
  const double tot = (double)(rbytes + wbytes);
  printf("R:W %d:%d  %7.1f MB/launch  blocks %5d  best %8.2f us  mean %8.2f us  total %6.0f GB/s  read %6.0f  write %6.0f (best)\n", R, W,
         tot / 1e6, blocks, best * 1e3, sum / reps * 1e3, tot / best / 1e6, rbytes / best / 1e6, wbytes / best / 1e6);
}

// random bits (argument "random"): data that looks like data, not like a memset
__global__ void fill_random(unsigned* p, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    unsigned long long z = 0x9E3779B97F4A7C15ull * (i + 1);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    p[i] = (unsigned)(z >> 32);
  }
}

int main(int argc, char** argv) {
  const size_t pool_bytes = 3ull << 30;
  char* pool;
  i32x4* sink;
  CK(hipMalloc(&pool, pool_bytes));
  CK(hipMalloc(&sink, 4096));
  CK(hipMemset(pool, 1, pool_bytes));
  const bool random = argc > 1 && argv[1][0] == 'r';
  if (random) hipLaunchKernelGGL(fill_random, dim3(8192), dim3(256), 0, 0, reinterpret_cast<unsigned*>(pool), pool_bytes / 4);
  printf("pool contents: %s\n", random ? "random bits" : "0x01 bytes");
  CK(hipDeviceSynchronize());
  const size_t sizes[] = {75u << 20, 150u << 20, 400u << 20};
  const int grids[] = {8192};
  for (size_t sz : sizes) {
    for (int g : grids) {
      run<1, 0>(pool, pool_bytes, sz, g, sink);
      run<0, 1>(pool, pool_bytes, sz, g, sink);
      run<1, 1>(pool, pool_bytes, sz, g, sink);
      run<4, 1>(pool, pool_bytes, sz, g, sink);
      run<2, 1>(pool, pool_bytes, sz, g, sink);
      run<1, 2>(pool, pool_bytes, sz, g, sink);
      run<1, 4>(pool, pool_bytes, sz, g, sink);
    }
    printf("\n");
  }
  return 0;
}
