// Microbenchmark: what ONE compute unit's vector-memory path delivers to registers / LDS, in bytes per shader clock -- the figure that
// bounds a GEMM-shaped kernel whose operands come through `global_load_dwordx4` / `buffer_load_dwordx4 ... lds` (conv1x1_pipe,
// gemm1x1.hip: 20 KiB per 16-MFMA chunk and CU).  Every wave issues ITER batches of 8 x 1-KiB loads (a lane moves 16 bytes, a wave
// instruction 1 KiB of contiguous memory) over a footprint of FP bytes per workgroup -- 16 KiB = L1-resident, 1 MiB = L2-resident (all
// workgroups read the SAME region: the weight-fragment pattern), or 1 MiB per workgroup of a 1 GiB pool (HBM / Infinity Cache) -- with
// 8 loads in flight per wave; shader cycles by s_memtime around the loop of wave 0.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/cu_load_rate tools/microbench/cu_load_rate.hip && /tmp/cu_load_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

typedef int i32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

template <bool LDS>
__global__ __launch_bounds__(1024) void rate(const unsigned char* src, size_t wg_stride, unsigned fp_mask, int iters, unsigned long long* out, i32x4* sink) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  const unsigned lds_addr = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)lds;
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const unsigned char* base = src + (size_t)blockIdx.x * wg_stride;
  const unsigned long long addr = (unsigned long long)(uintptr_t)base;
  const i32x4 rs = {(int)(unsigned)addr, (int)(unsigned)((addr >> 32) & 0xffffu), (int)0x7fffffff, 0x00020000};
  i32x4 acc = {0, 0, 0, 0};
  unsigned off = (unsigned)(wave * 8192 + lane * 16);
  __syncthreads();
  const unsigned long long t0 = clock64();
  for (int i = 0; i < iters; ++i) {
    if (LDS) {
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const unsigned o = (off + k * 1024) & fp_mask;
        asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, 0 offen lds" ::"v"(o), "s"(rs), "s"(lds_addr + (unsigned)(wave * 8192 + k * 1024)) : "memory", "m0");
      }
      asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    } else {
      i32x4 v[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) v[k] = *reinterpret_cast<const i32x4*>(base + ((off + k * 1024) & fp_mask));
#pragma unroll
      for (int k = 0; k < 8; ++k) acc += v[k];
    }
    off += 8192 * (blockDim.x >> 6);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  const unsigned long long t1 = clock64();
  if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
  if (acc[0] == 0x12345678) sink[threadIdx.x] = acc;
}

int main() {
  const size_t pool = 1ull << 30;
  unsigned char* src;
  unsigned long long* out;
  i32x4* sink;
  CK(hipMalloc(&src, pool + (2 << 20)));
  CK(hipMemset(src, 1, pool + (2 << 20)));
  CK(hipMalloc(&out, 4096 * 8));
  CK(hipMalloc(&sink, 1024 * 16));
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&rate<true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  printf("%-10s %-22s %6s %6s | %10s %12s\n", "path", "footprint", "WGs", "waves", "B/clk/CU", "GB/s chip");
  const int iters = 400;
  for (int lds = 0; lds < 2; ++lds)
    for (int fp = 0; fp < 3; ++fp)
      for (int wgs : {1, 256, 512})
        for (int waves : {4, 8, 16}) {
          if (wgs == 512 && waves == 16) continue;
          const unsigned mask = fp == 0 ? (16u << 10) - 1 : (1u << 20) - 1;
          const size_t stride = fp == 2 ? (1u << 20) : 0;      // private 1 MiB per workgroup (512 MiB for 512 workgroups) vs one shared region
          unsigned long long h[4096];
          float best = 0.f;
          for (int rep = 0; rep < 3; ++rep) {
            if (lds) hipLaunchKernelGGL(rate<true>, dim3(wgs), dim3(waves * 64), waves * 8192, 0, src, stride, mask, iters, out, sink);
            else hipLaunchKernelGGL(rate<false>, dim3(wgs), dim3(waves * 64), 0, 0, src, stride, mask, iters, out, sink);
            CK(hipDeviceSynchronize());
            CK(hipMemcpy(h, out, wgs * 8, hipMemcpyDeviceToHost));
            double cyc = 0;
            for (int i = 0; i < wgs; ++i) cyc += (double)h[i];
            cyc /= wgs;
            const float bpc = (float)((double)iters * 8 * 1024 * waves / cyc) * (wgs > 256 ? 2.f : 1.f);
            if (bpc > best) best = bpc;
          }
          printf("%-10s %-22s %6d %6d | %10.1f %12.0f\n", lds ? "lds-dma" : "registers", fp == 0 ? "16 KiB (L1)" : fp == 1 ? "1 MiB shared (L2)" : "1 MiB per WG (HBM/MALL)",
                 wgs, waves, best, best * 2.4 * (wgs >= 256 ? 256 : wgs));
        }
  return 0;
}
