// Microbenchmark: what one step of conv1x1_pipe's K loop (gemm1x1.hip) costs a lone 4-wave workgroup, instruction class by class --
// 16 MFMAs (16x16x32 f16) in two halves, 4 fragment loads (global_load_dwordx4, saddr form), 1 LDS-DMA piece, 4 ds_read_b128, one counted
// wait + barrier -- each class switched on by a template flag, over 512 steps, shader cycles (s_memtime) of wave 0 per step.
//   hipcc --offload-arch=gfx950 -O3 -mllvm -amdgpu-mfma-vgpr-form=1 -o tools/microbench/bin/g1_loop_cost tools/microbench/g1_loop_cost.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

enum { F_MFMA = 1, F_W = 2, F_DMA = 4, F_LDS = 8, F_BAR = 16, F_WAIT = 32, F_BR = 64, F_BRT = 128, F_SALU = 256 };

template <int IMM>
__device__ __forceinline__ void wload(i32x4& dst, unsigned voff, const unsigned char* sbase) {
  asm volatile("global_load_dwordx4 %0, %1, %2 offset:%3" : "=v"(dst) : "v"(voff), "s"(sbase), "n"(IMM) : "memory");
}
__device__ __forceinline__ void dma16(unsigned voff, const i32x4& rsrc, unsigned lds_dst) {
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, 0 offen lds" ::"v"(voff), "s"(rsrc), "s"(lds_dst) : "memory", "m0");
}
__device__ __forceinline__ void mma(f32x4& acc, const i32x4& a, const i32x4& b) {
  acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), acc, 0, 0, 0);
}

template <int F>
__global__ __launch_bounds__(256, 2) void step_cost(const unsigned char* w, const unsigned char* x, int steps, unsigned long long* out, float* sink) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  const unsigned lds_addr = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)lds;
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const unsigned long long xa = (unsigned long long)(uintptr_t)x;
  const i32x4 rx = {(int)(unsigned)xa, (int)(unsigned)((xa >> 32) & 0xffffu), (int)0x7fffffff, 0x00020000};
  const unsigned wl = (unsigned)lane * 16u;
  i32x4 wf[4][4];
  f32x4 acc[4][4];
  for (int m = 0; m < 4; ++m) for (int t = 0; t < 4; ++t) { acc[m][t] = f32x4{0.f, 0.f, 0.f, 0.f}; }
  for (int s = 0; s < 4; ++s) for (int t = 0; t < 4; ++t) wf[s][t] = i32x4{lane, s, t, 1};
  i32x4 bfa[2] = {i32x4{1, 2, 3, lane}, i32x4{4, 5, 6, lane}};
  const unsigned char* lrd = lds + lane * 16;
  __syncthreads();
  const unsigned long long t0 = clock64();
#pragma unroll 1
  for (int c0 = 0; c0 < steps; c0 += 4) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int c = c0 + u;
      __builtin_amdgcn_sched_barrier(0);
      i32x4 bfb[2];
      if (F & F_LDS) { bfb[0] = *reinterpret_cast<const i32x4*>(lrd + ((c & 7) * 4096) + 2048); bfb[1] = *reinterpret_cast<const i32x4*>(lrd + ((c & 7) * 4096) + 3072); }
      else { bfb[0] = bfa[1]; bfb[1] = bfa[0]; }
      if (F & F_W) {
        const unsigned char* sb = w + (size_t)((c + 3) & 31) * 16384 + wave * 4096;
        wload<0>(wf[(u + 3) & 3][0], wl, sb); wload<1024>(wf[(u + 3) & 3][1], wl, sb); wload<2048>(wf[(u + 3) & 3][2], wl, sb); wload<3072>(wf[(u + 3) & 3][3], wl, sb);
      }
      if (F & F_MFMA)
        for (int m = 0; m < 2; ++m) for (int t = 0; t < 4; ++t) mma(acc[m][t], wf[u][t], bfa[m]);
      __builtin_amdgcn_sched_barrier(0);
      // F_BR: four wave-uniform branches per step that are never taken; F_BRT: four that always are (over one s_nop); F_SALU: sixteen
      // dependent scalar adds per step -- what control flow and scalar arithmetic cost a loop that is otherwise bound by its MFMAs
      if (F & F_BR) {
#pragma unroll
        for (int k = 0; k < 4; ++k) if (steps == 12345 + k + u) asm volatile("s_nop 7\n\ts_nop 7" ::: "memory");
      }
      if (F & F_BRT) {
#pragma unroll
        for (int k = 0; k < 4; ++k) if (steps != 12345 + k + u) asm volatile("s_nop 0" ::: "memory");
      }
      if (F & F_SALU) {
        int sa = steps;
#pragma unroll
        for (int k = 0; k < 16; ++k) asm volatile("s_add_i32 %0, %0, %1" : "+s"(sa) : "s"(c));
        if (sa == 77) asm volatile("s_nop 1");
      }
      constexpr int N = ((F & F_W) ? 8 : 0) + ((F & F_DMA) ? 2 : 0);
      if ((F & F_WAIT) && (F & F_BAR)) asm volatile("s_waitcnt vmcnt(%4)\n\ts_barrier" : "+v"(wf[(u + 1) & 3][0]), "+v"(wf[(u + 1) & 3][1]), "+v"(wf[(u + 1) & 3][2]), "+v"(wf[(u + 1) & 3][3]) : "n"(N) : "memory");
      else if (F & F_WAIT) asm volatile("s_waitcnt vmcnt(%4)" : "+v"(wf[(u + 1) & 3][0]), "+v"(wf[(u + 1) & 3][1]), "+v"(wf[(u + 1) & 3][2]), "+v"(wf[(u + 1) & 3][3]) : "n"(N) : "memory");
      else if (F & F_BAR) asm volatile("s_barrier" ::: "memory");
      if (F & F_LDS) { bfa[0] = *reinterpret_cast<const i32x4*>(lrd + (((c + 1) & 7) * 4096)); bfa[1] = *reinterpret_cast<const i32x4*>(lrd + (((c + 1) & 7) * 4096) + 1024); }
      __builtin_amdgcn_sched_barrier(0);
      if (F & F_DMA) dma16((unsigned)(((c + 7) & 63) * 4096 + wave * 1024 + lane * 16), rx, lds_addr + (unsigned)(((c + 7) & 7) * 4096 + wave * 1024));
      if (F & F_MFMA)
        for (int m = 2; m < 4; ++m) for (int t = 0; t < 4; ++t) mma(acc[m][t], wf[u][t], bfb[m - 2]);
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  const unsigned long long t1 = clock64();
  if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
  float r = 0.f;
  for (int m = 0; m < 4; ++m) for (int t = 0; t < 4; ++t) r += acc[m][t][0] + acc[m][t][3];
  if (r == 123.456f) sink[threadIdx.x] = r;
}

template <int F>
void run(const char* name, const unsigned char* w, const unsigned char* x, unsigned long long* out, float* sink, int wgs) {
  const int steps = 512;
  unsigned long long h[1024];
  double best = 1e30;
  for (int rep = 0; rep < 3; ++rep) {
    hipLaunchKernelGGL(step_cost<F>, dim3(wgs), dim3(256), 32768, 0, w, x, steps, out, sink);
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(h, out, wgs * 8, hipMemcpyDeviceToHost));
    double cyc = 0;
    for (int i = 0; i < wgs; ++i) cyc += (double)h[i];
    cyc /= wgs * (double)steps;
    if (cyc < best) best = cyc;
  }
  printf("%-52s %4d WG | %7.1f cycles per step\n", name, wgs, best);
}

int main(int argc, char**) {
  unsigned char *w, *x;
  unsigned long long* out;
  float* sink;
  CK(hipMalloc(&w, 1 << 20)); CK(hipMemset(w, argc > 1 ? 0x3c : 0, 1 << 20));
  CK(hipMalloc(&x, 1 << 20)); CK(hipMemset(x, argc > 1 ? 0x35 : 0, 1 << 20));
  CK(hipMalloc(&out, 8192)); CK(hipMalloc(&sink, 4096));
  for (int wgs : {1, 256}) {
    run<F_MFMA>("16 MFMAs", w, x, out, sink, wgs);
    run<F_MFMA | F_BAR>("16 MFMAs + barrier", w, x, out, sink, wgs);
    run<F_MFMA | F_BAR | F_LDS>("16 MFMAs + barrier + 4 ds_read_b128", w, x, out, sink, wgs);
    run<F_MFMA | F_W | F_WAIT>("16 MFMAs + 4 fragment loads + wait", w, x, out, sink, wgs);
    run<F_MFMA | F_DMA | F_WAIT>("16 MFMAs + 1 DMA piece + wait", w, x, out, sink, wgs);
    run<F_MFMA | F_W | F_DMA | F_WAIT>("16 MFMAs + 4 loads + 1 piece + wait", w, x, out, sink, wgs);
    run<F_MFMA | F_W | F_DMA | F_WAIT | F_BAR>("16 MFMAs + 4 loads + 1 piece + wait + barrier", w, x, out, sink, wgs);
    run<F_MFMA | F_W | F_DMA | F_WAIT | F_BAR | F_LDS>("the whole step", w, x, out, sink, wgs);
    run<F_MFMA | F_W | F_DMA | F_WAIT | F_BAR | F_LDS | F_BR>("the whole step + 4 untaken branches", w, x, out, sink, wgs);
    run<F_MFMA | F_W | F_DMA | F_WAIT | F_BAR | F_LDS | F_BRT>("the whole step + 4 taken branches", w, x, out, sink, wgs);
    run<F_MFMA | F_W | F_DMA | F_WAIT | F_BAR | F_LDS | F_SALU>("the whole step + 16 scalar adds", w, x, out, sink, wgs);
    run<F_MFMA | F_BR>("16 MFMAs + 4 untaken branches", w, x, out, sink, wgs);
    run<F_MFMA | F_BRT>("16 MFMAs + 4 taken branches", w, x, out, sink, wgs);
    run<F_W | F_DMA | F_WAIT | F_BAR | F_LDS>("the whole step without MFMAs", w, x, out, sink, wgs);
    run<F_W | F_WAIT>("4 fragment loads + wait only", w, x, out, sink, wgs);
  }
  return 0;
}
