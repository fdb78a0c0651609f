// Microbenchmark: what does a global_load_lds_dwordx4 (HBM -> LDS DMA) cost the issuing wave, and do the wave's own
// LDS instructions wait for an outstanding DMA?  One wave per SIMD (256-thread blocks, one block per CU).
//   mode 0: loop { VALU work (~W cycles); 8 ds_read_b128 of an unrelated LDS region }              (no DMA)
//   mode 1: loop { 8 DMA into region B; VALU work; 8 ds_read_b128 of region A }                    (DMA, reads after W cycles)
//   mode 2: loop { 8 DMA into region B; 8 ds_read_b128 of region A; VALU work }                    (reads right behind the DMA)
//   mode 3: loop { 8 plain global_load_dwordx4 into registers (kept); VALU work; 8 ds_read_b128 }  (same traffic, no LDS write)
//   mode 4: like 1 with s_waitcnt vmcnt(0) right after the VALU work (cost of waiting for the data itself)
//   mode 5: wave 0 as mode 4 while waves 1-3 of the CU stream ds_read_b128 (does OTHER waves' LDS traffic slow the issue?)
//   mode 6: wave 0 as mode 4 while waves 1-3 issue bf16 MFMAs back to back
//   mode 7: wave 0 as mode 4, waves 1-3 idle (the reference for 5 and 6; all three report wave 0's clocks)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

__device__ __forceinline__ void dma16_so(const void* sbase, uint32_t voff, uint32_t lds_base) {
  uint32_t keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2 nt\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(voff), "s"(sbase), "s"(lds_base) : "memory");
}

__global__ __launch_bounds__(256) void k(int mode, int iters, int work, const float4* __restrict__ src, size_t stride4, float* out,
                                         unsigned long long* cyc) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  float* A = sm + wave * 8192;            // 16 KB of reads
  float* B = A + 4096;                    // 16 KB DMA landing zone
  for (int i = lane; i < 4096; i += 64) A[i] = (float)i;
  const uint32_t ldsB = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)B);
  const float4* mine = src + ((size_t)blockIdx.x * 4 + wave) * stride4;
  float acc = 0.f, v = lane * 1e-3f;
  float4 keep[8];
  for (int e = 0; e < 8; ++e) keep[e] = make_float4(0, 0, 0, 0);
  __syncthreads();
  if (mode >= 5) {
    typedef float f32x16 __attribute__((ext_vector_type(16)));
    typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
    if (wave != 0) {
      if (mode == 5) {
        for (int it = 0; it < iters * 40; ++it) {
#pragma unroll
          for (int e = 0; e < 8; ++e) { const float4 r = *reinterpret_cast<const float4*>(A + (((e * 64 + lane + it) * 4) & 4095)); acc += r.x + r.w; }
          __builtin_amdgcn_sched_barrier(0);
        }
      } else if (mode == 6) {
        f32x16 c = {0};
        bf16x8 x, y;
        for (int e = 0; e < 8; ++e) { x[e] = (__bf16)(lane * 1e-3f); y[e] = (__bf16)1.0f; }
        for (int it = 0; it < iters * 100; ++it) c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, c, 0, 0, 0);
        acc += c[0];
      }
      if (acc == 12345.678f) out[threadIdx.x] = acc;
      return;
    }
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
      const float4* p = mine + (size_t)(it & 63) * 512;
#pragma unroll
      for (int e = 0; e < 8; ++e) dma16_so(p, (uint32_t)((e * 64 + lane) * 16), ldsB + e * 1024);
      __builtin_amdgcn_sched_barrier(0);
      for (int w = 0; w < work; ++w) v = fmaf(v, 0.999f, 1e-3f);
      __builtin_amdgcn_sched_barrier(0);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
      for (int e = 0; e < 8; ++e) { const float4 r = *reinterpret_cast<const float4*>(A + ((e * 64 + lane) * 4 & 4095)); acc += r.x + r.w; }
      __builtin_amdgcn_sched_barrier(0);
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    if (acc + v == 12345.678f) out[threadIdx.x] = acc + B[lane];
    if (blockIdx.x == 0 && threadIdx.x == 0) cyc[mode] = t1 - t0;
    return;
  }
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
    const float4* p = mine + (size_t)(it & 63) * 512;
    if (mode == 1 || mode == 2 || mode == 4) {
#pragma unroll
      for (int e = 0; e < 8; ++e) dma16_so(p, (uint32_t)((e * 64 + lane) * 16), ldsB + e * 1024);
    }
    if (mode == 3) {
#pragma unroll
      for (int e = 0; e < 8; ++e) keep[e] = p[e * 64 + lane];
    }
    if (mode == 2) {
#pragma unroll
      for (int e = 0; e < 8; ++e) { const float4 r = *reinterpret_cast<const float4*>(A + ((e * 64 + lane) * 4 & 4095)); acc += r.x + r.w; }
      __builtin_amdgcn_sched_barrier(0);
    }
    for (int w = 0; w < work; ++w) v = fmaf(v, 0.999f, 1e-3f);
    __builtin_amdgcn_sched_barrier(0);
    if (mode == 4) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (mode != 2) {
#pragma unroll
      for (int e = 0; e < 8; ++e) { const float4 r = *reinterpret_cast<const float4*>(A + ((e * 64 + lane) * 4 & 4095)); acc += r.x + r.w; }
    }
    __builtin_amdgcn_sched_barrier(0);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  const unsigned long long t1 = __builtin_readcyclecounter();
  for (int e = 0; e < 8; ++e) acc += keep[e].x;
  if (acc + v == 12345.678f) out[threadIdx.x] = acc + B[lane];
  if (blockIdx.x == 0 && threadIdx.x == 0) cyc[mode] = t1 - t0;
}

int main() {
  const size_t stride4 = 64 * 512;                       // float4 per wave: 64 distinct 8 KB tiles
  float4* src; float* out; unsigned long long* cyc;
  hipMalloc(&src, 256 * 4 * stride4 * 16); hipMemset(src, 0, 256 * 4 * stride4 * 16);
  hipMalloc(&out, 4096); hipMalloc(&cyc, 128);
  hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 4 * 8192 * 4);
  for (int work : {100, 500, 2000}) {
    for (int mode = 0; mode < 8; ++mode) {
      k<<<256, 256, 4 * 8192 * 4>>>(mode, 50, work, src, stride4, out, cyc);
      hipDeviceSynchronize();
      hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
      const int iters = 2000;
      hipEventRecord(e0);
      k<<<256, 256, 4 * 8192 * 4>>>(mode, iters, work, src, stride4, out, cyc);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      unsigned long long c[16]; hipMemcpy(c, cyc, 128, hipMemcpyDeviceToHost);
      printf("work %4d mode %d: %.3f ms, %.0f clocks/iter (s_memtime), %.1f ns/iter\n", work, mode, ms, (double)c[mode] / iters, ms * 1e6 / iters);
    }
  }
  return 0;
}
