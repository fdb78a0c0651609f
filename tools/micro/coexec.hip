// Microbenchmark: do fp32 MFMA and plain VALU from two waves of one SIMD overlap on gfx950?
// 512-thread blocks (2 waves per SIMD); waves 0-3 run role A, waves 4-7 role B.
//   mode 0: A = MFMA f32 16x16x4, B = idle      mode 1: A = idle, B = v_fma chain
//   mode 2: A = MFMA f32, B = v_fma             mode 3: A = MFMA bf16 16x16x16, B = v_fma   mode 4: A = MFMA bf16, B idle
//   mode 5: both MFMA f32                       mode 6: both v_fma
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef short bf16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

__global__ __launch_bounds__(512) void k(int mode, int iters, float* out) {
  const int wave = threadIdx.x >> 6;
  const bool roleA = wave < 4;
  int what = 0;   // 0 idle, 1 mfma f32, 2 fma, 3 mfma bf16
  if (mode == 0) what = roleA ? 1 : 0;
  if (mode == 1) what = roleA ? 0 : 2;
  if (mode == 2) what = roleA ? 1 : 2;
  if (mode == 3) what = roleA ? 3 : 2;
  if (mode == 4) what = roleA ? 3 : 0;
  if (mode == 5) what = 1;
  if (mode == 6) what = 2;
  if (mode == 7) what = roleA ? 4 : 0;   // interleaved MFMA f32 + fma in ONE wave, partner idle
  if (mode == 8) what = 4;               // both waves interleaved
  if (mode == 9) what = roleA ? 5 : 0;   // interleaved MFMA bf16 + fma in one wave
  float r = 0.f;
  if (what == 1) {
    f32x4 a0 = {0, 0, 0, 0}, a1 = a0, a2 = a0, a3 = a0;
    float x = threadIdx.x * 1e-3f, y = 1.0f + x;
    for (int i = 0; i < iters; ++i) {
      a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a0, 0, 0, 0);
      a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a1, 0, 0, 0);
      a2 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a2, 0, 0, 0);
      a3 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a3, 0, 0, 0);
    }
    r = a0[0] + a1[1] + a2[2] + a3[3];
  } else if (what == 2) {
    float v0 = threadIdx.x, v1 = 1.f, v2 = 2.f, v3 = 3.f, v4 = 4.f, v5 = 5.f, v6 = 6.f, v7 = 7.f;
    const float m = 0.999f, c = 1e-3f;
    for (int i = 0; i < iters; ++i) {   // 32 independent-ish fmas per iteration (same issue time as 4 MFMA 16x16x4 = 128 cyc)
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        v0 = fmaf(v0, m, c); v1 = fmaf(v1, m, c); v2 = fmaf(v2, m, c); v3 = fmaf(v3, m, c);
        v4 = fmaf(v4, m, c); v5 = fmaf(v5, m, c); v6 = fmaf(v6, m, c); v7 = fmaf(v7, m, c);
      }
    }
    r = v0 + v1 + v2 + v3 + v4 + v5 + v6 + v7;
  } else if (what == 3) {
    f32x4 a0 = {0, 0, 0, 0}, a1 = a0, a2 = a0, a3 = a0;
    bf16x8 x, y;
    for (int e = 0; e < 8; ++e) { x[e] = (__bf16)(threadIdx.x * 1e-3f); y[e] = (__bf16)1.0f; }
    for (int i = 0; i < iters; ++i) {
      a0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(x, y, a0, 0, 0, 0);
      a1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(x, y, a1, 0, 0, 0);
      a2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(x, y, a2, 0, 0, 0);
      a3 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(x, y, a3, 0, 0, 0);
    }
    r = a0[0] + a1[1] + a2[2] + a3[3];
  }
  if (what == 4) {
    f32x4 a0 = {0, 0, 0, 0}, a1 = a0, a2 = a0, a3 = a0;
    float x = threadIdx.x * 1e-3f, y = 1.0f + x;
    float v0 = threadIdx.x, v1 = 1.f, v2 = 2.f, v3 = 3.f, v4 = 4.f, v5 = 5.f, v6 = 6.f, v7 = 7.f;
    const float m = 0.999f, c = 1e-3f;
    for (int i = 0; i < iters; ++i) {
#define FMA8 v0 = fmaf(v0, m, c); v1 = fmaf(v1, m, c); v2 = fmaf(v2, m, c); v3 = fmaf(v3, m, c); v4 = fmaf(v4, m, c); v5 = fmaf(v5, m, c); v6 = fmaf(v6, m, c); v7 = fmaf(v7, m, c);
      a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a0, 0, 0, 0); FMA8 __builtin_amdgcn_sched_barrier(0);
      a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a1, 0, 0, 0); FMA8 __builtin_amdgcn_sched_barrier(0);
      a2 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a2, 0, 0, 0); FMA8 __builtin_amdgcn_sched_barrier(0);
      a3 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a3, 0, 0, 0); FMA8 __builtin_amdgcn_sched_barrier(0);
    }
    r = a0[0] + a1[1] + a2[2] + a3[3] + v0 + v1 + v2 + v3 + v4 + v5 + v6 + v7;
  } else if (what == 5) {
    f32x4 a0 = {0, 0, 0, 0}, a1 = a0, a2 = a0, a3 = a0;
    bf16x8 x, y;
    for (int e = 0; e < 8; ++e) { x[e] = (__bf16)(threadIdx.x * 1e-3f); y[e] = (__bf16)1.0f; }
    float v0 = threadIdx.x, v1 = 1.f, v2 = 2.f, v3 = 3.f, v4 = 4.f, v5 = 5.f, v6 = 6.f, v7 = 7.f;
    const float m = 0.999f, c = 1e-3f;
    for (int i = 0; i < iters; ++i) {
      a0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(x, y, a0, 0, 0, 0); FMA8 __builtin_amdgcn_sched_barrier(0);
      a1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(x, y, a1, 0, 0, 0); FMA8 __builtin_amdgcn_sched_barrier(0);
      a2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(x, y, a2, 0, 0, 0); FMA8 __builtin_amdgcn_sched_barrier(0);
      a3 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(x, y, a3, 0, 0, 0); FMA8 __builtin_amdgcn_sched_barrier(0);
    }
    r = a0[0] + a1[1] + a2[2] + a3[3] + v0 + v1 + v2 + v3 + v4 + v5 + v6 + v7;
  }
  if (r == 12345.678f) out[threadIdx.x] = r;
}

int main() {
  float* out; hipMalloc(&out, 4096);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int iters = 20000;
  for (int mode = 0; mode < 10; ++mode) {
    k<<<256, 512>>>(mode, 100, out);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<<<256, 512>>>(mode, iters, out);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("mode %d: %.3f ms  (%.1f cycles/iter at 2.4 GHz)\n", mode, ms, ms * 1e-3 * 2.4e9 / iters);
  }
  return 0;
}
