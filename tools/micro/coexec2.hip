// Do the bf16 MFMAs of one wave and the split-type VALU work of ANOTHER wave of the same SIMD overlap on gfx950?
// (tools/micro/coexec.hip found "no" for fp32 MFMA and for v_mfma_f32_16x16x32_bf16 against an fma chain; the k_field_bwd_b3
// design question is about v_mfma_f32_32x32x16_bf16 -- 32 clocks of matrix pipe per 4 clocks of issue -- against the
// v_and / v_sub / v_perm stream of the operand splits.)  512-thread blocks = two waves per SIMD: waves 0-3 role A, 4-7 role B.
//   mode 0: A = MFMA stream, B idle            mode 1: A idle, B = split stream         mode 2: A = MFMA, B = splits
//   mode 3: both MFMA                          mode 4: both splits
//   mode 5: ONE wave per SIMD issuing MFMA + splits interleaved 1 : 6 (what k_field_bwd_b3 does), partner idle
//   mode 6: both waves of every SIMD run the interleaved stream
// Every wave reports its own clocks for `iters` units of work; a unit = 8 MFMAs (256 clocks of matrix pipe) or 48 split
// instructions (8 values split three ways).
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ uint32_t pack(uint32_t e, uint32_t o) { return __builtin_amdgcn_perm(o, e, 0x07060302u); }
__device__ __forceinline__ void split8(float (&x)[8], u32x4& H, u32x4& M, u32x4& L) {
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    const uint32_t h0 = __float_as_uint(x[2 * p]) & 0xffff0000u, h1 = __float_as_uint(x[2 * p + 1]) & 0xffff0000u;
    const float r0 = x[2 * p] - __uint_as_float(h0), r1 = x[2 * p + 1] - __uint_as_float(h1);
    const uint32_t m0 = __float_as_uint(r0) & 0xffff0000u, m1 = __float_as_uint(r1) & 0xffff0000u;
    const float q0 = r0 - __uint_as_float(m0), q1 = r1 - __uint_as_float(m1);
    H[p] = pack(h0, h1); M[p] = pack(m0, m1); L[p] = pack(__float_as_uint(q0), __float_as_uint(q1));
  }
}

__global__ __launch_bounds__(512) void k(int mode, int iters, unsigned long long* cyc, float* out) {
  const int wave = threadIdx.x >> 6;
  const bool roleA = wave < 4;
  int what = 0;       // 0 idle, 1 MFMA, 2 splits, 3 interleaved
  if (mode == 0) what = roleA ? 1 : 0;
  if (mode == 1) what = roleA ? 0 : 2;
  if (mode == 2) what = roleA ? 1 : 2;
  if (mode == 3) what = 1;
  if (mode == 4) what = 2;
  if (mode == 5) what = roleA ? 3 : 0;
  if (mode == 6) what = 3;
  f32x16 acc[4];
  for (int t = 0; t < 4; ++t) for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
  bf16x8 a, b;
  for (int e = 0; e < 8; ++e) { a[e] = (__bf16)(threadIdx.x * 1e-3f); b[e] = (__bf16)1.0f; }
  float x[8];
  for (int e = 0; e < 8; ++e) x[e] = 1.0f + 1e-3f * (threadIdx.x + 64 * e);
  uint32_t sink = 0;
  const unsigned long long t0 = __builtin_readcyclecounter();
  if (what == 1) {
    for (int i = 0; i < iters; ++i) {
#pragma unroll
      for (int u = 0; u < 8; ++u) acc[u & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[u & 3], 0, 0, 0);
    }
  } else if (what == 2) {
    for (int i = 0; i < iters; ++i) {
      u32x4 H, M, L;
      split8(x, H, M, L);
      sink ^= H[0] ^ M[1] ^ L[2] ^ H[3];
#pragma unroll
      for (int e = 0; e < 8; ++e) asm volatile("" : "+v"(x[e]));
    }
  } else if (what == 3) {
    for (int i = 0; i < iters; ++i) {
      u32x4 H, M, L;
      split8(x, H, M, L);
#pragma unroll
      for (int u = 0; u < 8; ++u) acc[u & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[u & 3], 0, 0, 0);
      sink ^= H[0] ^ M[1] ^ L[2] ^ H[3];
#pragma unroll
      for (int e = 0; e < 8; ++e) asm volatile("" : "+v"(x[e]));
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x002, 6, 0);
      }
    }
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  float r = 0.f;
  for (int t = 0; t < 4; ++t) r += acc[t][0] + acc[t][7];
  out[blockIdx.x * 512 + threadIdx.x] = r + __uint_as_float(sink) + x[0];
  if ((threadIdx.x & 63) == 0 && blockIdx.x == 0) cyc[wave] = t1 - t0;
}

int main() {
  unsigned long long* d; float* o;
  (void)hipMalloc(&d, 64); (void)hipMalloc(&o, 256 * 512 * 4);
  const int iters = 4000;
  const char* names[7] = {"A = MFMA, B idle", "A idle, B = splits", "A = MFMA, B = splits", "both MFMA", "both splits",
                          "one wave per SIMD, MFMA + splits interleaved", "both waves, MFMA + splits interleaved"};
  for (int mode = 0; mode < 7; ++mode) {
    k<<<256, 512>>>(mode, iters, d, o); (void)hipDeviceSynchronize();
    k<<<256, 512>>>(mode, iters, d, o); (void)hipDeviceSynchronize();
    unsigned long long c[8];
    (void)hipMemcpy(c, d, 64, hipMemcpyDeviceToHost);
    printf("mode %d  %-46s clocks per unit: role A (waves 0-3) %7.1f   role B (waves 4-7) %7.1f\n", mode, names[mode],
           (double)c[0] / iters, (double)c[4] / iters);
  }
  printf("unit = 8 x v_mfma_f32_32x32x16_bf16 (256 clocks of matrix pipe) / one three-way split of 8 values (44 VALU)\n");
  return 0;
}
