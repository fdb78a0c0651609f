// What would a bf16 split of the fp32 MLP tiles buy, and what would it cost in accuracy?  
// One wave computes C[32x32] = A[32x64] B[64x32] three ways:
//   f32    : 32 v_mfma_f32_32x32x2_f32                               (what the kernels do)
//   bf16x3 : x = hi + mid + lo (three bf16), 6 products hi*hi, hi*mid, mid*hi, hi*lo, lo*hi, mid*mid on
//            v_mfma_f32_32x32x16_bf16 = 24 MFMAs of half the duration, plus the VALU work of splitting B (the
//            activations; A = weights would be split once per workgroup)
//   bf16x1 : plain bf16 (1 product) for scale
// Prints shader-clock cycles per tile (MFMA part / split part) and the max abs error against an fp64 reference.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ void split3(float x, __bf16& h, __bf16& m, __bf16& l) {
  h = (__bf16)x; const float r1 = x - (float)h;
  m = (__bf16)r1; const float r2 = r1 - (float)m;
  l = (__bf16)r2;
}

// A: [32][64] row-major, B: [64][32] row-major (k-major), C: [32][32]
__global__ __launch_bounds__(64) void k(const float* A, const float* B, float* Cf, float* C3, float* C1, unsigned long long* cyc, int iters) {
  const int lane = threadIdx.x, i = lane & 31, kh = lane >> 5;
  // ---- fp32 path: k-step s uses A[i][2s + kh], B[2s + kh][i]
  float a32[32], b32[32];
  for (int s = 0; s < 32; ++s) { a32[s] = A[i * 64 + 2 * s + kh]; b32[s] = B[(2 * s + kh) * 32 + i]; }
  f32x16 acc = {0};
  unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int s = 0; s < 32; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a32[s], b32[s], acc, 0, 0, 0);
  }
  unsigned long long t1 = __builtin_readcyclecounter();
  {
    f32x16 one = {0};                      // accuracy from ONE pass (the timed loop accumulates iters passes)
#pragma unroll
    for (int s = 0; s < 32; ++s) one = __builtin_amdgcn_mfma_f32_32x32x2f32(a32[s], b32[s], one, 0, 0, 0);
    for (int r = 0; r < 16; ++r) Cf[((r & 3) + 8 * (r >> 2) + 4 * kh) * 32 + i] = one[r] + 0.f * acc[r];
  }
  // ---- bf16 paths: K block kb (16 k's): lane (i, kh) holds k = 16 kb + 8 kh + 0..7
  bf16x8 ah[4], am[4], al[4], bh[4], bm[4], bl[4];
  float bf[4][8];
  for (int kb = 0; kb < 4; ++kb)
    for (int e = 0; e < 8; ++e) {
      const int kk = 16 * kb + 8 * kh + e;
      __bf16 h, m, l;
      split3(A[i * 64 + kk], h, m, l); ah[kb][e] = h; am[kb][e] = m; al[kb][e] = l;
      bf[kb][e] = B[kk * 32 + i];
    }
  // split of B timed separately (it is per-sample work in the real kernel)
  unsigned long long t2 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int kb = 0; kb < 4; ++kb)
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        __bf16 h, m, l;
        split3(bf[kb][e], h, m, l); bh[kb][e] = h; bm[kb][e] = m; bl[kb][e] = l;
        asm volatile("" : "+v"(bf[kb][e]));        // the next iteration's split cannot be hoisted or merged
      }
  }
  unsigned long long t3 = __builtin_readcyclecounter();
  f32x16 c3 = {0}, c1 = {0};
  unsigned long long t4 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int kb = 0; kb < 4; ++kb) {
      c3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[kb], bh[kb], c3, 0, 0, 0);     // small terms first
      c3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[kb], bl[kb], c3, 0, 0, 0);
      c3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am[kb], bm[kb], c3, 0, 0, 0);
      c3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am[kb], bh[kb], c3, 0, 0, 0);
      c3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[kb], bm[kb], c3, 0, 0, 0);
      c3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[kb], bh[kb], c3, 0, 0, 0);
    }
  }
  unsigned long long t5 = __builtin_readcyclecounter();
  for (int kb = 0; kb < 4; ++kb) c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[kb], bh[kb], c1, 0, 0, 0);
  f32x16 o3 = {0};
  for (int kb = 0; kb < 4; ++kb) {
    o3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[kb], bh[kb], o3, 0, 0, 0);
    o3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[kb], bl[kb], o3, 0, 0, 0);
    o3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am[kb], bm[kb], o3, 0, 0, 0);
    o3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am[kb], bh[kb], o3, 0, 0, 0);
    o3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[kb], bm[kb], o3, 0, 0, 0);
    o3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[kb], bh[kb], o3, 0, 0, 0);
  }
  for (int r = 0; r < 16; ++r) {
    C3[((r & 3) + 8 * (r >> 2) + 4 * kh) * 32 + i] = o3[r] + 0.f * c3[r];
    C1[((r & 3) + 8 * (r >> 2) + 4 * kh) * 32 + i] = c1[r];
  }
  if (lane == 0) { cyc[0] = t1 - t0; cyc[1] = t3 - t2; cyc[2] = t5 - t4; }
}

int main() {
  const int iters = 200;
  std::vector<float> A(32 * 64), B(64 * 32), Cf(1024), C3(1024), C1(1024);
  unsigned long long st = 88172645463325252ull;
  auto rnd = [&]() { st ^= st << 13; st ^= st >> 7; st ^= st << 17; return (float)((double)(st >> 11) / (double)(1ull << 53) * 2.0 - 1.0); };
  for (auto& v : A) v = rnd() * 0.3f;          // weights
  for (auto& v : B) v = rnd() * 1.5f;          // activations
  float *dA, *dB, *dCf, *dC3, *dC1; unsigned long long* dc;
  hipMalloc(&dA, A.size() * 4); hipMalloc(&dB, B.size() * 4); hipMalloc(&dCf, 4096); hipMalloc(&dC3, 4096); hipMalloc(&dC1, 4096); hipMalloc(&dc, 64);
  hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);
  k<<<1, 64>>>(dA, dB, dCf, dC3, dC1, dc, iters); hipDeviceSynchronize();
  k<<<1, 64>>>(dA, dB, dCf, dC3, dC1, dc, iters); hipDeviceSynchronize();
  unsigned long long c[3];
  hipMemcpy(c, dc, 24, hipMemcpyDeviceToHost);
  hipMemcpy(Cf.data(), dCf, 4096, hipMemcpyDeviceToHost); hipMemcpy(C3.data(), dC3, 4096, hipMemcpyDeviceToHost); hipMemcpy(C1.data(), dC1, 4096, hipMemcpyDeviceToHost);
  double ef = 0, e3 = 0, e1 = 0, mx = 0;
  for (int i = 0; i < 32; ++i)
    for (int j = 0; j < 32; ++j) {
      double r = 0;
      for (int kk = 0; kk < 64; ++kk) r += (double)A[i * 64 + kk] * (double)B[kk * 32 + j];
      mx = fmax(mx, fabs(r));
      ef = fmax(ef, fabs(Cf[i * 32 + j] - r)); e3 = fmax(e3, fabs(C3[i * 32 + j] - r)); e1 = fmax(e1, fabs(C1[i * 32 + j] - r));
    }
  printf("cycles per 32x32x64 tile (s_memtime): f32 MFMA %.0f | bf16x3: 24 MFMA %.0f + split of the 64x32 activations %.0f | max|C| %.2f\n",
         (double)c[0] / iters, (double)c[2] / iters, (double)c[1] / iters, mx);
  printf("max abs error vs fp64: f32 MFMA %.3e   bf16x3 %.3e   plain bf16 %.3e\n", ef, e3, e1);
  return 0;
}
