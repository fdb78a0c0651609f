// Back-to-back issue rate of the fp32 MFMA shapes on one SIMD (one wave per SIMD, 4 independent accumulators).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x32 __attribute__((ext_vector_type(32)));

template <int KIND>
__global__ __launch_bounds__(256) void k(int iters, float* out) {
  float x = threadIdx.x * 1e-3f, y = 1.0f + x, r = 0.f;
  if (KIND == 0) {          // 16x16x4: 2048 flop
    f32x4 a0 = {0, 0, 0, 0}, a1 = a0, a2 = a0, a3 = a0;
    for (int i = 0; i < iters; ++i) {
      a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a0, 0, 0, 0); a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a1, 0, 0, 0);
      a2 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a2, 0, 0, 0); a3 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a3, 0, 0, 0);
    }
    r = a0[0] + a1[1] + a2[2] + a3[3];
  } else if (KIND == 1) {   // 32x32x2: 4096 flop
    f32x16 a0 = {}, a1 = {}, a2 = {}, a3 = {};
    for (int i = 0; i < iters; ++i) {
      a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a0, 0, 0, 0); a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a1, 0, 0, 0);
      a2 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a2, 0, 0, 0); a3 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a3, 0, 0, 0);
    }
    r = a0[0] + a1[1] + a2[2] + a3[3];
  } else if (KIND == 2) {   // 32x32x1, 2 blocks: 4096 flop
    f32x32 a0 = {}, a1 = {};
    for (int i = 0; i < iters; ++i) {
      a0 = __builtin_amdgcn_mfma_f32_32x32x1f32(x, y, a0, 0, 0, 0); a1 = __builtin_amdgcn_mfma_f32_32x32x1f32(x, y, a1, 0, 0, 0);
      a0 = __builtin_amdgcn_mfma_f32_32x32x1f32(y, x, a0, 0, 0, 0); a1 = __builtin_amdgcn_mfma_f32_32x32x1f32(y, x, a1, 0, 0, 0);
    }
    r = a0[0] + a1[1];
  } else if (KIND == 3) {   // 16x16x1, 4 blocks: 2048 flop
    f32x16 a0 = {}, a1 = {}, a2 = {}, a3 = {};
    for (int i = 0; i < iters; ++i) {
      a0 = __builtin_amdgcn_mfma_f32_16x16x1f32(x, y, a0, 0, 0, 0); a1 = __builtin_amdgcn_mfma_f32_16x16x1f32(x, y, a1, 0, 0, 0);
      a2 = __builtin_amdgcn_mfma_f32_16x16x1f32(x, y, a2, 0, 0, 0); a3 = __builtin_amdgcn_mfma_f32_16x16x1f32(x, y, a3, 0, 0, 0);
    }
    r = a0[0] + a1[1] + a2[2] + a3[3];
  } else {                  // 4x4x1, 16 blocks: 512 flop
    f32x4 a0 = {0, 0, 0, 0}, a1 = a0, a2 = a0, a3 = a0;
    for (int i = 0; i < iters; ++i) {
      a0 = __builtin_amdgcn_mfma_f32_4x4x1f32(x, y, a0, 0, 0, 0); a1 = __builtin_amdgcn_mfma_f32_4x4x1f32(x, y, a1, 0, 0, 0);
      a2 = __builtin_amdgcn_mfma_f32_4x4x1f32(x, y, a2, 0, 0, 0); a3 = __builtin_amdgcn_mfma_f32_4x4x1f32(x, y, a3, 0, 0, 0);
    }
    r = a0[0] + a1[1] + a2[2] + a3[3];
  }
  if (r == 12345.678f) out[threadIdx.x] = r;
}

template <int KIND>
void run(const char* name, double flop_per_mfma, float* out) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int iters = 20000;
  k<KIND><<<256, 256>>>(100, out);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  k<KIND><<<256, 256>>>(iters, out);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double mfmas = 4.0 * iters;                      // per wave (= per SIMD: 4 waves per CU, one per SIMD)
  const double tf = mfmas * flop_per_mfma * 256 * 4 / (ms * 1e-3) / 1e12;
  printf("%-22s %.3f ms  %.1f cycles/MFMA @2.4GHz  %.1f TFLOP/s chip\n", name, ms, ms * 1e-3 * 2.4e9 / mfmas, tf);
}

int main() {
  float* out; (void)hipMalloc(&out, 4096);
  run<0>("v_mfma_f32_16x16x4", 2048, out);
  run<1>("v_mfma_f32_32x32x2", 4096, out);
  run<2>("v_mfma_f32_32x32x1_2b", 4096, out);
  run<3>("v_mfma_f32_16x16x1_4b", 2048, out);
  run<4>("v_mfma_f32_4x4x1_16b", 512, out);
  return 0;
}
