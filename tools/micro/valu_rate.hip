// Issue rate of the fp32 VALU forms on gfx950: cycles (s_memtime) per wave64 instruction, one wave per SIMD and two.
//   v_fma_f32, v_pk_fma_f32 (independent chains and one dependent chain), v_sin_f32 (transcendental), v_max_f32
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v2f __attribute__((ext_vector_type(2)));
template <int WHAT>
__global__ __launch_bounds__(512) void k(int iters, unsigned long long* cyc, float* out) {
  float v[8]; v2f p[8];
  for (int i = 0; i < 8; ++i) { v[i] = threadIdx.x + i; p[i] = v2f{(float)threadIdx.x, (float)i}; }
  const float m = 0.999f, c = 1e-3f; const v2f m2 = {m, m}, c2 = {c, c};
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      if (WHAT == 0) { for (int i = 0; i < 8; ++i) v[i] = fmaf(v[i], m, c); }
      if (WHAT == 1) { for (int i = 0; i < 8; ++i) p[i] = __builtin_elementwise_fma(p[i], m2, c2); }
      if (WHAT == 2) { for (int i = 0; i < 8; ++i) p[0] = __builtin_elementwise_fma(p[0], m2, c2); }
      if (WHAT == 3) { for (int i = 0; i < 8; ++i) v[i] = __builtin_amdgcn_sinf(v[i]); }
      if (WHAT == 4) { for (int i = 0; i < 8; ++i) asm volatile("v_max_f32 %0, 0, %0" : "+v"(v[i])); }
      if (WHAT == 5) { for (int i = 0; i < 8; ++i) v[0] = fmaf(v[0], m, c); }
    }
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  float r = 0; for (int i = 0; i < 8; ++i) r += v[i] + p[i].x + p[i].y;
  out[blockIdx.x * blockDim.x + threadIdx.x] = r;
  if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}
template <int WHAT> void run(const char* name, int threads) {
  unsigned long long* d; float* o; hipMalloc(&d, 8); hipMalloc(&o, 1024 * 512 * 4);
  const int iters = 2000;
  k<WHAT><<<256, threads>>>(iters, d, o); hipDeviceSynchronize();
  k<WHAT><<<256, threads>>>(iters, d, o); hipDeviceSynchronize();
  unsigned long long c; hipMemcpy(&c, d, 8, hipMemcpyDeviceToHost);
  printf("%-28s %d waves/SIMD: %.2f cycles per wave instruction\n", name, threads / 256, (double)c / (iters * 32.0));
  hipFree(d); hipFree(o);
}
int main() {
  for (int t : {256, 512}) {
    if (t == 256) { run<0>("v_fma_f32 x8 chains", 256); run<1>("v_pk_fma_f32 x8 chains", 256); run<2>("v_pk_fma_f32 dependent", 256);
                    run<3>("v_sin_f32 x8", 256); run<4>("v_max_f32 x8", 256); run<5>("v_fma_f32 dependent", 256); }
    else { run<0>("v_fma_f32 x8 chains", 512); run<1>("v_pk_fma_f32 x8 chains", 512); run<2>("v_pk_fma_f32 dependent", 512);
           run<3>("v_sin_f32 x8", 512); run<4>("v_max_f32 x8", 512); run<5>("v_fma_f32 dependent", 512); }
  }
  return 0;
}
