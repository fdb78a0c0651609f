// Can v_dot2c_f32_bf16 (gfx950) form the exact residuals of the three-way bf16 split?
//   classic:  h = x & 0xffff0000; r1 = x - h; m = r1 & 0xffff0000; r2 = r1 - m; l = r2          (4 VALU + 1.5 perm / element)
//   dot2c:    hp = perm(x1, x0) (the packed hi pair the MFMA wants anyway); r1_0 = x0 + hp.lo * (-1) + hp.hi * 0; ...
//             (2 dot2c + 1.5 perm / element)
// Part 1: bit-for-bit comparison of the packed planes over random fp32 of every exponent, zeros, denormals, negatives.
// Part 2: cycles per wave64 split of 8 values (one wave per SIMD), both forms, and the raw rate of the instruction.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
typedef __bf16 bf2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ uint32_t pack(uint32_t even, uint32_t odd) { return __builtin_amdgcn_perm(odd, even, 0x07060302u); }

__device__ __forceinline__ void split2_classic(float x0, float x1, uint32_t& hp, uint32_t& mp, uint32_t& lp) {
  const uint32_t h0 = __float_as_uint(x0) & 0xffff0000u, h1 = __float_as_uint(x1) & 0xffff0000u;
  const float r0 = x0 - __uint_as_float(h0), r1 = x1 - __uint_as_float(h1);
  const uint32_t m0 = __float_as_uint(r0) & 0xffff0000u, m1 = __float_as_uint(r1) & 0xffff0000u;
  const float q0 = r0 - __uint_as_float(m0), q1 = r1 - __uint_as_float(m1);
  hp = pack(h0, h1); mp = pack(m0, m1); lp = pack(__float_as_uint(q0), __float_as_uint(q1));
}
// MODE 0: builtin with literal constants (the compiler may fold them into inline constants)
// MODE 1: builtin, constants hidden in VGPRs
// MODE 2: inline asm, constants in VGPRs
template <int MODE>
__device__ __forceinline__ float dot2c(float acc, uint32_t pair, uint32_t cst) {
  if (MODE == 2) {
    asm("v_dot2c_f32_bf16 %0, %1, %2" : "+v"(acc) : "v"(pair), "v"(cst));
    return acc;
  }
  return __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf2, pair), __builtin_bit_cast(bf2, cst), acc, false);
}
template <int MODE>
__device__ __forceinline__ void split2_dot(float x0, float x1, uint32_t& hp, uint32_t& mp, uint32_t& lp) {
  uint32_t clo = 0x0000bf80u, chi = 0xbf800000u;       // (-1, 0) and (0, -1) as (low, high) bf16
  if (MODE >= 1) { asm("" : "+v"(clo)); asm("" : "+v"(chi)); }
  hp = pack(__float_as_uint(x0), __float_as_uint(x1));
  const float r0 = dot2c<MODE>(x0, hp, clo), r1 = dot2c<MODE>(x1, hp, chi);
  mp = pack(__float_as_uint(r0), __float_as_uint(r1));
  const float q0 = dot2c<MODE>(r0, mp, clo), q1 = dot2c<MODE>(r1, mp, chi);
  lp = pack(__float_as_uint(q0), __float_as_uint(q1));
}

__global__ void k_check(const float* x, int n, uint32_t* out) {   // out[v][3][n/2], v = 0 classic, 1..3 dot modes
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (2 * p + 1 >= n) return;
  const float x0 = x[2 * p], x1 = x[2 * p + 1];
  const int np = n / 2;
  uint32_t h, m, l;
  split2_classic(x0, x1, h, m, l); out[(0 * 3 + 0) * np + p] = h; out[(0 * 3 + 1) * np + p] = m; out[(0 * 3 + 2) * np + p] = l;
  split2_dot<0>(x0, x1, h, m, l);  out[(1 * 3 + 0) * np + p] = h; out[(1 * 3 + 1) * np + p] = m; out[(1 * 3 + 2) * np + p] = l;
  split2_dot<1>(x0, x1, h, m, l);  out[(2 * 3 + 0) * np + p] = h; out[(2 * 3 + 1) * np + p] = m; out[(2 * 3 + 2) * np + p] = l;
  split2_dot<2>(x0, x1, h, m, l);  out[(3 * 3 + 0) * np + p] = h; out[(3 * 3 + 1) * np + p] = m; out[(3 * 3 + 2) * np + p] = l;
}

template <int WHAT>
__global__ __launch_bounds__(256) void k_rate(int iters, unsigned long long* cyc, uint32_t* out) {
  float v[8];
  for (int i = 0; i < 8; ++i) v[i] = 1.0f + 1e-3f * (threadIdx.x + 64 * i);
  uint32_t acc = 0;
  uint32_t clo = 0x0000bf80u;
  asm("" : "+v"(clo));
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      if (WHAT == 0) {
#pragma unroll
        for (int p = 0; p < 4; ++p) { uint32_t h, m, l; split2_classic(v[2 * p], v[2 * p + 1], h, m, l); acc ^= h ^ m ^ l; }
      } else if (WHAT == 1) {
#pragma unroll
        for (int p = 0; p < 4; ++p) { uint32_t h, m, l; split2_dot<1>(v[2 * p], v[2 * p + 1], h, m, l); acc ^= h ^ m ^ l; }
      } else {
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = dot2c<1>(v[i], 0x3f803f80u + i, clo);      // 8 independent chains
      }
#pragma unroll
      for (int i = 0; i < 8; ++i) asm volatile("" : "+v"(v[i]));
    }
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  float r = 0; for (int i = 0; i < 8; ++i) r += v[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = acc + __float_as_uint(r);
  if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}
template <int WHAT> double rate(const char* name, double per_iter) {
  unsigned long long* d; uint32_t* o; hipMalloc(&d, 8); hipMalloc(&o, 256 * 256 * 4);
  const int iters = 2000;
  k_rate<WHAT><<<256, 256>>>(iters, d, o); hipDeviceSynchronize();
  k_rate<WHAT><<<256, 256>>>(iters, d, o); hipDeviceSynchronize();
  unsigned long long c; hipMemcpy(&c, d, 8, hipMemcpyDeviceToHost);
  const double v = (double)c / (iters * 4.0 * per_iter);
  printf("%-44s %.2f cycles\n", name, v);
  hipFree(d); hipFree(o);
  return v;
}

int main() {
  std::vector<float> x;
  srand(7);
  auto rnd = []() { return (uint32_t)rand() ^ ((uint32_t)rand() << 15) ^ ((uint32_t)rand() << 30); };
  for (int i = 0; i < (1 << 20); ++i) {                 // every exponent (incl. denormals), random sign + mantissa; no inf / nan
    uint32_t u = rnd();
    if (((u >> 23) & 0xff) == 0xff) u &= ~(1u << 30);
    float f; memcpy(&f, &u, 4); x.push_back(f);
  }
  for (int i = 0; i < (1 << 18); ++i) x.push_back(((rand() % 2001) - 1000) * 1e-3f * (1.0f + (rand() % 1000) * 1e-6f));   // O(1) values
  for (float s : {0.f, -0.f, 1.f, -1.f, 1.17549435e-38f, 1e-39f, 3.4e38f, 1.0000001f, 0.99999994f, 255.99998f}) { x.push_back(s); x.push_back(-s); }
  const int n = (int)x.size() & ~1, np = n / 2;
  float* dx; uint32_t* dout;
  hipMalloc(&dx, n * 4); hipMalloc(&dout, (size_t)12 * np * 4);
  hipMemcpy(dx, x.data(), n * 4, hipMemcpyHostToDevice);
  k_check<<<(np + 255) / 256, 256>>>(dx, n, dout);
  std::vector<uint32_t> o((size_t)12 * np);
  if (hipMemcpy(o.data(), dout, o.size() * 4, hipMemcpyDeviceToHost) != hipSuccess) { printf("kernel failed\n"); return 1; }
  const char* names[4] = {"classic", "dot2c builtin, literal constants", "dot2c builtin, VGPR constants", "dot2c asm, VGPR constants"};
  for (int v = 1; v < 4; ++v) {
    long bad[3] = {0, 0, 0}, bad_normal = 0; int shown = 0;
    for (int p = 0; p < np; ++p)
      for (int pl = 0; pl < 3; ++pl) {
        const uint32_t a = o[(size_t)(0 * 3 + pl) * np + p], b = o[(size_t)(v * 3 + pl) * np + p];
        if (a != b) {
          ++bad[pl];
          const float m = fminf(fabsf(x[2 * p]), fabsf(x[2 * p + 1]));
          if (m > 1e-30f) ++bad_normal;
          if (shown < 4) { printf("   pair %d plane %d: x = %.9g %.9g classic %08x dot %08x\n", p, pl, x[2 * p], x[2 * p + 1], a, b); ++shown; }
        }
      }
    printf("[%s] mismatching packed words: hi %ld mid %ld lo %ld of %d pairs (with both |x| > 1e-30: %ld)\n", names[v], bad[0], bad[1], bad[2], np, bad_normal);
  }
  const double c0 = rate<0>("split of 8 values, classic (44 VALU)", 1.0);
  const double c1 = rate<1>("split of 8 values, dot2c (28 VALU)", 1.0);
  rate<2>("v_dot2c_f32_bf16, per instruction (8 chains)", 8.0);
  printf("speed-up of the split: %.2fx\n", c0 / c1);
  return 0;
}
