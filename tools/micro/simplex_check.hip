// Old against new simplex search of the permutohedral hash encoding (ngm_field.h permuto_simplex): bit-for-bit comparison of
// the four table indices and the four barycentric weights on random and on structured (tie-provoking) points, and the
// time of each.   hipcc --offload-arch=gfx950 -O3 tools/micro/simplex_check.hip -o /tmp/simplex_check && /tmp/simplex_check
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
__device__ __forceinline__ void simplex_ref(float x, float y, float z, const float* lp, uint32_t mask, uint32_t (&idx)[4], float (&bw)[4]) {
#pragma clang fp contract(off)
  const float c0 = (x + lp[4]) * lp[0], c1 = (y + lp[5]) * lp[1], c2 = (z + lp[6]) * lp[2];
  float el[4];
  float sm = 0.f;
  { const float t3 = 3.0f * c2; el[3] = sm - t3; sm = sm + c2; }
  { const float t2 = 2.0f * c1; el[2] = sm - t2; sm = sm + c1; }
  { const float t1 = 1.0f * c0; el[1] = sm - t1; sm = sm + c0; }
  el[0] = sm;
  int rem0[4], sum = 0;
  float diff[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const float v = el[i] * 0.25f;
    const float up = ceilf(v) * 4.0f, down = floorf(v) * 4.0f;
    const float r = ((up - el[i]) < (el[i] - down)) ? up : down;
    rem0[i] = (int)r;
    diff[i] = el[i] - r;
    sum += rem0[i];
  }
  sum /= 4;
  int rank[4] = {0, 0, 0, 0};
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = i + 1; j < 4; ++j) {
      const bool lt = diff[i] < diff[j];
      rank[i] += lt ? 1 : 0;
      rank[j] += lt ? 0 : 1;
    }
  float delta[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    rank[i] += sum;
    if (rank[i] < 0) { rank[i] += 4; rem0[i] += 4; }
    else if (rank[i] > 3) { rank[i] -= 4; rem0[i] -= 4; }
    delta[i] = (el[i] - (float)rem0[i]) * 0.25f;
  }
  constexpr uint32_t P1 = 2531011u, P2 = 2220443785u, P3 = 2937900635u;
  float d[4]; uint32_t q[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const bool e0 = rank[0] == k, e1 = rank[1] == k, e2 = rank[2] == k;
    d[k] = e0 ? delta[0] : e1 ? delta[1] : e2 ? delta[2] : delta[3];
    q[k] = e0 ? 4u * P3 : e1 ? 4u * P2 : e2 ? 4u * P1 : 0u;
  }
  bw[0] = d[3] + (1.0f + (0.f - d[0]));
  bw[1] = d[2] - d[3];
  bw[2] = d[1] - d[2];
  bw[3] = d[0] - d[1];
  constexpr uint32_t C = P1 + P2 + P3;
  uint32_t h = (uint32_t)rem0[0] * P3 + (uint32_t)rem0[1] * P2 + (uint32_t)rem0[2] * P1;
  idx[0] = h & mask;
  h += C - q[3]; idx[1] = h & mask;
  h += C - q[2]; idx[2] = h & mask;
  h += C - q[1]; idx[3] = h & mask;
}
__device__ __forceinline__ void simplex_new(float x, float y, float z, const float* lp, uint32_t mask, uint32_t (&idx)[4], float (&bw)[4]) {
#pragma clang fp contract(off)
  const float c0 = (x + lp[4]) * lp[0], c1 = (y + lp[5]) * lp[1], c2 = (z + lp[6]) * lp[2];
  float el[4];
  float sm = 0.f;
  { const float t3 = 3.0f * c2; el[3] = sm - t3; sm = sm + c2; }
  { const float t2 = 2.0f * c1; el[2] = sm - t2; sm = sm + c1; }
  { const float t1 = 1.0f * c0; el[1] = sm - t1; sm = sm + c0; }
  el[0] = sm;
  // nearest multiple of 4 below-or-at the midpoint: k = round-half-down(el / 4); t = k - el / 4 in [-0.5, 0.5)
  int k[4]; float t[4];
  int sum = 0;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const float v = el[i] * 0.25f;
    const float r = __builtin_rintf(v);
    const float tt = r - v;                       // exact
    const bool tie = tt == 0.5f;                  // rint went up at an exact tie: the reference's comparison goes down
    t[i] = tie ? -0.5f : tt;
    k[i] = (int)r - (tie ? 1 : 0);
    sum += k[i];
  }
  // rank_i = #{j : diff_i < diff_j} with index tie-break, diff = -4 t
  const int l01 = t[0] > t[1], l02 = t[0] > t[2], l03 = t[0] > t[3], l12 = t[1] > t[2], l13 = t[1] > t[3], l23 = t[2] > t[3];
  int rank[4];
  rank[0] = sum + l01 + l02 + l03;
  rank[1] = sum + 1 - l01 + l12 + l13;
  rank[2] = sum + 2 - l02 - l12 + l23;
  rank[3] = sum + 3 - l03 - l13 - l23;
  float delta[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int adj = rank[i] >> 2;                 // -1, 0, 1: the wrap-around of the reference's two branches
    rank[i] &= 3;
    k[i] -= adj;
    delta[i] = (float)adj - t[i];                 // = (el - 4 k) / 4 with one rounding, as the reference
  }
  constexpr uint32_t P1 = 2531011u, P2 = 2220443785u, P3 = 2937900635u;
  float d[4]; uint32_t q[4];
#pragma unroll
  for (int kk = 0; kk < 4; ++kk) {
    const bool e0 = rank[0] == kk, e1 = rank[1] == kk, e2 = rank[2] == kk;
    d[kk] = e0 ? delta[0] : e1 ? delta[1] : e2 ? delta[2] : delta[3];
    q[kk] = e0 ? 4u * P3 : e1 ? 4u * P2 : e2 ? 4u * P1 : 0u;
  }
  bw[0] = d[3] + (1.0f + (0.f - d[0]));
  bw[1] = d[2] - d[3];
  bw[2] = d[1] - d[2];
  bw[3] = d[0] - d[1];
  constexpr uint32_t C = P1 + P2 + P3;
  uint32_t h = ((uint32_t)k[0] * P3 + (uint32_t)k[1] * P2 + (uint32_t)k[2] * P1) * 4u;
  idx[0] = h & mask;
  h += C - q[3]; idx[1] = h & mask;
  h += C - q[2]; idx[2] = h & mask;
  h += C - q[1]; idx[3] = h & mask;
}

__device__ __forceinline__ uint32_t mix(uint32_t x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }
// mode 0: uniform in [-1, 1]^3 (the fields' unit cube); 1: multiples of 1/8 in [-4, 4] (exact ties of the rounding and of the
// ranking); 2: wide range +-1000; 3: tiny values around 0 (+-1e-6)
__device__ __forceinline__ float coord(uint32_t h, int mode) {
  const float u = (float)(h >> 8) * (1.0f / 16777216.0f);
  if (mode == 0) return 2.0f * u - 1.0f;
  if (mode == 1) return (float)((int)(h % 65u) - 32) * 0.125f;
  if (mode == 2) return (2.0f * u - 1.0f) * 1000.0f;
  return (2.0f * u - 1.0f) * 1e-6f;
}
__global__ void k_check(int mode, int level, unsigned long long base, unsigned long long* bad, float4* firstbad) {
  const unsigned long long t = base + blockIdx.x * (unsigned long long)blockDim.x + threadIdx.x;
  const uint32_t s = mix((uint32_t)t) ^ mix((uint32_t)(t >> 32) + 0x9e3779b9u);
  const float x = coord(mix(s + 1), mode), y = coord(mix(s + 2), mode), z = coord(mix(s + 3), mode);
  // level parameters as ngm_permuto_fill_scales lays them out: scale factors (3), -, shifts (3), -
  float lp[8];
  const float sc = (mode == 1) ? 1.0f : exp2f(0.8f * level);   // mode 1: scale factors 1, 1/2, 1/4-type values keep the ties exact
  lp[0] = sc / sqrtf(2.0f); lp[1] = sc / sqrtf(6.0f); lp[2] = sc / sqrtf(12.0f); lp[3] = 0.f;
  if (mode == 1) { lp[0] = 1.0f; lp[1] = 0.5f; lp[2] = 0.25f; }
  lp[4] = (mode == 1) ? 0.f : 0.123f * level; lp[5] = (mode == 1) ? 0.f : -0.377f * level; lp[6] = (mode == 1) ? 0.f : 0.911f * level; lp[7] = 0.f;
  uint32_t ia[4], ib[4]; float wa[4], wb[4];
  simplex_ref(x, y, z, lp, 4095u, ia, wa);
  simplex_new(x, y, z, lp, 4095u, ib, wb);
  bool same = true;
  for (int r = 0; r < 4; ++r) same = same && ia[r] == ib[r] && __float_as_uint(wa[r]) == __float_as_uint(wb[r]);
  if (!same) { if (atomicAdd(bad, 1ull) == 0) *firstbad = make_float4(x, y, z, (float)level); }
}
template <int WHICH>
__global__ void k_time(const float* lpg, float* out, int iters) {
  float x = threadIdx.x * 1e-3f, y = blockIdx.x * 1e-4f, z = 0.5f;
  float acc = 0.f; uint32_t hacc = 0;
  float lp[8];
  for (int i = 0; i < 8; ++i) lp[i] = lpg[i];
  for (int i = 0; i < iters; ++i) {
    uint32_t idx[4]; float bw[4];
    if (WHICH == 0) simplex_ref(x, y, z, lp, 4095u, idx, bw); else simplex_new(x, y, z, lp, 4095u, idx, bw);
    acc += bw[0] + bw[1] * 2.f + bw[2] * 3.f + bw[3] * 4.f; hacc ^= idx[0] ^ (idx[1] << 1) ^ (idx[2] << 2) ^ (idx[3] << 3);
    x += 0.37f * bw[1]; y -= 0.11f * bw[2]; z += 0.05f;
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = acc + (float)hacc;
}
int main() {
  unsigned long long* bad; float4* fb;
  (void)hipMalloc(&bad, 8); (void)hipMalloc(&fb, 16);
  unsigned long long total = 0, nbad_all = 0;
  for (int mode = 0; mode < 4; ++mode) {
    unsigned long long nb = 0;
    (void)hipMemset(bad, 0, 8);
    for (int level = 0; level < 16; ++level)
      for (int rep = 0; rep < 4; ++rep) {
        k_check<<<16384, 256>>>(mode, level, (unsigned long long)(mode * 64 + level * 4 + rep) << 24, bad, fb);
        total += 16384ull * 256ull;
      }
    (void)hipDeviceSynchronize();
    (void)hipMemcpy(&nb, bad, 8, hipMemcpyDeviceToHost);
    float4 f; (void)hipMemcpy(&f, fb, 16, hipMemcpyDeviceToHost);
    printf("mode %d: %llu mismatches of %llu points", mode, nb, 16ull * 4 * 16384 * 256);
    if (nb) printf("  first: (%.9g, %.9g, %.9g) level %g", f.x, f.y, f.z, f.w);
    printf("\n");
    nbad_all += nb;
  }
  float* lpg; float* out;
  float lph[8] = {3.1f, 1.7f, 1.2f, 0.f, 0.1f, -0.3f, 0.7f, 0.f};
  (void)hipMalloc(&lpg, 32); (void)hipMemcpy(lpg, lph, 32, hipMemcpyHostToDevice);
  (void)hipMalloc(&out, 1024 * 256 * 4);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  for (int which = 0; which < 2; ++which) {
    float best = 1e9f;
    for (int rep = 0; rep < 3; ++rep) {
      (void)hipEventRecord(e0);
      if (which == 0) k_time<0><<<1024, 256>>>(lpg, out, 2000); else k_time<1><<<1024, 256>>>(lpg, out, 2000);
      (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
      float ms; (void)hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
    }
    // 1024 x 4 waves on 1024 SIMDs: four waves per SIMD
    printf("%s: %.3f ms for 2000 calls per lane -> %.0f ns per call and wave\n", which ? "new" : "old", best, best * 1e6f / 2000.f);
  }
  printf("total points %llu, mismatches %llu\n", total, nbad_all);
  return nbad_all ? 1 : 0;
}
