// What does a random 8-byte gather cost on gfx950, by where the table lives and how the load is flagged?
// The hash-encoded forward issues 64 such gathers per sample (16 levels x 4 simplex vertices x float2); the fine
// levels (9 of 16) touch a different 128-byte line per lane.  Whole chip, 8 waves per CU, every lane draws its own
// pseudo-random entry of a table of `entries` float2:
//   mode 0  global_load_dwordx2                (default cache policy)
//   mode 1  global_load_dwordx2 nt             (non-temporal)
//   mode 2  global_load_dwordx2 sc0 sc1        (system coherent: bypasses the vector L1)
//   mode 3  global_load_dwordx2 sc1
//   mode 4  ds_read_b64 from a copy of the table in LDS (entries <= 16384)
//   mode 5  global_load_dword x2 (two 4-byte loads, e.g. split feature planes) -- for comparison
// Reported: clocks per wave64 gather instruction per CU (8 waves resident), and GB/s of useful bytes chip-wide.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

template <int MODE>
__global__ __launch_bounds__(512) void k(const float2* __restrict__ tab, uint32_t mask, int iters, unsigned long long* cyc, float* out) {
  extern __shared__ float2 lt[];
  if (MODE == 4) {
    for (uint32_t i = threadIdx.x; i <= mask; i += blockDim.x) lt[i] = tab[i];
    __syncthreads();
  }
  uint32_t s = (blockIdx.x * 512u + threadIdx.x) * 2654435761u + 12345u;
  float ax = 0.f, ay = 0.f;
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
    uint32_t idx[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) { s = s * 1664525u + 1013904223u; idx[u] = (s >> 9) & mask; }
    float2 v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const float2* p = tab + idx[u];
      if (MODE == 0) v[u] = *p;
      else if (MODE == 1) { typedef float v2 __attribute__((ext_vector_type(2))); const v2 t = __builtin_nontemporal_load(reinterpret_cast<const v2*>(p)); v[u] = make_float2(t.x, t.y); }
      else if (MODE == 2) { float2 t; asm volatile("global_load_dwordx2 %0, %1, off sc0 sc1" : "=v"(t) : "v"(p) : "memory"); v[u] = t; }
      else if (MODE == 3) { float2 t; asm volatile("global_load_dwordx2 %0, %1, off sc1" : "=v"(t) : "v"(p) : "memory"); v[u] = t; }
      else if (MODE == 4) v[u] = lt[idx[u]];
      else { const float* q = reinterpret_cast<const float*>(tab); v[u] = make_float2(q[idx[u]], q[mask + 1 + idx[u]]); }
    }
    if (MODE == 2 || MODE == 3) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int u = 0; u < 8; ++u) { ax += v[u].x; ay += v[u].y; }
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  out[blockIdx.x * 512 + threadIdx.x] = ax + ay;
  if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

template <int MODE> void run(const char* name, const float2* tab, uint32_t entries) {
  unsigned long long* d; float* o;
  (void)hipMalloc(&d, 8); (void)hipMalloc(&o, 256 * 512 * 4);
  const int iters = 400;
  const size_t lds = MODE == 4 ? (size_t)entries * 8 : 0;
  if (MODE == 4 && lds > 150 * 1024) { (void)hipFree(d); (void)hipFree(o); return; }
  (void)hipFuncSetAttribute((const void*)k<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  k<MODE><<<256, 512, lds>>>(tab, entries - 1, iters, d, o); (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0);
  k<MODE><<<256, 512, lds>>>(tab, entries - 1, iters, d, o);
  (void)hipEventRecord(e1); (void)hipDeviceSynchronize();
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  unsigned long long c; (void)hipMemcpy(&c, d, 8, hipMemcpyDeviceToHost);
  // per CU: 8 waves x iters x 8 gather instructions in c clocks
  printf("  %-36s %7.1f clocks per wave gather per CU   %7.1f GB/s useful  (%.3f ms)\n", name, (double)c / (iters * 8.0 * 8.0),
         256.0 * 512 * iters * 8 * 8 / (ms * 1e6), ms);
  (void)hipFree(d); (void)hipFree(o);
}

int main() {
  for (uint32_t entries : {4096u, 16384u, 65536u, 1u << 20}) {
    std::vector<float2> h(2 * entries);
    for (auto& v : h) v = make_float2(rand() * 1e-9f, rand() * 1e-9f);
    float2* tab; (void)hipMalloc(&tab, h.size() * 8); (void)hipMemcpy(tab, h.data(), h.size() * 8, hipMemcpyHostToDevice);
    printf("table of %u float2 entries (%u KB):\n", entries, entries * 8 / 1024);
    run<0>("global_load_dwordx2", tab, entries);
    run<1>("global_load_dwordx2 nt", tab, entries);
    run<2>("global_load_dwordx2 sc0 sc1", tab, entries);
    run<3>("global_load_dwordx2 sc1", tab, entries);
    run<4>("ds_read_b64 (LDS copy)", tab, entries);
    run<5>("2 x global_load_dword (planes)", tab, entries);
    (void)hipFree(tab);
  }
  return 0;
}
