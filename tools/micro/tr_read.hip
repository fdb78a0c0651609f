// What exactly does gfx950's ds_read_b64_tr_b16 deliver?  (The lever named in DESIGN 8 for the backward's second dY split:
// keep the bf16 planes in LDS in one orientation, read them as the other MFMA operand.)
// LDS holds u16 element e at byte 2 e with value e.  Every lane issues the read at its own address under three address
// patterns and the four 16-bit values it gets back are printed as element indices -- the lane <-> element map IS the answer.
//   pattern A: lane l reads at byte 8 l                      (64 x 4 row-major: lane = row)
//   pattern B: 16-lane group g, lane i: byte 128 g + 32 (i / 4) + 8 (i % 4)   ([4 rows][16 cols] block per group, 32-byte rows)
//   pattern C: 16-lane group g, lane i: byte 512 g + 32 i                      ([16 rows][16 cols] per group, lane = row, first 4 cols)
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <vector>
typedef unsigned v2u __attribute__((ext_vector_type(2)));

__global__ void k_tr(int pattern, uint16_t* out) {
  __shared__ __attribute__((aligned(16))) uint16_t lds[4096];
  const int l = threadIdx.x;
  for (int e = l; e < 4096; e += 64) lds[e] = (uint16_t)e;
  __syncthreads();
  const int g = l >> 4, i = l & 15;
  unsigned addr = 0;
  if (pattern == 0) addr = 8u * l;
  else if (pattern == 1) addr = 128u * g + 32u * (i / 4) + 8u * (i % 4);
  else addr = 512u * g + 32u * i;
  addr += (unsigned)(uintptr_t)lds;          // LDS byte address (generic -> local offset: the low 32 bits on gfx9)
  v2u v;
  asm volatile("ds_read_b64_tr_b16 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr) : "memory");
  out[4 * l] = (uint16_t)(v.x & 0xffff); out[4 * l + 1] = (uint16_t)(v.x >> 16);
  out[4 * l + 2] = (uint16_t)(v.y & 0xffff); out[4 * l + 3] = (uint16_t)(v.y >> 16);
}

int main() {
  uint16_t* d;
  if (hipMalloc(&d, 64 * 4 * 2) != hipSuccess) { printf("no device\n"); return 1; }
  std::vector<uint16_t> h(256);
  const char* names[3] = {"A: byte 8 l", "B: 128 g + 32 (i / 4) + 8 (i % 4)", "C: 512 g + 32 i"};
  for (int p = 0; p < 3; ++p) {
    hipLaunchKernelGGL(k_tr, dim3(1), dim3(64), 0, 0, p, d);
    if (hipMemcpy(h.data(), d, 512, hipMemcpyDeviceToHost) != hipSuccess) { printf("copy failed\n"); return 1; }
    printf("pattern %s   (lane: the element indices of its four 16-bit values, low half of register 0 first)\n", names[p]);
    for (int l = 0; l < 64; ++l)
      printf("  lane %2d: %4d %4d %4d %4d%s", l, h[4 * l], h[4 * l + 1], h[4 * l + 2], h[4 * l + 3], (l & 3) == 3 ? "\n" : " |");
  }
  (void)hipFree(d);
  return 0;
}
