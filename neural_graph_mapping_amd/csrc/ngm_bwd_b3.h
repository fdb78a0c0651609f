// Shared pieces of the backward kernels on the bf16 matrix pipe (ngm_field_bwd_b3.hip: one wave per tile;
// ngm_field_bwd_b3q.hip: a pair of waves per tile, both on one SIMD): tile layout in LDS, DMA issue, three-way splits, MFMA blocks.
#pragma once
#include "ngm_bwd16.h"

// -DNGM_DMAWAIT_TIMING: only the clocks spent in the tile-start DMA wait (slot 2) and the kernel total (slot 12)
#ifdef NGM_DMAWAIT_TIMING
#undef TICK_DECL
#undef TICK
#undef TICK_REPORT
#define TICK_DECL unsigned long long tw_ = 0, tl_ = 0; const unsigned long long ts_ = __builtin_readcyclecounter()
#define TICK(k)                                                                  \
  do {                                                                           \
    if ((k) == 10) tl_ = __builtin_readcyclecounter();                           \
    if ((k) == 2) tw_ += __builtin_readcyclecounter() - tl_;                     \
  } while (0)
#define TICK_REPORT                                                                                \
  if (a.debug_cycles && blockIdx.x == gridDim.x / 2 && threadIdx.x == 64) {                        \
    for (int k = 0; k < 13; ++k) a.debug_cycles[k] = 0;                                            \
    a.debug_cycles[2] = tw_; a.debug_cycles[12] = __builtin_readcyclecounter() - ts_;              \
  }
#endif
#define B3B_WAVES 4
#define B3B_THREADS 256
#define HT 2048                 // floats of one 32 x 64 activation tile
#define PLANE_G 512             // 16-byte granules of one weight plane: [nt 2][kb 4][kh 2][n 32]

template <int L, bool EG>
struct LdsB3b {
  static constexpr int NPL = (L - 1) + (EG ? 1 : 0);                 // layers whose data gradient is needed
  static constexpr int plane_slot(int l) { return EG ? l : l - 1; }  // 16-byte units: slot * 3 * PLANE_G
  static constexpr int PLANES = NPL * 3 * PLANE_G * 4 + 512 + 768 + 512;   // floats; + per-feature constants: float4 wout[64], enc[64], uint2 woutp[3][2][64], u32x4 id[2][64]
  static constexpr int CONSTS = NPL * 3 * PLANE_G * 4;
  // per wave: the output layer's input tile, (L = 2) layer 1's input tile, the input landing buffer, points, d_out.
  // Single buffers: the next tile's transfers are issued when all of them are free (see the tile loop).
  static constexpr int HL = 0;
  static constexpr int H1 = HT;
  static constexpr int INB = H1 + ((L == 2) ? HT : 0);               // float4 [2 halves][4 pieces][16] + (fused compositing) [2 pieces][32]
  static constexpr int PB = INB + 768;                               // float4 [32]
  static constexpr int OB = PB + 128;                                // float4 [32]
  // fused compositing (FC): the inputs of the tile after next, and the position / gradient rows of the next tile (one 64-lane
  // pass computes two tiles' worth)
  static constexpr int INB2 = OB + 128;
  static constexpr int PB2 = INB2 + 768;
  static constexpr int OB2 = PB2 + 128;
  static constexpr int WAVE_TOTAL = OB2 + 128;
  static constexpr int NT = 4 * L;                                   // 32x32 accumulator tiles per wave
  static constexpr int EPI = B3B_WAVES * NT * 1024;
  static constexpr int BODY = PLANES + B3B_WAVES * WAVE_TOTAL;
  static constexpr int TOTAL = BODY > EPI ? BODY : EPI;
};

// one activation tile, samples [n0, n0 + 32) of the field (clamped to end - 1): 8 DMA instructions of two chunks each
__device__ __forceinline__ void issue_tile32(const char* sbase, uint32_t gb, uint32_t n0, uint32_t end, int lane, uint32_t lds_tile) {
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const uint32_t c = 2 * k + (lane >> 5);
    uint32_t n = n0 + (((uint32_t)lane & 31u) ^ (c & 7u));
    if (n >= end) n = end - 1;
    const uint32_t u = n + gb;
    dma16_so(sbase, (((u >> 5) * 16u + c) * 32u + ((u & 31u) ^ (c & 7u))) * 16u, lds_tile + k * 1024);
  }
}

// Fast form of issue_tile32 for a tile that is whole and aligned with the stash's 32-sample tiles (wave-uniform test by
// the caller): the stash tile already IS the LDS image (ActStash, ngm_field.h), so the eight transfers are linear copies
// -- lane p moves bytes [1024 k + 16 p, + 16) -- with scalar addressing and one M0 write per four (the instruction's
// immediate offset is added to the LDS address and to the global address alike).
__device__ __forceinline__ void dma16_x4(const char* sb, uint32_t voff, uint32_t lds_base) {
  uint32_t keep;
  asm volatile(
      "s_mov_b32 %0, m0\n\t"
      "s_mov_b32 m0, %3\n\t"
      "s_nop 0\n\t"
      "global_load_lds_dwordx4 %1, %2 nt\n\t"
      "global_load_lds_dwordx4 %1, %2 offset:1024 nt\n\t"
      "global_load_lds_dwordx4 %1, %2 offset:2048 nt\n\t"
      "global_load_lds_dwordx4 %1, %2 offset:3072 nt\n\t"
      "s_mov_b32 m0, %0"
      : "=&s"(keep)
      : "v"(voff), "s"(sb), "s"(lds_base)
      : "memory");
}
__device__ __forceinline__ void issue_tile32_fast(const char* sbase, uint32_t tile, int lane, uint32_t lds_tile) {
  const char* b0 = sbase + (size_t)__builtin_amdgcn_readfirstlane(tile) * 8192;     // wave-uniform by construction: SGPR pair
  dma16_x4(b0, (uint32_t)lane * 16u, lds_tile);
  dma16_x4(b0 + 4096, (uint32_t)lane * 16u, lds_tile + 4096);
}

struct B3Op { ngm_bf16x8 h, m, l; };
template <int B>
__device__ __forceinline__ B3Op b3_regs(const f32x16& v) {
  float x[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) x[e] = v[8 * B + e];
  B3Op o;
  b3_split8(x, o.h, o.m, o.l);
  return o;
}
template <int B>
__device__ __forceinline__ B3Op b3_regs(const float (&v)[16]) {
  float x[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) x[e] = v[8 * B + e];
  B3Op o;
  b3_split8(x, o.h, o.m, o.l);
  return o;
}
__device__ __forceinline__ B3Op b3_rows(const float4& g0, const float4& g1) {
  const float x[8] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w};
  B3Op o;
  b3_split8(x, o.h, o.m, o.l);
  return o;
}

// dW[32 mo + .][32 mi + .] += sum_s dY[s][.] X[s][.] over the 16 samples of one k-block, both operands in the
// lane = feature layout and already split.  Product-major: consecutive MFMAs go to different accumulators.
__device__ __forceinline__ void wgrad_b3_block(const B3Op (&A)[2], const B3Op (&Bx)[2], f32x16 (&acc)[2][2]) {
#define NGM_WG_PRODUCT(PA, PB_)                                                                       \
  _Pragma("unroll") for (int mo = 0; mo < 2; ++mo)                                                    \
    _Pragma("unroll") for (int mi = 0; mi < 2; ++mi) acc[mo][mi] = mfma_bf16(A[mo].PA, Bx[mi].PB_, acc[mo][mi]); \
  __builtin_amdgcn_sched_barrier(0)
  NGM_WG_PRODUCT(l, h);
  NGM_WG_PRODUCT(h, l);
  NGM_WG_PRODUCT(m, m);
  NGM_WG_PRODUCT(m, h);
  NGM_WG_PRODUCT(h, m);
  NGM_WG_PRODUCT(h, h);
#undef NGM_WG_PRODUCT
}
// same MFMAs without scheduling fences, for regions whose order is given by NGM_INTERLEAVE
__device__ __forceinline__ void wgrad_b3_block_free(const B3Op (&A)[2], const B3Op (&Bx)[2], f32x16 (&acc)[2][2]) {
#define NGM_WG_PRODUCT(PA, PB_)                                                                       \
  _Pragma("unroll") for (int mo = 0; mo < 2; ++mo)                                                    \
    _Pragma("unroll") for (int mi = 0; mi < 2; ++mi) acc[mo][mi] = mfma_bf16(A[mo].PA, Bx[mi].PB_, acc[mo][mi])
  NGM_WG_PRODUCT(l, h);
  NGM_WG_PRODUCT(h, l);
  NGM_WG_PRODUCT(m, m);
  NGM_WG_PRODUCT(m, h);
  NGM_WG_PRODUCT(h, m);
  NGM_WG_PRODUCT(h, h);
#undef NGM_WG_PRODUCT
}
// scheduling directive for the region it closes: N times (1 MFMA, then K VALU instructions).  One wave per SIMD: an
// MFMA occupies the matrix pipe for 32 clocks but the issue port for 4, and up to ~5 independent single-issue
// instructions of the SAME wave go out in its shadow (MI355X_MICROARCH.md, "one wave per SIMD") -- so the operand
// splits of the NEXT block are issued between the MFMAs of this one.
#define NGM_INTERLEAVE(N, K)                                          \
  _Pragma("unroll") for (int ii_ = 0; ii_ < (N); ++ii_) {              \
    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                 \
    __builtin_amdgcn_sched_group_barrier(0x002, (K), 0);               \
  }

__device__ __forceinline__ B3Op b3_arr(const float (&x)[8]) {
  B3Op o;
  b3_split8(x, o.h, o.m, o.l);
  return o;
}

// float offset of the 16-byte chunk c of sample s inside a tile
__device__ __forceinline__ int tile_chunk(int c, int s) { return (c * 32 + (s ^ (c & 7))) * 4; }

struct PlaneRegs { ngm_u32x4 h[2], m[2], l[2]; };
__device__ __forceinline__ void load_planes(const ngm_u32x4* __restrict__ P, int kb, int lane, PlaneRegs& W) {
  const int n = lane & 31, kh = lane >> 5;
#pragma unroll
  for (int nt = 0; nt < 2; ++nt) {
    const int g = ((nt * 4 + kb) * 2 + kh) * 32 + n;
    W.h[nt] = P[g]; W.m[nt] = P[PLANE_G + g]; W.l[nt] = P[2 * PLANE_G + g];
  }
}

// dX^T[s][32 nt + n] = sum_o dY[s][o] W[o][32 nt + n]: A = rows of the tile (lane = sample, loaded by the caller well
// ahead), B = weight planes, the next k-block's planes in flight under this one's MFMAs (one wave per SIMD: nothing
// else hides the LDS latency).  W0 = planes of k-block 0, loaded by the caller.
__device__ __forceinline__ void dgrad_b3_kb(const B3Op& A, const PlaneRegs& Wk, bool first, f32x16 (&dX)[2]) {
  const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#define NGM_DG_PRODUCT(PA, PW, Z) \
  _Pragma("unroll") for (int nt = 0; nt < 2; ++nt) dX[nt] = mfma_bf16(A.PA, __builtin_bit_cast(ngm_bf16x8, Wk.PW[nt]), (Z) ? zero : dX[nt]); \
  __builtin_amdgcn_sched_barrier(0)
  NGM_DG_PRODUCT(l, h, first);
  NGM_DG_PRODUCT(h, l, false);
  NGM_DG_PRODUCT(m, m, false);
  NGM_DG_PRODUCT(m, h, false);
  NGM_DG_PRODUCT(h, m, false);
  NGM_DG_PRODUCT(h, h, false);
#undef NGM_DG_PRODUCT
}
// ---- dY in the other orientation, through the matrix pipe --------------------------------------------------------
// The data gradient's A operand wants lane = sample with eight FEATURES of dY per k-block; the wave holds dY with lane =
// feature and eight SAMPLES per k-block (the weight gradient's A operand, already split into three bf16 planes).  Instead of
// a second split of the same numbers read back from LDS in the other orientation (176 VALU per layer and tile + 32 ds_write +
// 8 ds_read_b128 + two wave barriers), each PLANE is transposed exactly by the pipe that has the slack (0.3 busy):
//   T[o][s] = sum_k A[o][k] I[k][s],   A = plane operand (row = feature o, k = sample), I = identity (col = sample s),
// one non-zero product per element, fp32 accumulate -> T holds the plane's bf16 values exactly; its C fragment (lane =
// column = sample s, register r <-> feature frow(r, hi) of the 32-feature tile) packs pairwise (one v_perm per two values)
// into the data gradient's A operand when the contraction index of that GEMM is ordered the same way:
//   k-block kb = 2 m + b, lane half kh, element e  <->  feature 32 m + frow(8 b + e, kh) = 16 kb + 8 (e >> 2) + 4 kh + (e & 3)
// (build_dgrad_planes stores the weight planes in that order).  2 MFMAs per (plane, feature tile): 12 per layer.
__host__ __device__ __forceinline__ constexpr int dgrad_k_feature(int kb, int kh, int e) { return 16 * kb + 8 * (e >> 2) + 4 * kh + (e & 3); }

// identity operands of the transposition: lane (s = lane & 31, kh = lane >> 5), k-block b, element e <-> sample
// 16 b + 8 (e >> 2) + 4 kh + (e & 3) of the tile (the k order of the weight gradient's operands); 1.0 where that is s
struct IdOps { ngm_u32x4 b[2]; };
__device__ __forceinline__ IdOps make_id_ops(int lane) {
  const int s = lane & 31, kh = lane >> 5;
  IdOps I;
#pragma unroll
  for (int b = 0; b < 2; ++b)
#pragma unroll
    for (int d = 0; d < 4; ++d) {
      uint32_t w = 0;
#pragma unroll
      for (int half = 0; half < 2; ++half) {
        const int e = 2 * d + half;
        if (16 * b + 8 * (e >> 2) + 4 * kh + (e & 3) == s) w |= 0x3f80u << (16 * half);
      }
      I.b[b][d] = w;
    }
  return I;
}
// One feature tile (32 features) at a time: three planes x two k-blocks of samples = six MFMAs into three accumulators
// (48 registers live instead of 96: the kernels that use this sit at their register limit).  a0 / a1: the weight gradient's
// A operands of the feature tile for k-block 0 / 1 (samples 0..15 / 16..31 of the tile).
struct XposeAcc { f32x16 t[3]; };        // [plane h, m, l]
// The six MFMAs are ONE inline-asm statement with VGPR destinations.  Through the builtin, a kernel that owns 512 registers
// gets every MFMA in its AGPR form (LLVM selects the form per function), and each of the 96 transposed values would then cost
// a v_accvgpr_read before the v_perm that packs it -- as many VALU instructions as the split this replaces.  hipcc pads no
// hazards inside asm (cdna_hip_programming.md 5.7): `s_nop 1` covers VALU-written operands -> MFMA, the accumulate chains
// need none, and `s_nop 11` (12 states, 8-pass XDL) ends the string so that any reader of the results may follow.
__device__ __forceinline__ void xpose_tile(const B3Op& a0, const B3Op& a1, const IdOps& I, XposeAcc& T) {
#ifdef NGM_B3_XBUILTIN
  const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  const ngm_bf16x8 i0 = __builtin_bit_cast(ngm_bf16x8, I.b[0]), i1 = __builtin_bit_cast(ngm_bf16x8, I.b[1]);
  T.t[0] = mfma_bf16(a0.h, i0, zero);
  T.t[1] = mfma_bf16(a0.m, i0, zero);
  T.t[2] = mfma_bf16(a0.l, i0, zero);
  T.t[0] = mfma_bf16(a1.h, i1, T.t[0]);
  T.t[1] = mfma_bf16(a1.m, i1, T.t[1]);
  T.t[2] = mfma_bf16(a1.l, i1, T.t[2]);
  return;
#endif
  asm volatile(
      "s_nop 1\n\t"
      "v_mfma_f32_32x32x16_bf16 %0, %3, %9, 0\n\t"
      "v_mfma_f32_32x32x16_bf16 %1, %4, %9, 0\n\t"
      "v_mfma_f32_32x32x16_bf16 %2, %5, %9, 0\n\t"
      "v_mfma_f32_32x32x16_bf16 %0, %6, %10, %0\n\t"
      "v_mfma_f32_32x32x16_bf16 %1, %7, %10, %1\n\t"
      "v_mfma_f32_32x32x16_bf16 %2, %8, %10, %2\n\t"
      "s_nop 11"
      : "=&v"(T.t[0]), "=&v"(T.t[1]), "=&v"(T.t[2])
      : "v"(__builtin_bit_cast(ngm_u32x4, a0.h)), "v"(__builtin_bit_cast(ngm_u32x4, a0.m)), "v"(__builtin_bit_cast(ngm_u32x4, a0.l)),
        "v"(__builtin_bit_cast(ngm_u32x4, a1.h)), "v"(__builtin_bit_cast(ngm_u32x4, a1.m)), "v"(__builtin_bit_cast(ngm_u32x4, a1.l)),
        "v"(I.b[0]), "v"(I.b[1]));
}
// dH^T[s][o] = sum_c d_out[s][c] Wout[c][o] for both feature tiles: six products of the split, two chains, VGPR destinations
// (the ReLU mask reads them); operands as 128-bit register tuples, upper k half zero
__device__ __forceinline__ void outlayer_mfma(const ngm_u32x4& ah, const ngm_u32x4& am, const ngm_u32x4& al, const ngm_u32x4 (&wh)[2],
                                              const ngm_u32x4 (&wm)[2], const ngm_u32x4 (&wl)[2], f32x16 (&dH)[2]) {
#ifdef NGM_B3_OLBUILTIN
  const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#define NGM_OLB(A, W, Z) _Pragma("unroll") for (int m = 0; m < 2; ++m) dH[m] = mfma_bf16(__builtin_bit_cast(ngm_bf16x8, A), __builtin_bit_cast(ngm_bf16x8, W[m]), (Z) ? zero : dH[m])
  NGM_OLB(al, wh, true); NGM_OLB(ah, wl, false); NGM_OLB(am, wm, false); NGM_OLB(am, wh, false); NGM_OLB(ah, wm, false); NGM_OLB(ah, wh, false);
#undef NGM_OLB
  return;
#endif
  asm volatile(
      "s_nop 1\n\t"
      "v_mfma_f32_32x32x16_bf16 %0, %4, %5, 0\n\t"      // lo x hi
      "v_mfma_f32_32x32x16_bf16 %1, %4, %6, 0\n\t"
      "v_mfma_f32_32x32x16_bf16 %0, %2, %9, %0\n\t"     // hi x lo
      "v_mfma_f32_32x32x16_bf16 %1, %2, %10, %1\n\t"
      "v_mfma_f32_32x32x16_bf16 %0, %3, %7, %0\n\t"     // mid x mid
      "v_mfma_f32_32x32x16_bf16 %1, %3, %8, %1\n\t"
      "v_mfma_f32_32x32x16_bf16 %0, %3, %5, %0\n\t"     // mid x hi
      "v_mfma_f32_32x32x16_bf16 %1, %3, %6, %1\n\t"
      "v_mfma_f32_32x32x16_bf16 %0, %2, %7, %0\n\t"     // hi x mid
      "v_mfma_f32_32x32x16_bf16 %1, %2, %8, %1\n\t"
      "v_mfma_f32_32x32x16_bf16 %0, %2, %5, %0\n\t"     // hi x hi
      "v_mfma_f32_32x32x16_bf16 %1, %2, %6, %1\n\t"
      "s_nop 11"
      : "=&v"(dH[0]), "=&v"(dH[1])
      : "v"(ah), "v"(am), "v"(al), "v"(wh[0]), "v"(wh[1]), "v"(wm[0]), "v"(wm[1]), "v"(wl[0]), "v"(wl[1]));
}
// the two k-blocks' A operands of the data gradient this feature tile supplies (k-blocks 2 m, 2 m + 1): 24 v_perm
__device__ __forceinline__ void xpose_pack(const XposeAcc& T, B3Op& At0, B3Op& At1) {
#pragma unroll
  for (int b = 0; b < 2; ++b) {
    ngm_u32x4 h, mm, l;
#pragma unroll
    for (int d = 0; d < 4; ++d) {
      h[d] = b3_pack(__float_as_uint(T.t[0][8 * b + 2 * d]), __float_as_uint(T.t[0][8 * b + 2 * d + 1]));
      mm[d] = b3_pack(__float_as_uint(T.t[1][8 * b + 2 * d]), __float_as_uint(T.t[1][8 * b + 2 * d + 1]));
      l[d] = b3_pack(__float_as_uint(T.t[2][8 * b + 2 * d]), __float_as_uint(T.t[2][8 * b + 2 * d + 1]));
    }
    B3Op& o = b ? At1 : At0;
    o.h = __builtin_bit_cast(ngm_bf16x8, h);
    o.m = __builtin_bit_cast(ngm_bf16x8, mm);
    o.l = __builtin_bit_cast(ngm_bf16x8, l);
  }
}
// data gradient from ready A operands: only the plane reads of the next k-block travel under this one's MFMAs
__device__ __forceinline__ void dgrad_b3_t(const ngm_u32x4* __restrict__ P, const B3Op (&At)[4], const PlaneRegs& W0, int lane, f32x16 (&dX)[2]) {
  PlaneRegs Wa, Wb;
  load_planes(P, 1, lane, Wb);
  dgrad_b3_kb(At[0], W0, true, dX);
  load_planes(P, 2, lane, Wa);
  dgrad_b3_kb(At[1], Wb, false, dX);
  load_planes(P, 3, lane, Wb);
  dgrad_b3_kb(At[2], Wa, false, dX);
  dgrad_b3_kb(At[3], Wb, false, dX);
}

// weight planes of layer l for the data gradient: granule (plane, nt, kb, kh, n) = W[dgrad_k_feature(kb, kh, e)][32 nt + n], e = 0..7
__device__ __forceinline__ void build_dgrad_planes(const ngm_field_cfg& fc, const ngm_params& pr, int64_t row, int l, ngm_u32x4* P) {
  const int Din = (l == 0) ? fc.dim_enc : fc.dim_hidden, H = fc.dim_hidden;
  const float* W = pr.w[l];
  const int64_t w0 = row * pr.w_stride[l];
  for (int g = threadIdx.x; g < PLANE_G; g += B3B_THREADS) {
    const int n = g & 31, kh = (g >> 5) & 1, kb = (g >> 6) & 3, nt = g >> 8;
    const int c = 32 * nt + n;
    // the eight loads first, then their use: element by element (ngm_ldp inside a bounds check) every load was waited for
    // where it was issued -- 32 serial L2 round trips per thread in front of the first tile
    float x[8];
    int off[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int o = dgrad_k_feature(kb, kh, e);
      off[e] = (o < H && c < Din) ? o * Din + c : 0;
    }
    ngm_ldp_gather<8>(W, w0, off, pr.dtype, x);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int o = dgrad_k_feature(kb, kh, e);
      x[e] = (o < H && c < Din) ? x[e] : 0.f;
    }
    ngm_bf16x8 h, m, lo;
    b3_split8(x, h, m, lo);
    P[g] = __builtin_bit_cast(ngm_u32x4, h);
    P[PLANE_G + g] = __builtin_bit_cast(ngm_u32x4, m);
    P[2 * PLANE_G + g] = __builtin_bit_cast(ngm_u32x4, lo);
  }
}

// encoding row of feature f: (w.x, w.y, w.z, kind) -- the table FieldStage16::issue builds, one entry per lane here
__device__ __forceinline__ float4 enc_row_of(const ngm_field_cfg& fc, const ngm_params& pr, int64_t row, int f) {
  float4 e = make_float4(0.f, 0.f, 0.f, NGM_FK_ZERO);
  if (f >= fc.dim_enc) return e;
  if (fc.encoding == NGM_ENC_FOURIER) {
    const int n_raw = fc.raw_coords ? 3 : 0;
    if (f < n_raw) return make_float4(f == 0 ? 1.f : 0.f, f == 1 ? 1.f : 0.f, f == 2 ? 1.f : 0.f, NGM_FK_RAW);
    const int64_t e0 = row * pr.enc_w_stride + (int64_t)(f - n_raw) * 3;
    return make_float4(ngm_ldp(pr.enc_w, e0, pr.dtype), ngm_ldp(pr.enc_w, e0 + 1, pr.dtype), ngm_ldp(pr.enc_w, e0 + 2, pr.dtype), NGM_FK_SIN);
  }
  if (fc.encoding == NGM_ENC_NERF) {
    const int half = 3 * fc.num_octaves;
    const int g = (f < half) ? f : f - half;
    const int d = g / fc.num_octaves, o = g % fc.num_octaves;
    const float m = exp2f((float)(fc.start_octave + o)) * 3.14159265358979323846f;
    return make_float4(d == 0 ? m : 0.f, d == 1 ? m : 0.f, d == 2 ? m : 0.f, (f < half) ? NGM_FK_SIN : NGM_FK_COS);
  }
  return make_float4(f == 0 ? 1.f : 0.f, f == 1 ? 1.f : 0.f, f == 2 ? 1.f : 0.f, NGM_FK_RAW);
}

