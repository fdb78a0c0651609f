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

// HS ("half stash", L = 2 only): the forward stashed layer 0's output alone; the output layer's input H2 = relu(W1 H1 + b1) is
// recomputed per tile on the matrix pipe (48 more MFMAs) from the H1 tile that is DMA'd anyway.  One more plane set (W1 in the
// forward orientation), one tile buffer per wave instead of two (H1 is in registers -- rows for the recompute, columns for the
// weight gradient -- before dY overwrites it), half the transfers and half the stash traffic.
template <int L, bool EG, bool HS = false>
struct LdsB3b {
  static_assert(!HS || L == 2, "half stash: two hidden layers");
  static constexpr int NPL = (L - 1) + (EG ? 1 : 0) + (HS ? 1 : 0);  // layers whose data gradient is needed (+ HS: layer 1 forward)
  static constexpr int plane_slot(int l) { return EG ? l : l - 1; }  // 16-byte units: slot * 3 * PLANE_G
  static constexpr int fwd_slot() { return NPL - 1; }                // HS: W1 for the recompute, granule = W1[32 nt + n][16 kb + 8 kh + e]
  static constexpr int PLANES = NPL * 3 * PLANE_G * 4 + 512 + 64;    // floats; + per-feature constants: float4 wout[64], enc[64], float b1[64]
  static constexpr int CONSTS = NPL * 3 * PLANE_G * 4;
  // per wave: the output layer's input tile, (L = 2) layer 1's input tile, the input landing buffer, points, d_out.
  // Single buffers: the next tile's transfers are issued when all of them are free (see the tile loop).
  static constexpr int HL = 0;
  static constexpr int H1 = HS ? 0 : HT;                             // HS: the one tile buffer (H1 -> dY of layer 1 -> dY of layer 0)
  static constexpr int INB = H1 + ((L == 2) ? HT : 0);               // float4 [2 halves][4 pieces][16] + (fused compositing) [2 pieces][32]
  static constexpr int PB = INB + 768;                               // float4 [32]
  static constexpr int OB = PB + 128;                                // float4 [32]
  // fused compositing (FC): the inputs of the tile after next, and the position / gradient rows of the next tile (one 64-lane
  // pass computes two tiles' worth)
  static constexpr int INB2 = OB + 128;
  static constexpr int PB2 = INB2 + 768;
  static constexpr int OB2 = PB2 + 128;
  // HS: the per-feature sums (output-weight, bias, Fourier-matrix gradients: 18 floats per lane) live here between the phases
  // that touch them -- as loop-long registers next to the recompute's operands they were spilled to scratch, and a scratch
  // reload is a full exposed memory round trip on a one-wave SIMD.  float4 [5][64 lanes]
  static constexpr int ACCL = OB2 + 128;
  static constexpr int WAVE_TOTAL = ACCL + (HS ? 5 * 256 : 0);
  static constexpr int NT = 4 * L;                                   // 32x32 accumulator tiles per wave
  static constexpr int EPI_ACC = B3B_WAVES * NT * 1024;               // epilogue staging: the waves' accumulator tiles ...
  static constexpr int EPI = EPI_ACC + B3B_WAVES * (2 * L + 18) * 64;  // ... and their per-feature vectors behind them
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
      "global_load_lds_dwordx4 %1, %2 " NGM_DMA_HINT "\n\t"
      "global_load_lds_dwordx4 %1, %2 offset:1024 " NGM_DMA_HINT "\n\t"
      "global_load_lds_dwordx4 %1, %2 offset:2048 " NGM_DMA_HINT "\n\t"
      "global_load_lds_dwordx4 %1, %2 offset:3072 " NGM_DMA_HINT "\n\t"
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

// rows of a tile for the data gradient's A operand (lane = sample n, k-half kh): chunks 4 kb + 2 kh, + 1 of every k-block
struct RowRegs { float4 g[4][2]; };
__device__ __forceinline__ void load_rows(const float* __restrict__ tile, int lane, RowRegs& R) {
  const int n = lane & 31, kh = lane >> 5;
#pragma unroll
  for (int kb = 0; kb < 4; ++kb) {
    const int c0 = 4 * kb + 2 * kh;
    R.g[kb][0] = *reinterpret_cast<const float4*>(tile + tile_chunk(c0, n));
    R.g[kb][1] = *reinterpret_cast<const float4*>(tile + tile_chunk(c0 + 1, n));
  }
}
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
__device__ __forceinline__ void dgrad_b3_kb_free(const B3Op& A, const PlaneRegs& Wk, bool first, f32x16 (&dX)[2]) {
  const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#define NGM_DG_PRODUCT(PA, PW, Z) \
  _Pragma("unroll") for (int nt = 0; nt < 2; ++nt) dX[nt] = mfma_bf16(A.PA, __builtin_bit_cast(ngm_bf16x8, Wk.PW[nt]), (Z) ? zero : dX[nt])
  NGM_DG_PRODUCT(l, h, first);
  NGM_DG_PRODUCT(h, l, false);
  NGM_DG_PRODUCT(m, m, false);
  NGM_DG_PRODUCT(m, h, false);
  NGM_DG_PRODUCT(h, m, false);
  NGM_DG_PRODUCT(h, h, false);
#undef NGM_DG_PRODUCT
}
// k-block kb's 12 MFMAs share their scheduling region with the row split of k-block kb + 1 (and the plane reads of kb + 1)
__device__ __forceinline__ void dgrad_b3(const ngm_u32x4* __restrict__ P, const RowRegs& R, const PlaneRegs& W0, int lane, f32x16 (&dX)[2]) {
  PlaneRegs Wa, Wb;
  const B3Op A0 = b3_rows(R.g[0][0], R.g[0][1]);
  __builtin_amdgcn_sched_barrier(0);
  load_planes(P, 1, lane, Wb);
  const B3Op A1 = b3_rows(R.g[1][0], R.g[1][1]);
  dgrad_b3_kb_free(A0, W0, true, dX);
  NGM_INTERLEAVE(12, 4)
  __builtin_amdgcn_sched_barrier(0);
  load_planes(P, 2, lane, Wa);
  const B3Op A2 = b3_rows(R.g[2][0], R.g[2][1]);
  dgrad_b3_kb_free(A1, Wb, false, dX);
  NGM_INTERLEAVE(12, 4)
  __builtin_amdgcn_sched_barrier(0);
  load_planes(P, 3, lane, Wb);
  const B3Op A3 = b3_rows(R.g[3][0], R.g[3][1]);
  dgrad_b3_kb_free(A2, Wa, false, dX);
  NGM_INTERLEAVE(12, 4)
  __builtin_amdgcn_sched_barrier(0);
  dgrad_b3_kb(A3, Wb, false, dX);
  __builtin_amdgcn_sched_barrier(0);
}

// weight planes of layer l for the data gradient: granule (plane, nt, kb, kh, n) = W[16 kb + 8 kh + e][32 nt + n], e = 0..7
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
      const int o = 16 * kb + 8 * kh + e;
      off[e] = (o < H && c < Din) ? o * Din + c : 0;
    }
    ngm_ldp_gather<8>(W, w0, off, pr.dtype, x);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int o = 16 * kb + 8 * kh + e;
      x[e] = (o < H && c < Din) ? x[e] : 0.f;
    }
    ngm_bf16x8 h, m, lo;
    b3_split8(x, h, m, lo);
    P[g] = __builtin_bit_cast(ngm_u32x4, h);
    P[PLANE_G + g] = __builtin_bit_cast(ngm_u32x4, m);
    P[2 * PLANE_G + g] = __builtin_bit_cast(ngm_u32x4, lo);
  }
}

// The same in two phases, for a prologue that has every global load of the workgroup in flight at once (round 6: built one
// after the other -- constants, loss partials, two plane iterations per layer, each load waited for where it was issued -- the
// prologue was ~9 serial memory round trips: 15.9 k clocks of a 300 k-clock kernel at the M1 batch, 11 k of 43 k at one field
// x 512 rays x 24 samples).  planes_offsets / ngm_ldp_gather issue, planes_commit masks, splits and stores.
// FWD: the forward orientation (build_fwd_planes) instead of the data-gradient one.
template <bool FWD>
__device__ __forceinline__ void planes_offsets(const ngm_field_cfg& fc, int l, int it, int (&off)[8]) {
  const int Din = (l == 0) ? fc.dim_enc : fc.dim_hidden, H = fc.dim_hidden;
  const int g = threadIdx.x + it * B3B_THREADS;
  const int n = g & 31, kh = (g >> 5) & 1, kb = (g >> 6) & 3, nt = g >> 8;
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const int o = FWD ? 32 * nt + n : 16 * kb + 8 * kh + e, c = FWD ? 16 * kb + 8 * kh + e : 32 * nt + n;
    off[e] = (o < H && c < Din) ? o * Din + c : 0;
  }
}
template <bool FWD>
__device__ __forceinline__ void planes_commit(const ngm_field_cfg& fc, int l, int it, const float* xin, ngm_u32x4* P) {
  const int Din = (l == 0) ? fc.dim_enc : fc.dim_hidden, H = fc.dim_hidden;
  const int g = threadIdx.x + it * B3B_THREADS;
  const int n = g & 31, kh = (g >> 5) & 1, kb = (g >> 6) & 3, nt = g >> 8;
  float x[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const int o = FWD ? 32 * nt + n : 16 * kb + 8 * kh + e, c = FWD ? 16 * kb + 8 * kh + e : 32 * nt + n;
    x[e] = (o < H && c < Din) ? xin[e] : 0.f;
  }
  ngm_bf16x8 h, m, lo;
  b3_split8(x, h, m, lo);
  P[g] = __builtin_bit_cast(ngm_u32x4, h);
  P[PLANE_G + g] = __builtin_bit_cast(ngm_u32x4, m);
  P[2 * PLANE_G + g] = __builtin_bit_cast(ngm_u32x4, lo);
}
static_assert(PLANE_G == 2 * B3B_THREADS, "two plane granules per thread");

// HS: weight planes of layer l in the FORWARD orientation, for Y^T[s][32 nt + n] = sum_i X[s][i] W[32 nt + n][i] with the rows
// of the X tile as A operand (same lane / k order as the data gradient's): granule (plane, nt, kb, kh, n) = W[32 nt + n][16 kb + 8 kh + e]
__device__ __forceinline__ void build_fwd_planes(const ngm_field_cfg& fc, const ngm_params& pr, int64_t row, int l, ngm_u32x4* P) {
  const int Din = (l == 0) ? fc.dim_enc : fc.dim_hidden, H = fc.dim_hidden;
  const float* W = pr.w[l];
  const int64_t w0 = row * pr.w_stride[l];
  for (int g = threadIdx.x; g < PLANE_G; g += B3B_THREADS) {
    const int n = g & 31, kh = (g >> 5) & 1, kb = (g >> 6) & 3, nt = g >> 8;
    const int o = 32 * nt + n;
    float x[8];
    int off[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int c = 16 * kb + 8 * kh + e;
      off[e] = (o < H && c < Din) ? o * Din + c : 0;
    }
    ngm_ldp_gather<8>(W, w0, off, pr.dtype, x);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int c = 16 * kb + 8 * kh + e;
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

