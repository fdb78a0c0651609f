// k_field_bwd_b3p: the MLP backward of the fused training step for TWO hidden layers when the forward stashed layer 0's output
// as bf16 PLANES (ActStash, ngm_field.h; FieldBwdArgs::act_half == 2).  Same arithmetic as k_field_bwd_b3 (three-way bf16
// split, six products on v_mfma_f32_32x32x16_bf16, fp32 accumulate; fused compositing backward; deterministic) -- what
// changes is HOW the operands reach the matrix pipe.  k_field_bwd_b3 is bound by its vector instruction stream (~2 900 VALU
// per 32-sample tile, 43 % of them the seven operand splits: v_and / v_sub / v_perm); here
//   * H1 (layer 0's output) arrives ALREADY SPLIT: the forward stores the three planes its own layer-1 products use.  Rows
//     (A operand of the recompute of H2 = relu(W1 H1 + b1)) are two 8-byte LDS cells per plane and k-block; columns (B operand
//     of layer 1's weight gradient) come through gfx950's LDS transpose read, ds_read_b64_tr_b16.  No split of H1 at all.
//   * every dY is split ONCE, in the "lane = feature" orientation the weight gradient wants; its planes go to LDS as 8-byte
//     cells (feature, 4 consecutive samples) and the data gradient reads them transposed (lane = sample) -- the fp32 round
//     trip and the second split are gone.
//   * W1 is kept as ONE plane set in the forward orientation: the recompute reads cells, layer 1's data gradient reads the
//     same cells transposed.  (That is what makes room for the 12 KB plane tiles: 150 KB of LDS.)
// Splits left per tile: dY of layer 1, dY of layer 0, the encoding -- 3 of 7.
// Cell layouts (8-byte cells of four bf16, position XOR 4 (group & 3): the 16 lanes of a transpose read then hit 16
// different cells of 128 contiguous bytes, and a half-wave's row / store accesses are a permutation of 256 contiguous bytes):
//   H1 tile   [plane 3][fg = feature >> 2 : 16][sample 32]       cell = features 4 fg .. + 3 of one sample        (4 KB / plane)
//   dY tile   [plane 3][sg = sample >> 2 : 8][feature 64]        cell = samples 4 sg .. + 3 of one feature        (4 KB / plane)
//   W1        [plane 3][ig = input >> 2 : 16][output 64]         cell = W1[o][4 ig .. + 3]                        (8 KB / plane)
// ds_read_b64_tr_b16 (tools/micro/tr_read.hip): within a 16-lane group source lane j supplies the address of one cell
// in[j][0..3]; result lane i receives out[i][k] = in[4 k + (i >> 2)][i & 3], k = 0..3.
#include "ngm_bwd_b3.h"

typedef unsigned v2u __attribute__((ext_vector_type(2)));
typedef short v4s_ __attribute__((ext_vector_type(4)));
#define P3_TILE_BYTES 12288

struct LdsB3p {                                  // byte offsets
  static constexpr int W0P = 0;                  // layer 0's data-gradient planes (Fourier): 3 x 512 x 16 B, the k_field_bwd_b3 format
  static constexpr int W1C = 24576;              // W1 cell planes: 3 x 8 KB
  static constexpr int CONSTS = W1C + 24576;     // float4 wout[64], float4 enc[64], float b1[64]
  static constexpr int WAVES = CONSTS + 2304;
  static constexpr int TILE = 0;                 // per wave: the 12 KB plane tile (H1, then dY of layer 1, then dY of layer 0)
  static constexpr int INB = P3_TILE_BYTES;      // the small landing / row buffers of k_field_bwd_b3 (floats there, bytes here)
  static constexpr int PB = INB + 3072;
  static constexpr int OB = PB + 512;
  static constexpr int INB2 = OB + 512;
  static constexpr int PB2 = INB2 + 3072;
  static constexpr int OB2 = PB2 + 512;
  static constexpr int ACCL = OB2 + 512;         // float4 [5][64]: the per-feature sums between the phases that touch them
  static constexpr int WAVE_TOTAL = ACCL + 5120;
  static constexpr int NT = 8;
  static constexpr int EPI = B3B_WAVES * NT * 4096;
  static constexpr int BODY = WAVES + B3B_WAVES * WAVE_TOTAL;
  static constexpr int TOTAL = BODY > EPI ? BODY : EPI;
};

__device__ __forceinline__ v2u lds_tr(const char* p) {
  const v4s_ v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) v4s_*)(p));
  return __builtin_bit_cast(v2u, v);
}
__device__ __forceinline__ ngm_bf16x8 bf8(v2u lo, v2u hi) { return __builtin_bit_cast(ngm_bf16x8, ngm_u32x4{lo.x, lo.y, hi.x, hi.y}); }

// W1 as cells: (plane, ig, o) <- W1[o][4 ig .. 4 ig + 3] split three ways; 1024 cells per plane
__device__ __forceinline__ void build_w1_cells(const ngm_field_cfg& fc, const ngm_params& pr, int64_t row, char* dst) {
  const int H = fc.dim_hidden;
  const float* W = pr.w[1];
  const int64_t w0 = row * pr.w_stride[1];
  for (int c = threadIdx.x; c < 1024; c += B3B_THREADS) {
    const int o = c & 63, ig = c >> 6;
    float x[4];
    int off[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) off[e] = (o < H && 4 * ig + e < H) ? o * H + 4 * ig + e : 0;
    ngm_ldp_gather<4>(W, w0, off, pr.dtype, x);
#pragma unroll
    for (int e = 0; e < 4; ++e) x[e] = (o < H && 4 * ig + e < H) ? x[e] : 0.f;
    uint32_t h0, m0, l0, h1, m1, l1;
    b3_split2(x[0], x[1], h0, m0, l0);
    b3_split2(x[2], x[3], h1, m1, l1);
    const int cell = (ig * 64 + (o ^ (4 * (ig & 3)))) * 8;
    *reinterpret_cast<v2u*>(dst + cell) = v2u{h0, h1};
    *reinterpret_cast<v2u*>(dst + 8192 + cell) = v2u{m0, m1};
    *reinterpret_cast<v2u*>(dst + 16384 + cell) = v2u{l0, l1};
  }
}

// per-lane byte offsets into the cell arrays (constant over the kernel)
struct P3Lane {
  int row_h1;     // H1 tile, rows: cell (fg = 2 kh, sample n)            + 1024 kb (fg += 4) ; second cell: fg + 1 -> see h1_row()
  int col_h1;     // H1 tile, transposed: source cell of this lane for (m = 0, b = 0, half 0): + 2048 m + 128 b, ^ 64 for half 1
  int row_w1;     // W1 cells, rows: cell (ig = 2 kh, o = n)              + 2048 kb, + 256 nt; second cell: ig + 1
  int tr_dy;      // dY tile, transposed: source cell for (kb = 0, half 0): + 128 kb, ^ 32 for half 1
  int tr_w1;      // W1 cells, transposed: source cell for (nt = 0, kb = 0, half 0): + 4096 nt, + 128 kb, ^ 32 for half 1
  int st_dy;      // dY tile, store: cell (sg = hi, feature i) of (m = 0, b = 0, half 0): + 256 m ... see dy_store()
};
__device__ __forceinline__ P3Lane p3_lane(int lane) {
  const int n = lane & 31, kh = lane >> 5, j = lane & 15, gq = lane >> 4, k = j >> 2, a = j & 3;
  P3Lane L;
  L.row_h1 = 0; L.row_w1 = 0; L.st_dy = 0;     // (computed where used: they depend on the group index of the cell)
  // H1 columns: result lanes = 16 consecutive features (fgb = 4 (gq & 1) [+ 8 m]), k = 4 consecutive samples s0 + k,
  // s0 = 4 hi (hi = gq >> 1) [+ 16 b, + 8 half]
  L.col_h1 = ((4 * (gq & 1) + a) * 32 + ((4 * (gq >> 1) + k) ^ (4 * a))) * 8;
  // dY rows: result lanes = 16 consecutive samples (sgb = 4 (gq & 1)), k = 4 consecutive features f0 + k, f0 = 8 kh (kh = gq >> 1)
  L.tr_dy = ((4 * (gq & 1) + a) * 64 + ((8 * (gq >> 1) + k) ^ (4 * a))) * 8;
  // W1 transposed: result lanes = 16 consecutive inputs i (igb = 4 (gq & 1) [+ 8 nt]), k = 4 consecutive outputs o0 + k, o0 = 8 kh
  L.tr_w1 = L.tr_dy;
  (void)n; (void)kh;
  return L;
}
// rows of the H1 tile / of W1 for k-block kb: the two cells (g, x), (g + 1, x) of group g = 4 kb + 2 kh hold the eight
// contraction values 16 kb + 8 kh + e of row x (x = sample n / output 32 nt + n); NX = cells per group (32 / 64)
template <int NX>
__device__ __forceinline__ ngm_bf16x8 cells_row(const char* plane, int kb, int kh, int x) {
  const int g = 4 * kb + 2 * kh;
  const v2u c0 = *reinterpret_cast<const v2u*>(plane + (g * NX + (x ^ (4 * (g & 3)))) * 8);
  const v2u c1 = *reinterpret_cast<const v2u*>(plane + ((g + 1) * NX + (x ^ (4 * ((g + 1) & 3)))) * 8);
  return bf8(c0, c1);
}

// ------------------------------------------------------------------------------------------------
// FC: the compositing backward of k_stash_bwd fused into the input phase (FieldBwdArgs::fused_comp): the d_out stream then
// carries the forward's (colour, geometry) stash, every wave walks a contiguous ray-aligned range of tiles BACK TO FRONT
// and carries the suffix value Q of the per-ray recursion Q_{k-1} = a_k o_k + (1 - o_k) Q_k from tile to tile.
// one 12 KB plane tile of the field, HBM -> LDS: twelve linear 1 KB copies (the stash tile IS the LDS image)
__device__ __forceinline__ void issue_plane_tile(const char* fbase, uint32_t tile_idx, int lane, uint32_t lds_tile) {
  const char* b0 = fbase + (size_t)__builtin_amdgcn_readfirstlane(tile_idx) * P3_TILE_BYTES;
  dma16_x4(b0, (uint32_t)lane * 16u, lds_tile);
  dma16_x4(b0 + 4096, (uint32_t)lane * 16u, lds_tile + 4096);
  dma16_x4(b0 + 8192, (uint32_t)lane * 16u, lds_tile + 8192);
}
// the field's last tile when P % 32 != 0: cells of samples >= nv were never written by the forward (whatever bits the
// workspace held, NaN patterns included, would reach the MFMAs with a zero gradient: 0 x NaN) -> zero them
__device__ __forceinline__ void sanitize_plane_tile(char* tile, int nv, int lane) {
  for (int c = lane; c < 3 * 512; c += 64) {
    const int pos = c & 31, fg = (c >> 5) & 15;
    const int s = pos ^ (4 * (fg & 3));
    if (s >= nv) *reinterpret_cast<v2u*>(tile + 8 * c) = v2u{0u, 0u};
  }
  WAVE_SYNC();
}

// dY planes -> cells: lane (i, hi) holds, for feature 32 m + i and k-block b, the 8 samples 16 b + 4 hi + {0..3} (cell
// sg = 4 b + hi) and 16 b + 8 + 4 hi + {0..3} (cell sg + 2) as the two dword pairs of each plane
__device__ __forceinline__ void dy_store_cells(char* tile, int lane, int b, const B3Op (&A)[2]) {
  const int i = lane & 31, hi = lane >> 5;
#pragma unroll
  for (int m = 0; m < 2; ++m) {
    const ngm_u32x4 h4 = __builtin_bit_cast(ngm_u32x4, A[m].h), m4 = __builtin_bit_cast(ngm_u32x4, A[m].m),
                    l4 = __builtin_bit_cast(ngm_u32x4, A[m].l);
#pragma unroll
    for (int w = 0; w < 2; ++w) {
      const int sg = 4 * b + 2 * w + hi;
      const int off = (sg * 64 + ((32 * m + i) ^ (4 * (sg & 3)))) * 8;
      *reinterpret_cast<v2u*>(tile + off) = v2u{h4[2 * w], h4[2 * w + 1]};
      *reinterpret_cast<v2u*>(tile + 4096 + off) = v2u{m4[2 * w], m4[2 * w + 1]};
      *reinterpret_cast<v2u*>(tile + 8192 + off) = v2u{l4[2 * w], l4[2 * w + 1]};
    }
  }
}

// ... and back: the lane's OWN cells are its weight-gradient A operands (no transposition: lane = feature on both sides)
__device__ __forceinline__ void dy_load_cells(const char* tile, int lane, int b, B3Op (&A)[2]) {
  const int i = lane & 31, hi = lane >> 5;
#pragma unroll
  for (int m = 0; m < 2; ++m) {
    v2u h[2], mm[2], l[2];
#pragma unroll
    for (int w = 0; w < 2; ++w) {
      const int sg = 4 * b + 2 * w + hi;
      const int off = (sg * 64 + ((32 * m + i) ^ (4 * (sg & 3)))) * 8;
      h[w] = *reinterpret_cast<const v2u*>(tile + off);
      mm[w] = *reinterpret_cast<const v2u*>(tile + 4096 + off);
      l[w] = *reinterpret_cast<const v2u*>(tile + 8192 + off);
    }
    A[m].h = bf8(h[0], h[1]); A[m].m = bf8(mm[0], mm[1]); A[m].l = bf8(l[0], l[1]);
  }
}

// weight-gradient blocks with the two B operands (feature tiles mi = 0, 1) passed separately
__device__ __forceinline__ void wgrad_b3_block_free(const B3Op (&A)[2], const B3Op& B0, const B3Op& B1, f32x16 (&acc)[2][2]) {
  const B3Op Bx[2] = {B0, B1};
  wgrad_b3_block_free(A, Bx, acc);
}
__device__ __forceinline__ void wgrad_b3_block2(const B3Op (&A)[2], const B3Op& B0, const B3Op& B1, f32x16 (&acc)[2][2]) {
  const B3Op Bx[2] = {B0, B1};
  wgrad_b3_block(A, Bx, acc);
}

// data gradient dX^T[s][32 nt + n] = sum_o dY[s][o] W[o][32 nt + n] with the A operand read TRANSPOSED from the dY cells
// (lane = sample, no split) and the B operand either transposed from W1's cells (W1T) or from layer 0's plane set in
// k_field_bwd_b3's format.  Every operand of a k-block is in flight one k-block ahead (one wave per SIMD).
struct TrOps { ngm_bf16x8 ah, am, al; ngm_bf16x8 bh[2], bm[2], bl[2]; };
template <bool W1T>
__device__ __forceinline__ void dgrad_cells_load(const char* tile, const char* w1c, const ngm_u32x4* P, const P3Lane& pl, int lane,
                                                 int kb, TrOps& o) {
  const char* t0 = tile + pl.tr_dy + 128 * kb;
  const char* t1 = tile + (pl.tr_dy ^ 32) + 128 * kb;
  o.ah = bf8(lds_tr(t0), lds_tr(t1));
  o.am = bf8(lds_tr(t0 + 4096), lds_tr(t1 + 4096));
  o.al = bf8(lds_tr(t0 + 8192), lds_tr(t1 + 8192));
  if constexpr (W1T) {
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
      const char* w0 = w1c + pl.tr_w1 + 4096 * nt + 128 * kb;
      const char* w1 = w1c + (pl.tr_w1 ^ 32) + 4096 * nt + 128 * kb;
      o.bh[nt] = bf8(lds_tr(w0), lds_tr(w1));
      o.bm[nt] = bf8(lds_tr(w0 + 8192), lds_tr(w1 + 8192));
      o.bl[nt] = bf8(lds_tr(w0 + 16384), lds_tr(w1 + 16384));
    }
  } else {
    const int n = lane & 31, kh = lane >> 5;
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
      const int g = ((nt * 4 + kb) * 2 + kh) * 32 + n;
      o.bh[nt] = __builtin_bit_cast(ngm_bf16x8, P[g]);
      o.bm[nt] = __builtin_bit_cast(ngm_bf16x8, P[PLANE_G + g]);
      o.bl[nt] = __builtin_bit_cast(ngm_bf16x8, P[2 * PLANE_G + g]);
    }
  }
}
template <bool W1T>
__device__ __forceinline__ void dgrad_cells(const char* tile, const char* w1c, const ngm_u32x4* P, const P3Lane& pl, int lane,
                                            f32x16 (&dX)[2]) {
  const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  TrOps o[2];
  dgrad_cells_load<W1T>(tile, w1c, P, pl, lane, 0, o[0]);
#pragma unroll
  for (int kb = 0; kb < 4; ++kb) {
    if (kb < 3) dgrad_cells_load<W1T>(tile, w1c, P, pl, lane, kb + 1, o[(kb + 1) & 1]);
    const TrOps& c = o[kb & 1];
    __builtin_amdgcn_sched_barrier(0);
#define NGM_DC_PRODUCT(PA, PB_, Z) \
    _Pragma("unroll") for (int nt = 0; nt < 2; ++nt) dX[nt] = mfma_bf16(c.PA, c.PB_[nt], (Z) ? zero : dX[nt]);
    NGM_DC_PRODUCT(al, bh, kb == 0)
    NGM_DC_PRODUCT(ah, bl, false)
    NGM_DC_PRODUCT(am, bm, false)
    NGM_DC_PRODUCT(am, bh, false)
    NGM_DC_PRODUCT(ah, bm, false)
    NGM_DC_PRODUCT(ah, bh, false)
#undef NGM_DC_PRODUCT
    __builtin_amdgcn_sched_barrier(0);
  }
}

// ---- GEMM phases "by steps": every MFMA is followed by one slice of INDEPENDENT vector work and a full scheduling fence.
// One wave per SIMD: an MFMA holds the matrix pipe 32 clocks but the issue port 4; only this wave's own instructions can use
// the rest, and only if they stand between two MFMAs in program order (the sequencer is in order: vector work behind a block
// of MFMAs overlaps with the last one alone).  sched_group_barrier pipelines proved unreliable for work whose uses are a
// phase away (instruction selection sinks pure instructions to their uses), so the order is built by construction: the
// A operand of the step is pinned (the MFMA cannot float up), the slice pins its own inputs and results.
template <class F>
__device__ __forceinline__ void wgrad_steps(const B3Op (&Ain)[2], const B3Op& B0, const B3Op& B1, f32x16 (&acc)[2][2], int j0, F&& work) {
  const B3Op A[2] = {Ain[0], Ain[1]};
  const B3Op Bx[2] = {B0, B1};
  int j = j0;
#define NGM_WS_PRODUCT(PA, PB_)                                                          \
  _Pragma("unroll") for (int mo = 0; mo < 2; ++mo)                                        \
    _Pragma("unroll") for (int mi = 0; mi < 2; ++mi) {                                    \
      acc[mo][mi] = mfma_bf16(A[mo].PA, Bx[mi].PB_, acc[mo][mi]);                         \
      asm volatile("" : "+a"(acc[mo][mi]));     /* the MFMA stays in its step: its accumulator (an AGPR tuple in this kernel) is pinned */ \
      work(j++);                                                                          \
      __builtin_amdgcn_sched_barrier(0);                                                  \
    }
  NGM_WS_PRODUCT(l, h)
  NGM_WS_PRODUCT(h, l)
  NGM_WS_PRODUCT(m, m)
  NGM_WS_PRODUCT(m, h)
  NGM_WS_PRODUCT(h, m)
  NGM_WS_PRODUCT(h, h)
#undef NGM_WS_PRODUCT
}
template <bool W1T, class F>
__device__ __forceinline__ void dgrad_cells_steps(const char* tile, const char* w1c, const ngm_u32x4* P, const P3Lane& pl, int lane,
                                                  f32x16 (&dX)[2], F&& work) {
  const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  TrOps o[2];
  dgrad_cells_load<W1T>(tile, w1c, P, pl, lane, 0, o[0]);
  int j = 0;
#pragma unroll
  for (int kb = 0; kb < 4; ++kb) {
    if (kb < 3) dgrad_cells_load<W1T>(tile, w1c, P, pl, lane, kb + 1, o[(kb + 1) & 1]);
    const TrOps& c = o[kb & 1];
    __builtin_amdgcn_sched_barrier(0);
#define NGM_DS_PRODUCT(PA, PB_, Z)                                                        \
    _Pragma("unroll") for (int nt = 0; nt < 2; ++nt) {                                    \
      dX[nt] = mfma_bf16(c.PA, c.PB_[nt], (Z) ? zero : dX[nt]);                           \
      asm volatile("" : "+a"(dX[nt]));                                                    \
      work(j++);                                                                          \
      __builtin_amdgcn_sched_barrier(0);                                                  \
    }
    NGM_DS_PRODUCT(al, bh, kb == 0)
    NGM_DS_PRODUCT(ah, bl, false)
    NGM_DS_PRODUCT(am, bm, false)
    NGM_DS_PRODUCT(am, bh, false)
    NGM_DS_PRODUCT(ah, bm, false)
    NGM_DS_PRODUCT(ah, bh, false)
#undef NGM_DS_PRODUCT
  }
}
// slice of a three-way split: one PAIR of values -> one dword of each plane (11 vector instructions), pinned to its slice
struct SplitAcc { uint32_t h[2][2][4], m[2][2][4], l[2][2][4]; };     // [feature tile][k-block][dword]
__device__ __forceinline__ void split_pair_slice(float x0, float x1, SplitAcc& S, int mt, int b, int q) {
  asm volatile("" : "+v"(x0), "+v"(x1));
  uint32_t hp, mp, lp;
  b3_split2(x0, x1, hp, mp, lp);
  asm volatile("" : "+v"(hp), "+v"(mp), "+v"(lp));
  S.h[mt][b][q] = hp; S.m[mt][b][q] = mp; S.l[mt][b][q] = lp;
}
__device__ __forceinline__ B3Op split_op(const SplitAcc& S, int mt, int b) {
  B3Op o;
  o.h = __builtin_bit_cast(ngm_bf16x8, ngm_u32x4{S.h[mt][b][0], S.h[mt][b][1], S.h[mt][b][2], S.h[mt][b][3]});
  o.m = __builtin_bit_cast(ngm_bf16x8, ngm_u32x4{S.m[mt][b][0], S.m[mt][b][1], S.m[mt][b][2], S.m[mt][b][3]});
  o.l = __builtin_bit_cast(ngm_bf16x8, ngm_u32x4{S.l[mt][b][0], S.l[mt][b][1], S.l[mt][b][2], S.l[mt][b][3]});
  return o;
}

// H2^T = H1 W1^T (48 MFMAs) with both operands read as planes -- rows of the H1 tile (A), rows of W1's cells (B) -- by steps:
// work(j), j = 0..47, is the slice of independent vector work behind MFMA j.
template <class F>
__device__ __forceinline__ void recompute_planes_steps(const char* __restrict__ w1c, const char* __restrict__ tile, int lane,
                                                       f32x16 (&Hc)[2], F&& work) {
  const int n = lane & 31, hi = lane >> 5;
  const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  struct Ops { ngm_bf16x8 ah, am, al, bh[2], bm[2], bl[2]; };
  Ops o[2];
  auto ldo = [&](int kb, Ops& d) __attribute__((always_inline)) {
    d.ah = cells_row<32>(tile, kb, hi, n);
    d.am = cells_row<32>(tile + 4096, kb, hi, n);
    d.al = cells_row<32>(tile + 8192, kb, hi, n);
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
      d.bh[nt] = cells_row<64>(w1c, kb, hi, 32 * nt + n);
      d.bm[nt] = cells_row<64>(w1c + 8192, kb, hi, 32 * nt + n);
      d.bl[nt] = cells_row<64>(w1c + 16384, kb, hi, 32 * nt + n);
    }
  };
  ldo(0, o[0]);
  __builtin_amdgcn_sched_barrier(0);
  int j = 0;
#pragma unroll
  for (int kb = 0; kb < 4; ++kb) {
    Ops& c = o[kb & 1];
    if (kb < 3) ldo(kb + 1, o[(kb + 1) & 1]);
#define NGM_HP_PRODUCT(PA, PB_, Z)                                                        \
    _Pragma("unroll") for (int nt = 0; nt < 2; ++nt) {                                    \
      Hc[nt] = mfma_bf16(c.PA, c.PB_[nt], (Z) ? zero : Hc[nt]);                           \
      asm volatile("" : "+a"(Hc[nt]));                                                    \
      work(j++);                                                                          \
      __builtin_amdgcn_sched_barrier(0);                                                  \
    }
    NGM_HP_PRODUCT(al, bh, kb == 0)
    NGM_HP_PRODUCT(ah, bl, false)
    NGM_HP_PRODUCT(am, bm, false)
    NGM_HP_PRODUCT(am, bh, false)
    NGM_HP_PRODUCT(ah, bm, false)
    NGM_HP_PRODUCT(ah, bh, false)
#undef NGM_HP_PRODUCT
  }
}
// one value of the encoding (positional_encodings.py:197-201 / 262-276): v = raw coordinate | sin | cos of w . p, rev = the
// argument in revolutions (for the cosine of the Fourier-matrix gradient); the hardware sine / cosine of k_field_bwd_b3
template <bool NEED_COS>
__device__ __forceinline__ float enc_value(const float4& w, float x, float y, float z, bool first_tile, float& rev) {
  const float inv2pi = 0.15915494309189535f;
  const float arg = fmaf(w.z, z, fmaf(w.y, y, w.x * x));
  rev = __builtin_amdgcn_fractf(arg * inv2pi);
  const float sn = __builtin_amdgcn_sinf(rev);
  float v = sn;
  if (NEED_COS) v = (w.w == NGM_FK_COS) ? __builtin_amdgcn_cosf(rev) : sn;
  if (first_tile) v = (w.w == NGM_FK_RAW) ? arg : v;                    // raw coordinates are features 0..2
  return v;
}

// (the structure of k_field_bwd_b3<2, .., HS = true>; the differences are marked "planes")
struct LdsB3pF {                                 // the same offsets in floats, for the code shared with k_field_bwd_b3
  static constexpr int CONSTS = LdsB3p::CONSTS / 4, INB = LdsB3p::INB / 4, PB = LdsB3p::PB / 4, OB = LdsB3p::OB / 4,
                       INB2 = LdsB3p::INB2 / 4, PB2 = LdsB3p::PB2 / 4, OB2 = LdsB3p::OB2 / 4, ACCL = LdsB3p::ACCL / 4, NT = LdsB3p::NT;
};
template <bool NEED_COS, bool ENC_GRAD, bool FC>
__global__ __launch_bounds__(B3B_THREADS) void k_field_bwd_b3p(FieldBwdArgs a) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  constexpr int L = 2;
  constexpr bool HS = true;
  using LY = LdsB3pF;
  char* smc = reinterpret_cast<char*>(sm);
  const int f = blockIdx.x % a.F, chunk = blockIdx.x / a.F;
  const int64_t row = a.pr.field_index ? a.pr.field_index[f] : f;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int i = lane & 31, hi = lane >> 5;
  ngm_u32x4* planes = reinterpret_cast<ngm_u32x4*>(smc + LdsB3p::W0P);      // layer 0's data-gradient planes (k_field_bwd_b3's format)
  char* w1c = smc + LdsB3p::W1C;                                             // planes: W1 as cells
  char* wlc = smc + LdsB3p::WAVES + wave * LdsB3p::WAVE_TOTAL;
  float* wl = reinterpret_cast<float*>(wlc);
  char* tile = wlc + LdsB3p::TILE;                                           // planes: the 12 KB plane tile
  const P3Lane pl = p3_lane(lane);
  float* inb = wl + LY::INB;
  float* pbuf = wl + LY::PB;
  float* obuf = wl + LY::OB;
  float* inb2 = wl + LY::INB2;
  float* pbuf2 = wl + LY::PB2;
  float* obuf2 = wl + LY::OB2;
  const uint32_t wl_lds = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)wl);
  float4* accl = reinterpret_cast<float4*>(wl + LY::ACCL) + lane;      // HS: slot q of this lane at accl[64 q]
  if constexpr (HS) {
#pragma unroll
    for (int q = 0; q < 5; ++q) accl[64 * q] = make_float4(0.f, 0.f, 0.f, 0.f);
  }

  f32x16 acc[L][2][2];
#pragma unroll
  for (int l = 0; l < L; ++l)
#pragma unroll
    for (int mo = 0; mo < 2; ++mo)
#pragma unroll
      for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[l][mo][mi][r] = 0.f;
  float dbh[L][2], dwo[2][4], dwf[2][3], dbo[4];
#pragma unroll
  for (int m = 0; m < 2; ++m) {
#pragma unroll
    for (int l = 0; l < L; ++l) dbh[l][m] = 0.f;
#pragma unroll
    for (int c = 0; c < 4; ++c) dwo[m][c] = 0.f;
    dwf[m][0] = dwf[m][1] = dwf[m][2] = 0.f;
  }
  dbo[0] = dbo[1] = dbo[2] = dbo[3] = 0.f;

  const uint32_t beg = (uint32_t)chunk * (uint32_t)a.per_block, bend = (uint32_t)min(a.P, (int64_t)beg + a.per_block);
  constexpr uint32_t TSTRIDE = 32 * B3B_WAVES;
  // this wave's tiles: tile it of ntiles starts at first + it * tstep (FC: a contiguous quarter of the block's range, last
  // tile first; else every fourth tile of the block); samples at or beyond `end` do not exist
  uint32_t first, end, ntiles;
  int32_t tstep;
  if constexpr (FC) {
    const uint32_t wr = (uint32_t)a.per_block / B3B_WAVES, wb = min(bend, beg + (uint32_t)wave * wr);
    end = min(bend, wb + wr);
    ntiles = (end - wb + 31u) >> 5;
    first = wb + 32u * (ntiles - 1u);          // unused when ntiles == 0
    tstep = -32;
  } else {
    end = bend;
    first = beg + 32u * (uint32_t)wave;
    ntiles = first < end ? (end - first + TSTRIDE - 1u) / TSTRIDE : 0u;
    tstep = (int32_t)TSTRIDE;
  }
  FieldStreams fs;
  {
    const int64_t g0 = (int64_t)f * a.P;
    fs.raytab = reinterpret_cast<const char*>(a.raytab) + 32 * (g0 / a.S);
    fs.dout = reinterpret_cast<const char*>(a.d_out + g0);
    fs.tpair = reinterpret_cast<const char*>(a.stashB + (g0 & ~(int64_t)1));
    fs.par = (uint32_t)(g0 & 1);
    fs.gb = (uint32_t)(g0 & 31);
    fs.act[0] = reinterpret_cast<const char*>(a.act) + (int64_t)f * a.act_planes_field_stride;   // planes: tiles aligned per field
    fs.act[1] = nullptr;
  }
  const char* seeds = FC ? reinterpret_cast<const char*>(a.rayseed) + 32 * (((int64_t)f * a.P) / a.S) : nullptr;
  const float inv_s = 1.0f / (float)a.S;
  // FC: the per-ray loss seeds of the tile's 32 samples, pieces 0 / 1 on the two half-waves (one more transfer per tile)
  // FC: a tile's small inputs (ray rows, stash row, (t, T) pair, seeds) into landing buffer 0 (even tiles) or 1 (odd tiles)
  auto issue_small = [&](uint32_t n0, int which) __attribute__((always_inline)) {
    const uint32_t dst = wl_lds + (which ? LY::INB2 : LY::INB) * 4;
    issue_inputs(fs, a.S, n0, end, lane, dst);
    issue_inputs(fs, a.S, n0 + 16, end, lane, dst + 1024);
    uint32_t n = n0 + (uint32_t)(lane & 31);
    if (n >= end) n = end - 1;
    dma16_so_c(seeds, 32u * (uint32_t)fdiv_idx32((int)n, inv_s, a.S) + 16u * (uint32_t)(lane >> 5), dst + 2048);
  };
  // first tile's transfers, then the per-lane constants and the weight planes while they are in flight
  if (ntiles) {
    if constexpr (FC) {
      issue_small(first, 0);
      if (ntiles > 1u) issue_small(first - 32u, 1);
    } else {
      issue_inputs(fs, a.S, first, end, lane, wl_lds + LY::INB * 4);
      issue_inputs(fs, a.S, first + 16, end, lane, wl_lds + LY::INB * 4 + 1024);
    }
    issue_plane_tile(fs.act[0], first >> 5, lane, wl_lds);
  }
  // per-feature constants (output-layer column, encoding row): LDS, re-read by the phase that needs them -- as
  // loop-long register residents they were spilled to scratch, and a scratch reload waits on vmcnt, i.e. on the DMA
  float4* cwout = reinterpret_cast<float4*>(sm + LY::CONSTS);
  float4* cenc = cwout + 64;
  if (threadIdx.x < 64) {
    const int ft = threadIdx.x, H = a.fc.dim_hidden;
    const float* W = a.pr.w[L];
    const int64_t w0 = row * a.pr.w_stride[L];
    cwout[ft] = (ft < H) ? make_float4(ngm_ldp(W, w0 + ft, a.pr.dtype), ngm_ldp(W, w0 + H + ft, a.pr.dtype),
                                       ngm_ldp(W, w0 + 2 * H + ft, a.pr.dtype), ngm_ldp(W, w0 + 3 * H + ft, a.pr.dtype))
                         : make_float4(0.f, 0.f, 0.f, 0.f);
    cenc[ft] = enc_row_of(a.fc, a.pr, row, ft);
    if constexpr (HS) sm[LY::CONSTS + 512 + ft] = (ft < H) ? ngm_ldp(a.pr.b[1], row * a.pr.b_stride[1] + ft, a.pr.dtype) : 0.f;
  }
  // FC: the global loss normalisers, as in k_stash_bwd (from the all-reduced sums, or summed here by every workgroup from
  // the forward's per-workgroup partials in k_loss_reduce's fixed order: identical everywhere, deterministic)
  __shared__ float s_red[FC ? 16 : 1][17];
  __shared__ float s_sums[NGM_NUM_LOSS_SUMS];
  if constexpr (FC) {
    if (a.loss_partials) {
      const int slot = threadIdx.x & 15, part = threadIdx.x >> 4;
      float s = 0.f;
      for (int b = part; b < a.n_partials; b += 16) s += a.loss_partials[(int64_t)b * NGM_NUM_LOSS_SUMS + slot];
      s_red[part][slot] = s;
    }
  }
  if (ENC_GRAD) build_dgrad_planes(a.fc, a.pr, row, 0, planes);
  build_w1_cells(a.fc, a.pr, row, w1c);
  __syncthreads();
  __shared__ __attribute__((aligned(16))) float s_k[8];   // FC: the five normalisers, re-read per tile (loop-long registers would spill)
  if constexpr (FC) {
    if (threadIdx.x < NGM_NUM_LOSS_SUMS) {
      float t = 0.f;
      if (a.loss_partials) {
#pragma unroll
        for (int p = 0; p < 16; ++p) t += s_red[p][threadIdx.x];
      } else t = a.loss_sums[threadIdx.x];
      s_sums[threadIdx.x] = t;
      if (blockIdx.x == 0 && a.sums_out) a.sums_out[threadIdx.x] = t;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
      const float n_m = s_sums[NGM_LS_PHOTO_CNT], n_d = s_sums[NGM_LS_DEPTH_CNT], n_t = s_sums[NGM_LS_TERM_CNT],
                  n_fs = s_sums[NGM_LS_FS_CNT], n_ts = s_sums[NGM_LS_TSDF_CNT];
      float k_photo = n_m > 0 ? a.rc.w_photometric / (3.0f * n_m) : 0.f;
      if (a.rc.photometric_mode == NGM_PHOTO_L2) k_photo = 2.0f * k_photo;    // d mean(e^2): the seed carries e, not sign(e)
      s_k[0] = k_photo;
      s_k[1] = n_d > 0 ? a.rc.w_depth / n_d : 0.f;
      s_k[2] = n_t > 0 ? a.rc.w_termination * 2.0f / n_t : 0.f;
      s_k[3] = n_fs > 0 ? a.rc.w_freespace * 2.0f / n_fs : 0.f;
      s_k[4] = n_ts > 0 ? a.rc.w_tsdf * 2.0f / n_ts : 0.f;
      if (blockIdx.x == 0) {
        if (a.loss_partials && a.counter) *a.counter += 1ull;
        if (a.loss_out) loss_values_from_sums(a.rc, s_sums, a.loss_out);
      }
    }
    __syncthreads();
  }
  float carryQ = 0.f;                 // FC: suffix value of the ray that continues into the next (= previous in memory) tile
  if constexpr (FC) {
    // the range ends with a tile, not necessarily with a ray: the recursion over the rest of the cut ray first
    if (ntiles) carryQ = comp_suffix_beyond(a, (int64_t)f * a.P, end, lane, inv_s, s_k[0], s_k[1], s_k[2]);
  }

  // lane-constant LDS offsets (floats): column element (feature 32 m + i, sample frow(r, hi)) of a tile sits at
  //   m * 1024 + (i >> 2) * 128 + (i & 3) + 4 * ((8 (r >> 2) + 4 hi + (r & 3)) ^ (i >> 2))
  int col[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) col[j] = (i >> 2) * 128 + (i & 3) + 4 * ((4 * hi + j) ^ (i >> 2));
#define COL_OFF(m, r) ((m) * 1024 + col[(r) & 3] + 32 * ((r) >> 2))

  DMA_WAIT(0);
  TICK_DECL;
  TICK(0);
  for (uint32_t it = 0; it < ntiles; ++it) {
    const uint32_t base = first + (uint32_t)((int32_t)it * tstep);
    const uint32_t nxt = base + (uint32_t)tstep;
    const bool more = it + 1u < ntiles;
    if (base + 32u > end) sanitize_plane_tile(tile, (int)(end - base), lane);   // planes: the field's last, partial tile
    // ---- inputs: lane = sample
    float* pb_c = pbuf;              // where this tile's position / gradient rows are
    float* ob_c = obuf;
    if constexpr (FC) {
      // k_stash_bwd's arithmetic, TWO tiles per pass: lanes 32..63 hold the 32 samples of this tile, lanes 0..31 those of
      // the next one (lower addresses: the wave walks back to front), i.e. 64 consecutive samples as in k_stash_bwd's
      // steps; the next tile's rows wait in the second buffers.  Odd tiles only pick those up.
      if (it & 1u) { pb_c = pbuf2; ob_c = obuf2; }
      else {
        const int j = i, h = j >> 4, jj = j & 15;
        const float4* blk = reinterpret_cast<const float4*>(hi ? inb : inb2);
        const float4* in4 = blk + 64 * h;
        const float4 r0 = in4[jj], r1 = in4[16 + jj], dd = in4[32 + jj], sp = in4[48 + jj];
        const float4 q0 = blk[128 + j], q1 = blk[160 + j];
        const uint32_t n = (hi ? base : base - 32u) + (uint32_t)j;     // the lower half only counts when there is a next tile
        const bool valid = hi ? (n < end) : more;
        const uint32_t nc = (hi && !valid) ? end - 1 : n;
        const bool odd = ((nc + fs.par) & 1u) != 0u;
        const float t = odd ? sp.z : sp.x, T = odd ? sp.w : sp.y;
        const float x = fmaf(t, r0.w, r0.x), y = fmaf(t, r1.x, r0.y), z = fmaf(t, r1.y, r0.z);
        const float4 kn = *reinterpret_cast<const float4*>(s_k);
        const float k_photo = kn.x, k_depth = kn.y, k_term = kn.z, k_fs = kn.w, k_ts = s_k[4];
        const int rayi = fdiv_idx32((int)nc, inv_s, a.S);        // nc < 2^24 (the API fuses only then)
        const int k = (int)nc - rayi * a.S, kr = a.S - 1 - k;
        const float dzc = r1.z, gt = r1.w, geom = dd.w;
        const float dC0 = k_photo * q0.x, dC1 = k_photo * q0.y, dC2 = k_photo * q0.z, dD = k_depth * q0.w, dT = k_term * q1.x;
        const float depth = -(dzc * t);
        float dodg = 0.f;
        const float occ = occ_pointwise_fast(a.rc.geometry_mode, a.rc.geometry_factor, geom, &dodg);
        const float ak = dC0 * dd.x + dC1 * dd.y + dC2 * dd.z + dD * depth + dT;
        float A = valid ? ak * occ : 0.f, B = valid ? 1.0f - occ : 1.0f;
        const int krv = valid ? kr : 0;
        seg_rscan_affine64(A, B, krv, lane);
        const float Qend = (valid && kr > 63 - lane) ? carryQ : 0.f;
        const float nA = lane_next(A, 0.f), nB = lane_next(B, 1.f);
        const float Qk = (krv >= 1) ? fmaf(nB, Qend, nA) : Qend;
        carryQ = lane_value(fmaf(B, Qend, A), 0);
        const float tau = a.rc.truncation_distance, cf = a.rc.color_factor;
        const float w = occ * T;
        float dg = T * (ak - Qk) * dodg;
        const float thr = (gt - tau) * (gt != 0.0f ? 1.0f : 0.0f);
        if (t < thr) dg += k_fs * (geom * tau - tau) * tau;
        const float dl = gt - t;
        if (fabsf(dl) < tau && gt != 0.0f) dg += k_ts * (geom * tau - dl) * tau;
        if (a.rc.overwrite_behind_camera && dzc * t > 0.f) dg = 0.f;      // overwritten sample: no gradient reaches the MLP output
        const float4 dout = valid ? make_float4(cf * w * dC0, cf * w * dC1, cf * w * dC2, dg) : make_float4(0.f, 0.f, 0.f, 0.f);
        WAVE_SYNC();
        *reinterpret_cast<float4*>((hi ? pbuf : pbuf2) + 4 * j) = make_float4(x, y, z, 0.f);
        *reinterpret_cast<float4*>((hi ? obuf : obuf2) + 4 * j) = dout;
        dbo[0] += dout.x; dbo[1] += dout.y; dbo[2] += dout.z; dbo[3] += dout.w;
        WAVE_SYNC();
      }
    } else {
      // (both halves compute, half 0 stores)
      const int j = i, h = j >> 4, jj = j & 15;
      const float4* in4 = reinterpret_cast<const float4*>(inb) + 64 * h;
      const float4 r0 = in4[jj], r1 = in4[16 + jj], dd = in4[32 + jj], sp = in4[48 + jj];
      const uint32_t n = base + (uint32_t)j;
      const bool valid = n < end;
      const uint32_t nc = valid ? n : end - 1;
      const float t = ((nc + fs.par) & 1u) ? sp.z : sp.x;
      const float x = fmaf(t, r0.w, r0.x), y = fmaf(t, r1.x, r0.y), z = fmaf(t, r1.y, r0.z);
      const float4 dout = valid ? dd : make_float4(0.f, 0.f, 0.f, 0.f);
      WAVE_SYNC();
      if (hi == 0) {
        *reinterpret_cast<float4*>(pbuf + 4 * j) = make_float4(x, y, z, 0.f);
        *reinterpret_cast<float4*>(obuf + 4 * j) = dout;
        dbo[0] += dout.x; dbo[1] += dout.y; dbo[2] += dout.z; dbo[3] += dout.w;
      }
      WAVE_SYNC();
    }
    TICK(1);
    // ---- phase A: H2 = relu(W1 H1 + b1) recomputed transposed, both operands READ as planes (no split).  In the shadows of
    // its 48 MFMAs: dh = Wout^T d_out (what the output layer's gradient needs and H2 does not enter: 8 fma per sample) and, for
    // the networks without a Fourier matrix, the tile's encoding (with one, the encoding is evaluated where it is split: phase F)
    f32x16 dY[2], Hc[2];
    B3Op Xp[2][2];                   // layer 1's input columns as weight-gradient operands (feature tile, k-block), straight from LDS
    float Eb[2][2][8];               // (no Fourier matrix) the tile's encoding, weight-gradient operand layout
    const float4 encw[2] = {cenc[i], cenc[32 + i]};
    {
      const float4 wout[2] = {cwout[i], cwout[32 + i]};
      float4 dq[2], pq[2];
      dq[0] = *reinterpret_cast<const float4*>(ob_c + 4 * (4 * hi));
      if constexpr (!ENC_GRAD) pq[0] = *reinterpret_cast<const float4*>(pb_c + 4 * (4 * hi));
      recompute_planes_steps(w1c, tile, lane, Hc, [&](int j) __attribute__((always_inline)) {
        const int kb = j / 12, st = j % 12;
        if (st == 3 || st == 6 || st == 9 || st == 11) {                 // dh of sample r, both feature tiles
          const int r = 4 * kb + (st == 3 ? 0 : st == 6 ? 1 : st == 9 ? 2 : 3);
          float4 d = dq[r & 1];
          asm volatile("" : "+v"(d.x));
#pragma unroll
          for (int m = 0; m < 2; ++m) {
            float dh = fmaf(wout[m].w, d.w, fmaf(wout[m].z, d.z, fmaf(wout[m].y, d.y, wout[m].x * d.x)));
            asm volatile("" : "+v"(dh));
            dY[m][r] = dh;
          }
        }
        if ((st == 4 || st == 7 || st == 10 || st == 0) && !(kb == 0 && st == 0)) {   // the next d_out row: a slice ahead
          const int r = 4 * kb + (st == 4 ? 1 : st == 7 ? 2 : st == 10 ? 3 : 0);
          if (r < 16) dq[r & 1] = *reinterpret_cast<const float4*>(ob_c + 4 * (8 * (r >> 2) + 4 * hi + (r & 3)));
        }
        if constexpr (!ENC_GRAD) {                                        // encoding: sample r = 4 kb + st / 3 (two feature tiles: st % 3 = 0, 1)
          if (st % 3 != 2 && st < 12) {
            const int r = 4 * kb + st / 3, m = st % 3;
            float4 p = pq[r & 1];
            asm volatile("" : "+v"(p.x));
            float rev;
            float v = enc_value<NEED_COS>(encw[m], p.x, p.y, p.z, m == 0, rev);
            asm volatile("" : "+v"(v));
            Eb[r >> 3][m][r & 7] = v;
          }
          if (st % 3 == 2 && 4 * kb + st / 3 + 1 < 16) {
            const int r = 4 * kb + st / 3 + 1;
            pq[r & 1] = *reinterpret_cast<const float4*>(pb_c + 4 * (8 * (r >> 2) + 4 * hi + (r & 3)));
          }
        }
      });
      TICK(6);
#pragma unroll
      for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int b = 0; b < 2; ++b) {
          const char* c0 = tile + pl.col_h1 + 2048 * m + 128 * b;
          const char* c1 = tile + (pl.col_h1 ^ 64) + 2048 * m + 128 * b;
          Xp[m][b].h = bf8(lds_tr(c0), lds_tr(c1));
          Xp[m][b].m = bf8(lds_tr(c0 + 4096), lds_tr(c1 + 4096));
          Xp[m][b].l = bf8(lds_tr(c0 + 8192), lds_tr(c1 + 8192));
        }
      const float b1v[2] = {sm[LY::CONSTS + 512 + i], sm[LY::CONSTS + 512 + 32 + i]};
      const float2 a2 = reinterpret_cast<const float2*>(accl + 128)[0];
      dbh[1][0] = a2.x; dbh[1][1] = a2.y;
      __builtin_amdgcn_sched_barrier(0);
      // ---- phase B (exposed): bias, ReLU, dY1 = relu'(H2) dh, hidden-bias gradient.  (The output-WEIGHT gradient d_out x H2
      // does not feed dY1: deferred into the shadows of layer 1's data gradient, phase C.)
#pragma unroll
      for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float h = fmaxf(Hc[m][r] + b1v[m], 0.f);
          Hc[m][r] = h;
          const float g = (h > 0.f) ? dY[m][r] : 0.f;
          dY[m][r] = g;
          dbh[1][m] += g;
        }
      reinterpret_cast<float2*>(accl + 128)[0] = make_float2(dbh[1][0], dbh[1][1]);
    }
    __builtin_amdgcn_sched_barrier(0);
    TICK(4);
    // ---- phase B2 (exposed): dY1 is split ONCE (lane = feature: the weight gradient's A operands); the planes go to the tile
    // as cells and come back transposed (lane = sample) as the data gradient's A operand
    WAVE_SYNC();                       // every read of the H1 planes has been issued (in-order LDS queue): overwrite them
    { const B3Op t0[2] = {b3_regs<0>(dY[0]), b3_regs<0>(dY[1])}; dy_store_cells(tile, lane, 0, t0); }
    { const B3Op t1[2] = {b3_regs<1>(dY[0]), b3_regs<1>(dY[1])}; dy_store_cells(tile, lane, 1, t1); }
    WAVE_SYNC();                       // (the weight gradient re-reads its own cells in phase E: the planes need not stay in registers)
    // ---- phase C: layer 1's data gradient (48 MFMAs, both operands transposed reads) with the output-weight gradient in its
    // shadows: one sample per slice (8 fma), every third step
    f32x16 dX[2];
    {
      const float4 a0 = accl[0], a1 = accl[64];
      dwo[0][0] = a0.x; dwo[0][1] = a0.y; dwo[0][2] = a0.z; dwo[0][3] = a0.w;
      dwo[1][0] = a1.x; dwo[1][1] = a1.y; dwo[1][2] = a1.z; dwo[1][3] = a1.w;
      float4 dq[2];                    // the d_out row of the slice in work / of the next one (re-read from LDS: 64 registers less)
      dq[0] = *reinterpret_cast<const float4*>(ob_c + 4 * (4 * hi));
      dgrad_cells_steps<true>(tile, w1c, nullptr, pl, lane, dX, [&](int j) __attribute__((always_inline)) {
        if (j % 3 == 1 && j / 3 < 15) {
          const int r = j / 3 + 1;
          dq[r & 1] = *reinterpret_cast<const float4*>(ob_c + 4 * (8 * (r >> 2) + 4 * hi + (r & 3)));
        }
        if (j % 3 == 0) {
          const int r = j / 3;
          float4 d = dq[r & 1];
          asm volatile("" : "+v"(d.x), "+v"(d.y), "+v"(d.z), "+v"(d.w));
#pragma unroll
          for (int m = 0; m < 2; ++m) {
            const float h = Hc[m][r];
            dwo[m][0] = fmaf(d.x, h, dwo[m][0]); dwo[m][1] = fmaf(d.y, h, dwo[m][1]);
            dwo[m][2] = fmaf(d.z, h, dwo[m][2]); dwo[m][3] = fmaf(d.w, h, dwo[m][3]);
            asm volatile("" : "+v"(dwo[m][0]), "+v"(dwo[m][1]), "+v"(dwo[m][2]), "+v"(dwo[m][3]));
          }
        }
      });
      accl[0] = make_float4(dwo[0][0], dwo[0][1], dwo[0][2], dwo[0][3]);
      accl[64] = make_float4(dwo[1][0], dwo[1][1], dwo[1][2], dwo[1][3]);
    }
    TICK(7);
    // ---- phase D (exposed): ReLU mask of layer 0 (from the hi plane of H1: post-ReLU, so positive <=> leading bf16 non-zero)
    {
      const float2 b0 = reinterpret_cast<const float2*>(accl + 128)[1];
      dbh[0][0] = b0.x; dbh[0][1] = b0.y;
#pragma unroll
      for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const uint32_t hw = __builtin_bit_cast(ngm_u32x4, Xp[m][r >> 3].h)[(r & 7) >> 1];
          const bool pos = (hw & ((r & 1) ? 0xffff0000u : 0x0000ffffu)) != 0u;
          const float g = pos ? dX[m][r] : 0.f;
          dY[m][r] = g;
          dbh[0][m] += g;
        }
      reinterpret_cast<float2*>(accl + 128)[1] = make_float2(dbh[0][0], dbh[0][1]);
    }
    __builtin_amdgcn_sched_barrier(0);
    // ---- phase E: layer 1's weight gradient (48 MFMAs, operands in registers) with the ONE split of dY0 in its shadows (16
    // pair slices; without a Fourier matrix there is no phase F and the encoding's split rides here as well)
    SplitAcc S0, SE;
    {
      auto work = [&](int j) __attribute__((always_inline)) {
        if (ENC_GRAD) {
          if (j % 3 == 0) { const int p = j / 3, m = p >> 3, b = (p >> 2) & 1, q = p & 3; split_pair_slice(dY[m][8 * b + 2 * q], dY[m][8 * b + 2 * q + 1], S0, m, b, q); }
        } else {
          if (j % 3 == 0) { const int p = j / 3, m = p >> 3, b = (p >> 2) & 1, q = p & 3; split_pair_slice(dY[m][8 * b + 2 * q], dY[m][8 * b + 2 * q + 1], S0, m, b, q); }
          if (j % 3 == 1) { const int p = j / 3, m = p >> 3, b = (p >> 2) & 1, q = p & 3; split_pair_slice(Eb[b][m][2 * q], Eb[b][m][2 * q + 1], SE, m, b, q); }
        }
      };
      B3Op A0[2], A1[2];
      dy_load_cells(tile, lane, 0, A0);
      dy_load_cells(tile, lane, 1, A1);
      __builtin_amdgcn_sched_barrier(0);
      wgrad_steps(A0, Xp[0][0], Xp[1][0], acc[1], 0, work);
      wgrad_steps(A1, Xp[0][1], Xp[1][1], acc[1], 24, work);
    }
    B3Op Y0[2][2];
#pragma unroll
    for (int m = 0; m < 2; ++m) { Y0[m][0] = split_op(S0, m, 0); Y0[m][1] = split_op(S0, m, 1); }
    TICK(9);
    // ---- phase F (Fourier): dY0's cells; layer 0's data gradient (48 MFMAs, A transposed from the cells, B = W0's planes) with
    // the ENCODING in its shadows: a slice evaluates one pair of sines and splits it (no fp32 copy of the encoding is kept)
    f32x16 dE[2];
    float px[16], py[16], pz[16];      // this lane's 16 sample positions: phase F (arguments), phase G (Fourier-matrix gradient)
    if constexpr (ENC_GRAD) {
      WAVE_SYNC();
      { const B3Op t0[2] = {Y0[0][0], Y0[1][0]}, t1[2] = {Y0[0][1], Y0[1][1]}; dy_store_cells(tile, lane, 0, t0); dy_store_cells(tile, lane, 1, t1); }
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float4 p = *reinterpret_cast<const float4*>(pb_c + 4 * (8 * (r >> 2) + 4 * hi + (r & 3)));
        px[r] = p.x; py[r] = p.y; pz[r] = p.z;
      }
      WAVE_SYNC();
      dgrad_cells_steps<false>(tile, nullptr, planes, pl, lane, dE, [&](int j) __attribute__((always_inline)) {
        if (j % 3 == 0) {
          const int p = j / 3, m = p >> 3, b = (p >> 2) & 1, q = p & 3, r = 8 * b + 2 * q;
          float x0 = px[r], x1 = px[r + 1];
          asm volatile("" : "+v"(x0), "+v"(x1));
          float rev;
          const float v0 = enc_value<NEED_COS>(encw[m], x0, py[r], pz[r], m == 0, rev);
          const float v1 = enc_value<NEED_COS>(encw[m], x1, py[r + 1], pz[r + 1], m == 0, rev);
          split_pair_slice(v0, v1, SE, m, b, q);
        }
      });
    }
    TICK(8);
    B3Op EbP[2][2];
#pragma unroll
    for (int m = 0; m < 2; ++m) { EbP[m][0] = split_op(SE, m, 0); EbP[m][1] = split_op(SE, m, 1); }
    // ---- phase G: the next tile's transfers, then layer 0's weight gradient (48 MFMAs, operands in registers) with the
    // Fourier-matrix gradient d sin(w.x)/d w = cos(w.x) x in its shadows (one (sample, feature tile) per slice, the cosine
    // re-evaluated from the position: keeping 32 of them from phase F spilled).  No LDS instruction between the transfers'
    // issue and their wait: a wave's LDS instructions crawl while it has HBM -> LDS transfers it has not waited for.
    if constexpr (ENC_GRAD) {
      const float4 f0 = accl[192], f1 = accl[256];
      dwf[0][0] = f0.x; dwf[0][1] = f0.y; dwf[0][2] = f0.z;
      dwf[1][0] = f1.x; dwf[1][1] = f1.y; dwf[1][2] = f1.z;
    }
    WAVE_SYNC();
    TICK(3);
    if constexpr (FC) {
      // the small inputs of the tile after next: its landing buffer (0 for even tiles, 1 for odd ones) was consumed by the
      // pass at the top of this tile (even) or of the previous one (odd)
      if (it + 2u < ntiles) issue_small(base - 64u, (int)(it & 1u));
    }
    if (more) {
      if constexpr (!FC) {
        issue_inputs(fs, a.S, nxt, end, lane, wl_lds + LY::INB * 4);
        issue_inputs(fs, a.S, nxt + 16, end, lane, wl_lds + LY::INB * 4 + 1024);
      }
      issue_plane_tile(fs.act[0], nxt >> 5, lane, wl_lds);       // 12 linear 1 KB copies, tiles aligned per field
    }
    __builtin_amdgcn_sched_barrier(0);
    TICK(5);
    {
      auto work = [&](int j) __attribute__((always_inline)) {
        if (ENC_GRAD && j % 3 != 2) {
          const int r = j / 3, m = j % 3;
          float x = px[r];
          asm volatile("" : "+v"(x));
          const float4 w = encw[m];
          const float rev = __builtin_amdgcn_fractf(fmaf(w.z, pz[r], fmaf(w.y, py[r], w.x * x)) * 0.15915494309189535f);
          const float g = dE[m][r] * __builtin_amdgcn_cosf(rev);
          dwf[m][0] = fmaf(g, x, dwf[m][0]); dwf[m][1] = fmaf(g, py[r], dwf[m][1]); dwf[m][2] = fmaf(g, pz[r], dwf[m][2]);
          asm volatile("" : "+v"(dwf[m][0]), "+v"(dwf[m][1]), "+v"(dwf[m][2]));
        }
      };
      const B3Op A0[2] = {Y0[0][0], Y0[1][0]}, A1[2] = {Y0[0][1], Y0[1][1]};
      wgrad_steps(A0, EbP[0][0], EbP[1][0], acc[0], 0, work);
      wgrad_steps(A1, EbP[0][1], EbP[1][1], acc[0], 24, work);
    }
    __builtin_amdgcn_sched_barrier(0);
    TICK(10);
    DMA_WAIT(0);
    TICK(2);
    if constexpr (ENC_GRAD) {
      accl[192] = make_float4(dwf[0][0], dwf[0][1], dwf[0][2], 0.f);
      accl[256] = make_float4(dwf[1][0], dwf[1][1], dwf[1][2], 0.f);
    }
    WAVE_SYNC();
  }
#undef COL_OFF
  if constexpr (HS) {                 // the per-feature sums back into registers before the staging area takes all of LDS
    const float4 a0 = accl[0], a1 = accl[64], a2 = accl[128], f0 = accl[192], f1 = accl[256];
    dwo[0][0] = a0.x; dwo[0][1] = a0.y; dwo[0][2] = a0.z; dwo[0][3] = a0.w;
    dwo[1][0] = a1.x; dwo[1][1] = a1.y; dwo[1][2] = a1.z; dwo[1][3] = a1.w;
    dbh[L - 1][0] = a2.x; dbh[L - 1][1] = a2.y; dbh[0][0] = a2.z; dbh[0][1] = a2.w;
    dwf[0][0] = f0.x; dwf[0][1] = f0.y; dwf[0][2] = f0.z;
    dwf[1][0] = f1.x; dwf[1][1] = f1.y; dwf[1][2] = f1.z;
  }
  __syncthreads();

  // ---- epilogue: the four waves' accumulators are summed in fixed wave order (all of LDS is free now)
  float* stage = sm;
  constexpr int NT = LY::NT;
  // staging layout [wave][tile][q = r >> 2][lane][4 floats]: 16-byte LDS accesses on both sides
#pragma unroll
  for (int l = 0; l < L; ++l)
#pragma unroll
    for (int mo = 0; mo < 2; ++mo)
#pragma unroll
      for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const f32x16& c = acc[l][mo][mi];
          *reinterpret_cast<float4*>(stage + (((wave * NT + (l * 4 + mo * 2 + mi)) * 4 + q) * 64 + lane) * 4) =
              make_float4(c[4 * q], c[4 * q + 1], c[4 * q + 2], c[4 * q + 3]);
        }
  __syncthreads();
  int64_t enc_off, w_off[NGM_MAX_LAYERS + 1], b_off[NGM_MAX_LAYERS + 1];
  (void)ngm_param_offsets(&a.fc, &enc_off, w_off, b_off);
  float* dst = a.partials + (int64_t)blockIdx.x * a.p_pad;
  const int D = a.fc.dim_enc, H = a.fc.dim_hidden;
  for (int e4 = threadIdx.x; e4 < NT * 256; e4 += B3B_THREADS) {
    float4 s4 = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int w = 0; w < B3B_WAVES; ++w) {                           // fixed wave order: deterministic
      const float4 v = *reinterpret_cast<const float4*>(stage + (w * NT * 256 + e4) * 4);
      s4.x += v.x; s4.y += v.y; s4.z += v.z; s4.w += v.w;
    }
    const int t = e4 >> 8, q = (e4 >> 6) & 3, ln = e4 & 63;
    const int l = t >> 2, mo = (t >> 1) & 1, mi = t & 1;
    const int o0 = 32 * mo + 8 * q + 4 * (ln >> 5), c = 32 * mi + (ln & 31), din = (l == 0) ? D : H;   // rows frow(4q + j, hi) = o0 + j
    if (c < din) {
      float* d = dst + w_off[l] + (int64_t)o0 * din + c;
      if (o0 < H) d[0] = s4.x;
      if (o0 + 1 < H) d[din] = s4.y;
      if (o0 + 2 < H) d[2 * din] = s4.z;
      if (o0 + 3 < H) d[3 * din] = s4.w;
    }
  }
  __syncthreads();
  // per-feature vectors: lane (i, hi) holds the partial sums of feature 32 m + i over its half of the samples
  constexpr int NV = 2 * L + 8 + 6 + 4;
  {
    float* sw = stage + wave * NV * 64;
    int k = 0;
#pragma unroll
    for (int l = 0; l < L; ++l)
#pragma unroll
      for (int m = 0; m < 2; ++m) sw[(k++) * 64 + lane] = dbh[l][m];
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
      for (int c = 0; c < 4; ++c) sw[(k++) * 64 + lane] = dwo[m][c];
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
      for (int c = 0; c < 3; ++c) sw[(k++) * 64 + lane] = dwf[m][c];
#pragma unroll
    for (int c = 0; c < 4; ++c) sw[(k++) * 64 + lane] = wave_sum(dbo[c]);
  }
  __syncthreads();
  const bool fourier = ENC_GRAD && a.fc.encoding == NGM_ENC_FOURIER;
  const int n_raw = a.fc.raw_coords ? 3 : 0;
  for (int e = threadIdx.x; e < NV * 32; e += B3B_THREADS) {
    const int k = e >> 5, ii = e & 31;
    float s0 = 0.f;
#pragma unroll
    for (int w = 0; w < B3B_WAVES; ++w) s0 += stage[(w * NV + k) * 64 + ii] + stage[(w * NV + k) * 64 + 32 + ii];
    if (k < 2 * L) {
      const int l = k >> 1, ft = 32 * (k & 1) + ii;
      if (ft < H) dst[b_off[l] + ft] = s0;
    } else if (k < 2 * L + 8) {
      const int u = k - 2 * L, ft = 32 * (u >> 2) + ii, c = u & 3;
      if (ft < H) dst[w_off[L] + (int64_t)c * H + ft] = s0;
    } else if (k < 2 * L + 14) {
      const int u = k - 2 * L - 8, ft = 32 * (u / 3) + ii, c = u % 3;
      if (fourier && ft < D && ft >= n_raw) dst[enc_off + (int64_t)(ft - n_raw) * 3 + c] = s0;
    } else if (ii == 0) {
      dst[b_off[L] + (k - 2 * L - 14)] = 0.5f * s0;     // wave_sum put the total into every lane: both halves counted it
    }
  }
  if (!fourier)   // the encoding slot of the partial vector (if any) carries no gradient
    for (int64_t p = enc_off + threadIdx.x; p < w_off[0]; p += B3B_THREADS) dst[p] = 0.f;
  TICK(11);
  TICK_REPORT
}

// ------------------------------------------------------------------------------------------------
bool ngm_field_bwd_b3p_compiled(const ngm_field_cfg* fc) {
  const int MI = (fc->dim_enc + 31) / 32, MH = (fc->dim_hidden + 31) / 32;
  if (fc->skip_mode != NGM_SKIP_NO || fc->matmul_mode == NGM_MATMUL_F32 || MI != 2 || MH != 2 || fc->num_layers != 2) return false;
  return fc->encoding == NGM_ENC_FOURIER || fc->encoding == NGM_ENC_NERF || fc->encoding == NGM_ENC_NONE;
}

// returns NGM_E_UNSUPPORTED when this variant does not apply
int ngm_launch_field_bwd_b3p(const FieldBwdArgs& a, int blocks, hipStream_t st) {
  if (a.act_half != 2 || !a.act || a.points || !ngm_field_bwd_b3p_compiled(&a.fc)) return NGM_E_UNSUPPORTED;
  if ((a.P + 64) * 384 >= ((int64_t)1 << 32)) return NGM_E_UNSUPPORTED;
  if (a.per_block % (B3B_WAVES * 32)) return NGM_E_INVALID;
  if (a.fused_comp && !a.rayseed) return NGM_E_INVALID;
  NgmProfScope prof_(NGM_K_FIELD_BWD, st);
  const size_t lds = (size_t)LdsB3p::TOTAL;
  static_assert(LdsB3p::TOTAL <= 160 * 1024, "LDS plan of k_field_bwd_b3p");
#define NGM_LBP(NC, EG)                                                                                              \
  do {                                                                                                               \
    if (a.fused_comp) {                                                                                              \
      (void)hipFuncSetAttribute((const void*)k_field_bwd_b3p<NC, EG, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);  \
      hipLaunchKernelGGL((k_field_bwd_b3p<NC, EG, true>), dim3(blocks), dim3(B3B_THREADS), lds, st, a);              \
    } else {                                                                                                         \
      (void)hipFuncSetAttribute((const void*)k_field_bwd_b3p<NC, EG, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
      hipLaunchKernelGGL((k_field_bwd_b3p<NC, EG, false>), dim3(blocks), dim3(B3B_THREADS), lds, st, a);             \
    }                                                                                                                \
  } while (0)
  if (a.fc.encoding == NGM_ENC_FOURIER) NGM_LBP(false, true);
  else if (a.fc.encoding == NGM_ENC_NERF) NGM_LBP(true, false);
  else NGM_LBP(false, false);
#undef NGM_LBP
  return 0;
}
