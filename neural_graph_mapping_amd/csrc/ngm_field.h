// Per-field tiny-MLP machinery shared by the forward and backward kernels:
//   * LDS-resident weights of ONE field per workgroup, stored as MFMA A-fragments,
//   * positional encoding evaluated straight into MFMA B-operand registers,
//   * layer chaining on v_mfma_f32_32x32x2_f32 with activations kept in registers.
// Template parameters: MI = ceil(D/32) encoding tiles, MH = ceil(H/32) hidden tiles, L = hidden layers.
#pragma once
#include "ngm_device.h"

// encoding feature descriptor kinds (LDS table encW[f] = {wx, wy, wz, kind})
#define NGM_FK_ZERO 0.0f
#define NGM_FK_RAW 1.0f   // value = w . x       (raw coordinates: unit weight vector)
#define NGM_FK_SIN 2.0f   // value = sin(w . x)
#define NGM_FK_COS 3.0f   // value = cos(w . x)

// CAT: skip_mode "concat" (models.py:159-161): every layer after the first (and the output layer) reads
// cat(hidden, encoding): MH + MI input tiles, the encoding part starting at tile column 32 * MH.
template <int MI, int MH, int L, bool CAT = false>
struct FieldLds {
  static constexpr int MIN(int l) { return l == 0 ? MI : (CAT ? MH + MI : MH); }   // input tiles of hidden layer l
  static constexpr int MOUT_IN = CAT ? MH + MI : MH;                                // input tiles of the output layer
  static constexpr int ENCW = 0;                             // float4[MI*32]
  static constexpr int ENCW_SIZE = MI * 32 * 4;
  static constexpr int w_off(int l) {                        // hidden layer l weights
    int o = ENCW + ENCW_SIZE;
    for (int i = 0; i < l; ++i) o += wfrag_size(MH, MIN(i)) + MH * 32;
    return o;
  }
  static constexpr int b_off(int l) { return w_off(l) + wfrag_size(MH, MIN(l)); }
  static constexpr int WOUT = b_off(L - 1) + MH * 32;         // float4[MOUT_IN*32]: the 4 output weights per input feature
  static constexpr int BOUT = WOUT + MOUT_IN * 32 * 4;        // float[4]
  static constexpr int TOTAL = (BOUT + 4 + 3) & ~3;           // floats, 16-byte multiple
};

struct FieldDims {
  int D, H, L, n_raw, n_feat;  // logical sizes; Fourier: n_raw = 3 if raw_coords, n_feat rows of enc_w
};

// Two-phase staging of one field's parameters for kernels that have other latency-bound work to do while the
// weights travel: issue() only starts the global loads (straight-line code: everything is in flight at once;
// 16-byte loads for the matrices when the rows allow it), commit() permutes them into the A-fragment order of the
// forward kernels (FieldLds) and builds the per-feature encoding table.
// Needs blockDim.x >= 256 (>= 32*MH threads for the small tables); caller must __syncthreads() after commit().
template <int MI, int MH, int L, bool CAT = false>
struct FieldStage {
  using LYS = FieldLds<MI, MH, L, CAT>;
  static constexpr int NIT = (MH * 32 * (CAT ? MH + MI : MH) * 32 / 4 + 255) / 256;   // 16-byte chunks per thread and layer (upper bound)
  float4 v[L][NIT];
  float4 enc0, enc1, wout, wout_e;
  float bias[L], bout;
  __device__ __forceinline__ void issue(const ngm_field_cfg& fc, const ngm_params& pr, int64_t row) {
    const int tid = threadIdx.x, nthr = blockDim.x;
    const int D = fc.dim_enc, H = fc.dim_hidden;
#pragma unroll
    for (int l = 0; l < L; ++l) {
      constexpr int MIN = LYS::MIN(1);           // l >= 1 (l == 0 handled through the ternaries below)
      const int min_l = (l == 0) ? MI : MIN;
      const bool cat_l = CAT && l > 0;           // input = cat(hidden (H), encoding (D)): logical row length H + D
      const int Din = (l == 0) ? D : (cat_l ? H + D : H);
      const int dt = pr.dtype;
      const float* W = pr.w[l];                          // element offsets from here: the storage may be 16-bit
      const int64_t w0 = row * pr.w_stride[l];
      const int ncol4 = min_l * 8, total4 = MH * 32 * ncol4;
      const uintptr_t wa = reinterpret_cast<uintptr_t>(W) + (uintptr_t)w0 * (dt == NGM_DT_F32 ? 4 : 2);
      const bool vec = ((Din & 3) == 0) && (!cat_l || (H & 3) == 0) && ((wa & (dt == NGM_DT_F32 ? 15 : 7)) == 0);
#pragma unroll
      for (int it = 0; it < NIT; ++it) {
        const int e4 = tid + it * nthr;
        float4 x = make_float4(0.f, 0.f, 0.f, 0.f);
        if (e4 < total4) {
          const int o = e4 / ncol4, ct = 4 * (e4 - o * ncol4);      // tile column
          // concat: tile columns [0, 32 MH) = hidden features, [32 MH, ...) = encoding features (logical column H + .)
          const bool enc_part = cat_l && ct >= 32 * MH;
          const int c = enc_part ? H + (ct - 32 * MH) : ct;          // logical column
          const int lim = enc_part ? H + D : (cat_l ? H : Din);      // end of this part's logical columns
          if (o < H && c < lim) x = ngm_ldp4(W, w0 + (int64_t)o * Din + c, dt, vec, lim - c);
        }
        v[l][it] = x;
      }
      bias[l] = (tid < H) ? ngm_ldp(pr.b[l], row * pr.b_stride[l] + tid, dt) : 0.f;
    }
    const int dt = pr.dtype;
    // output layer (4 x H) -> float4 per hidden feature
    wout = make_float4(0.f, 0.f, 0.f, 0.f);
    wout_e = make_float4(0.f, 0.f, 0.f, 0.f);
    {
      const int Dout_in = CAT ? H + D : H;                 // row length of the (4, .) output matrix
      const float* W = pr.w[L];
      const int64_t w0 = row * pr.w_stride[L];
      if (tid < H) wout = make_float4(ngm_ldp(W, w0 + tid, dt), ngm_ldp(W, w0 + Dout_in + tid, dt),
                                      ngm_ldp(W, w0 + 2 * Dout_in + tid, dt), ngm_ldp(W, w0 + 3 * Dout_in + tid, dt));
      if (CAT && tid < D) wout_e = make_float4(ngm_ldp(W, w0 + H + tid, dt), ngm_ldp(W, w0 + Dout_in + H + tid, dt),
                                               ngm_ldp(W, w0 + 2 * Dout_in + H + tid, dt), ngm_ldp(W, w0 + 3 * Dout_in + H + tid, dt));
    }
    bout = (tid < 4) ? ngm_ldp(pr.b[L], row * pr.b_stride[L] + tid, dt) : 0.f;
    // encoding table: one float4 per feature (weights of the argument, kind)
    enc0 = make_float4(0.f, 0.f, 0.f, NGM_FK_ZERO);
    enc1 = make_float4(0.f, 0.f, 0.f, 0.f);
    if (fc.encoding == NGM_ENC_PERMUTO) {
      enc0.w = 0.f;
      if (tid < fc.nr_levels) {
        const float* hs = pr.shift + row * pr.shift_stride + 3 * tid;
        enc0 = make_float4(fc.level_scale[3 * tid], fc.level_scale[3 * tid + 1], fc.level_scale[3 * tid + 2], 0.f);
        enc1 = make_float4(hs[0], hs[1], hs[2], 0.f);
      }
    } else if (tid < D) {
      const int f = tid;
      if (fc.encoding == NGM_ENC_FOURIER) {
        const int n_raw = fc.raw_coords ? 3 : 0;
        if (f < n_raw) {
          enc0 = make_float4(f == 0 ? 1.f : 0.f, f == 1 ? 1.f : 0.f, f == 2 ? 1.f : 0.f, NGM_FK_RAW);
        } else {
          const int64_t e0 = row * pr.enc_w_stride + (int64_t)(f - n_raw) * 3;
          enc0 = make_float4(ngm_ldp(pr.enc_w, e0, dt), ngm_ldp(pr.enc_w, e0 + 1, dt), ngm_ldp(pr.enc_w, e0 + 2, dt), NGM_FK_SIN);
        }
      } else if (fc.encoding == NGM_ENC_NERF) {
        const int half = 3 * fc.num_octaves;
        const int g = (f < half) ? f : f - half;
        const int d = g / fc.num_octaves, o = g % fc.num_octaves;
        const float m = exp2f((float)(fc.start_octave + o)) * 3.14159265358979323846f;
        enc0 = make_float4(d == 0 ? m : 0.f, d == 1 ? m : 0.f, d == 2 ? m : 0.f, (f < half) ? NGM_FK_SIN : NGM_FK_COS);
      } else {
        enc0 = make_float4(f == 0 ? 1.f : 0.f, f == 1 ? 1.f : 0.f, f == 2 ? 1.f : 0.f, NGM_FK_RAW);
      }
    }
  }
  __device__ __forceinline__ void commit(float* sm, const ngm_field_cfg& fc) const {
    using LY = LYS;
    const int tid = threadIdx.x, nthr = blockDim.x;
    if (fc.encoding == NGM_ENC_PERMUTO) {
      if (tid < 16) {
        reinterpret_cast<float4*>(sm + LY::ENCW)[2 * tid] = enc0;
        reinterpret_cast<float4*>(sm + LY::ENCW)[2 * tid + 1] = enc1;
      }
    } else if (tid < MI * 32) {
      reinterpret_cast<float4*>(sm + LY::ENCW)[tid] = enc0;
    }
#pragma unroll
    for (int l = 0; l < L; ++l) {
      const int MIN = (l == 0) ? MI : LYS::MIN(1);
      const int ncol4 = MIN * 8, total4 = MH * 32 * ncol4;
      float* dst = sm + LY::w_off(l);
#pragma unroll
      for (int it = 0; it < NIT; ++it) {
        const int e4 = tid + it * nthr;
        if (e4 < total4) {
          const int o = e4 / ncol4, c = 4 * (e4 - o * ncol4);
          const int mo = o >> 5, io = o & 31, mi = c >> 5, ic = c & 31;
          const float x[4] = {v[l][it].x, v[l][it].y, v[l][it].z, v[l][it].w};
#pragma unroll
          for (int j = 0; j < 4; ++j)
            dst[((((mo * MIN + mi) * 16 + col_r(ic + j)) * 2 + col_hi(ic + j)) * NGM_WGS) + io] = x[j];
        }
      }
      if (tid < MH * 32) sm[LY::b_off(l) + tid] = bias[l];
    }
    if (tid < MH * 32) reinterpret_cast<float4*>(sm + LY::WOUT)[tid] = wout;
    if (CAT && tid < MI * 32) reinterpret_cast<float4*>(sm + LY::WOUT)[MH * 32 + tid] = wout_e;
    if (tid < 4) sm[LY::BOUT + tid] = bout;
  }
};

// Encoding of one sample position into the lane's B-operand registers.
// Lane (j = lane&31, hi = lane>>5) holds features 32*mi + frow(r,hi) of sample j.
// Feature kinds: RAW features can only sit in slots (mi = 0, r < 3, hi = 0) (features 0..2); every other
// slot is sin(w.x) -- or cos(w.x) when NEED_COS (NeRF octaves).  Padded features (>= D) have zero table
// rows and meet zero weight columns, so their value (sin 0 / cos 0) never reaches an output.
// WITH_DERIV additionally returns d(value)/d(arg) for the learnable Fourier matrix.
template <int MI, bool NEED_COS, bool WITH_DERIV>
__device__ __forceinline__ void encode_sample(const float* sm_encw, int hi, float x, float y, float z,
                                              f32x16 (&E)[MI], f32x16 (&dE)[MI]) {
  const float4* tab = reinterpret_cast<const float4*>(sm_encw);
#pragma unroll
  for (int mi = 0; mi < MI; ++mi) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float4 w = tab[32 * mi + frow(r, 0) + 4 * hi];
      const float arg = fmaf(w.z, z, fmaf(w.y, y, w.x * x));
      float s, c = 0.f;
      if constexpr (!NEED_COS && !WITH_DERIV) s = ngm_sinf(arg);     // forward of the Fourier encoding: sine only
      else ngm_sincosf(arg, &s, &c);
      float v = s, d = c;
      if (NEED_COS) { const bool is_cos = (w.w == NGM_FK_COS); v = is_cos ? c : s; d = is_cos ? -s : c; }
      if (mi == 0 && r < 3) { const bool raw = (w.w == NGM_FK_RAW); v = raw ? arg : v; d = raw ? 0.f : d; }
      E[mi][r] = v;
      if (WITH_DERIV) dE[mi][r] = d;
      if ((r & 3) == 3) __builtin_amdgcn_sched_barrier(0);
    }
  }
}

// Sine-only encoding (Fourier features, forward) of the lane's TWO samples (one per 32-column tile) at once:
// both use the same table rows, so every step is a packed-fp32 instruction over (tile 0, tile 1).
template <int MI>
__device__ __forceinline__ void encode_pair_sin(const float* sm_encw, int hi, ngm_v2f x, ngm_v2f y, ngm_v2f z,
                                                f32x16 (&E0)[MI], f32x16 (&E1)[MI]) {
  const float4* tab = reinterpret_cast<const float4*>(sm_encw);
#ifndef NGM_FWD_POLYSIN
  const ngm_v2f xr = x * ngm_splat2(0.15915494309189535f), yr = y * ngm_splat2(0.15915494309189535f),
                zr = z * ngm_splat2(0.15915494309189535f);
#endif
#pragma unroll
  for (int mi = 0; mi < MI; ++mi) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float4 w = tab[32 * mi + frow(r, 0) + 4 * hi];
#ifndef NGM_FWD_POLYSIN  // default: v_sin_f32 on the argument in revolutions (position pre-scaled by 1/(2 pi)); measured
      // parity error of the fused forward against the oracle is unchanged (7e-7 vs 1e-6 abs at sigma 4, 1.8e-6 vs 3.2e-6
      // at sigma 25: the fp32 evaluation order dominates, not the sine) and the kernel is 5 us faster.
      // -DNGM_FWD_POLYSIN restores the 1.2e-7 polynomial (ngm_sinf2).
      const ngm_v2f rev = ngm_fma2(ngm_splat2(w.z), zr, ngm_fma2(ngm_splat2(w.y), yr, ngm_splat2(w.x) * xr));
      ngm_v2f v = {__builtin_amdgcn_sinf(__builtin_amdgcn_fractf(rev.x)), __builtin_amdgcn_sinf(__builtin_amdgcn_fractf(rev.y))};
      const ngm_v2f arg = {r == 0 ? x.x : r == 1 ? y.x : z.x, r == 0 ? x.y : r == 1 ? y.y : z.y};   // raw rows only
#else
      const ngm_v2f arg = ngm_fma2(ngm_splat2(w.z), z, ngm_fma2(ngm_splat2(w.y), y, ngm_splat2(w.x) * x));
      ngm_v2f v = ngm_sinf2(arg);
#endif
      if (mi == 0 && r < 3) { const bool raw = (w.w == NGM_FK_RAW); v.x = raw ? arg.x : v.x; v.y = raw ? arg.y : v.y; }
      E0[mi][r] = v.x; E1[mi][r] = v.y;
      if ((r & 3) == 3) __builtin_amdgcn_sched_barrier(0);
    }
  }
}

// ------------------------------------------------------------------------------------------------
// Permutohedral-lattice hash encoding (positional_encodings.py:19-66).  PARITY UNPINNED: restates the
// published algorithm exactly as oracle/ngm_oracle.py:encode_permuto (elevate -> nearest remainder-0
// point -> rank -> barycentric -> hash the 4 simplex vertices).  FP contraction is off so that the
// lattice coordinates (up to ~1e5 at the finest level) round exactly like the oracle's torch ops.
// ------------------------------------------------------------------------------------------------
struct HashCtx {
  const void* tab;       // this field's table: [L][T] x 2 features, fp32 or 16-bit storage (dt)
  int dt;
  uint32_t mask;         // T - 1
  int T, nlev;
  float* gtab;           // gradient table (backward) or nullptr
};

// Simplex search of one level (Adams et al. 2010, as restated in oracle/ngm_oracle.py:encode_permuto, whose operation order
// the comments below refer to as "the restatement").  Round 3 rewrite, bit-identical to the literal transcription it
// replaces (tools/micro/simplex_check.hip: 1.07e9 random and tie-provoking points, 0 differences; 122 instead of 175 VALU
// instructions per call, 1182 -> 835 ns per call and wave at four waves per SIMD):
//  * nearest remainder-0 point: the restatement takes ceil / floor of el / 4 and compares the two distances (the lower one
//    wins a tie); that is round-half-down of v = el / 4 = rint(v), minus one where rint went UP at an exact tie
//    (rint(v) - v is exact, so the tie test is);
//  * the residual el - 4 k is only ever compared and scaled: t = k - v (exact) replaces it, diff_i < diff_j <=> t_i > t_j;
//  * the two wrap-around branches on rank + sum are adj = (rank + sum) >> 2 in {-1, 0, 1}, rank & 3, k - adj; and
//    delta = (el - 4 (k - adj)) / 4 with its one rounding is adj - t with its one rounding (k - v exact, adj a small integer);
//  * ranks from the six comparisons by inclusion-exclusion instead of twelve conditional increments.
__device__ __forceinline__ void permuto_simplex(float x, float y, float z, const float* lp, uint32_t mask,
                                                uint32_t (&idx)[4], float (&bw)[4]) {
#pragma clang fp contract(off)
  const float c0 = (x + lp[4]) * lp[0], c1 = (y + lp[5]) * lp[1], c2 = (z + lp[6]) * lp[2];
  float el[4];
  float sm = 0.f;
  { const float t3 = 3.0f * c2; el[3] = sm - t3; sm = sm + c2; }
  { const float t2 = 2.0f * c1; el[2] = sm - t2; sm = sm + c1; }
  { const float t1 = 1.0f * c0; el[1] = sm - t1; sm = sm + c0; }
  el[0] = sm;
  int k[4]; float t[4];
  int sum = 0;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const float v = el[i] * 0.25f;
    const float r = __builtin_rintf(v);
    const float tt = r - v;
    const bool tie = tt == 0.5f;
    t[i] = tie ? -0.5f : tt;
    k[i] = (int)r - (tie ? 1 : 0);
    sum += k[i];                                  // = (sum of the remainder-0 coordinates) / 4
  }
  const int l01 = t[0] > t[1], l02 = t[0] > t[2], l03 = t[0] > t[3], l12 = t[1] > t[2], l13 = t[1] > t[3], l23 = t[2] > t[3];
  int rank[4];
  rank[0] = sum + l01 + l02 + l03;
  rank[1] = sum + 1 - l01 + l12 + l13;
  rank[2] = sum + 2 - l02 - l12 + l23;
  rank[3] = sum + 3 - l03 - l13 - l23;
  float delta[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int adj = rank[i] >> 2;
    rank[i] &= 3;
    k[i] -= adj;
    delta[i] = (float)adj - t[i];
  }
  // The ranks are a permutation of 0..3.  With d[k] = delta of the coordinate ranked k, the restatement's scatter
  // bary[3 - rank_i] += delta_i, bary[4 - rank_i] -= delta_i is bary[s] = d[3 - s] - d[4 - s] (same roundings), and
  // the hash ((k0 P + k1) P + k2) P of vertex r (key_i = rem0_i + r - 4 [rank_i > 3 - r]) is linear mod 2^32:
  // h(r) = h(r - 1) + (P + P^2 + P^3) - 4 P^(3 - i) for the hashed coordinate i ranked 4 - r.  3 integer
  // multiplies instead of 12 and 24 selects instead of 80.
  constexpr uint32_t P1 = 2531011u, P2 = 2220443785u, P3 = 2937900635u;   // P, P^2, P^3 mod 2^32
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
  uint32_t h = ((uint32_t)k[0] * P3 + (uint32_t)k[1] * P2 + (uint32_t)k[2] * P1) * 4u;   // remainder-0 coordinates are 4 k
  idx[0] = h & mask;
  h += C - q[3]; idx[1] = h & mask;
  h += C - q[2]; idx[2] = h & mask;
  h += C - q[1]; idx[3] = h & mask;
}

// features of the lane's 8 levels straight into the MFMA B-operand registers of the single 32-feature
// tile: register r = 4q + 2p + c holds feature c of level 4q + 2*hi + p.
__device__ __forceinline__ void encode_hash(const float* sm_lvl, const HashCtx& hc, int hi, float x, float y, float z,
                                            f32x16& E) {
  if (hc.dt == NGM_DT_F32) {
    // fp32 tables (the default): straight-line code over the lane's 8 levels -- no branch between one level's gathers and
    // the next level's simplex search, so the scheduler keeps the next gathers in flight while it consumes these.  Levels
    // beyond nr_levels (lane-dependent: level = 4 q + 2 hi + p) read level 0's table and are zeroed by a select.
    const float2* tab = reinterpret_cast<const float2*>(hc.tab);
    // software-pipelined by hand: the gathers of level k + 1 are issued before the features of level k are formed
    uint32_t idx[4]; float bw[4]; float2 v[4]; bool on;
    auto fetch = [&](int k) __attribute__((always_inline)) {
      const int level = 4 * (k >> 1) + 2 * hi + (k & 1);
      on = level < hc.nlev;
      const int lv = on ? level : 0;
      permuto_simplex(x, y, z, sm_lvl + 8 * lv, hc.mask, idx, bw);
      const float2* t = tab + (size_t)lv * hc.T;
#pragma unroll
      for (int r = 0; r < 4; ++r) v[r] = t[idx[r]];
    };
    fetch(0);
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      float2 vc[4]; float bc[4];
      const bool onc = on;
#pragma unroll
      for (int r = 0; r < 4; ++r) { vc[r] = v[r]; bc[r] = bw[r]; }
      if (k < 7) fetch(k + 1);
      float f0 = 0.f, f1 = 0.f;
#pragma unroll
      for (int r = 0; r < 4; ++r) { f0 = fmaf(vc[r].x, bc[r], f0); f1 = fmaf(vc[r].y, bc[r], f1); }
      E[2 * k] = onc ? f0 : 0.f;
      E[2 * k + 1] = onc ? f1 : 0.f;
    }
    return;
  }
#pragma unroll
  for (int q = 0; q < 4; ++q)
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      const int level = 4 * q + 2 * hi + p;
      float f0 = 0.f, f1 = 0.f;
      if (level < hc.nlev) {
        uint32_t idx[4]; float bw[4];
        permuto_simplex(x, y, z, sm_lvl + 8 * level, hc.mask, idx, bw);
        float2 v[4];
        ngm_ldp2x4(hc.tab, (size_t)level * hc.T, idx, hc.dt, v);
#pragma unroll
        for (int r = 0; r < 4; ++r) { f0 = fmaf(v[r].x, bw[r], f0); f1 = fmaf(v[r].y, bw[r], f1); }
      }
      E[4 * q + 2 * p] = f0;
      E[4 * q + 2 * p + 1] = f1;
    }
}

// (Staging one level's table at a time in LDS for these gathers was measured and dropped: 196 us against 111 us for the fused
// forward, profiles/r03a_hash_lds_experiment.txt and DESIGN.md 3.14.)

// scatter dL/dE of the lane's 8 levels into the gradient table (float atomics in L2)
__device__ __forceinline__ void scatter_hash_grad(const float* sm_lvl, const HashCtx& hc, int hi, float x, float y, float z,
                                                  const f32x16& dE, bool valid) {
#pragma unroll
  for (int q = 0; q < 4; ++q)
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      const int level = 4 * q + 2 * hi + p;
      if (level < hc.nlev && valid) {
        uint32_t idx[4]; float bw[4];
        permuto_simplex(x, y, z, sm_lvl + 8 * level, hc.mask, idx, bw);
        float* g = hc.gtab + ((size_t)level * hc.T) * 2;
        const float d0 = dE[4 * q + 2 * p], d1 = dE[4 * q + 2 * p + 1];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          unsafeAtomicAdd(g + 2 * idx[r], d0 * bw[r]);
          unsafeAtomicAdd(g + 2 * idx[r] + 1, d1 * bw[r]);
        }
      }
    }
}

__device__ __forceinline__ HashCtx make_hash_ctx(const ngm_field_cfg& fc, const ngm_params& pr, int64_t row, float* gtab) {
  HashCtx hc;
  hc.T = 1 << fc.log2_hashmap_size;
  hc.mask = (uint32_t)hc.T - 1u;
  hc.nlev = fc.nr_levels;
  hc.dt = pr.dtype;
  hc.tab = nullptr;
  if (fc.encoding == NGM_ENC_PERMUTO)
    hc.tab = reinterpret_cast<const char*>(pr.lattice) + (size_t)(row * pr.lattice_stride) * (pr.dtype == NGM_DT_F32 ? 4 : 2);
  hc.gtab = gtab;
  return hc;
}

// ------------------------------------------------------------------------------------------------
// Triplane encoding (positional_encodings.py:69-161): planes (3, C, res, res) of one field; plane p samples the
// projection (u, v) = (x,y), (x,z), (y,z) of the point with torch.nn.functional.grid_sample semantics
// (align_corners=True: pixel = (c + 1) / 2 * (res - 1); padding_mode="border": clamped; bilinear; u indexes the last
// (width) axis, v the height axis).  Feature c: sum / product of the three plane values, or (concat) feature p C + c.
// ------------------------------------------------------------------------------------------------
struct TriCtx {
  const float* planes;   // this field's (3, C, res, res)
  int res, C, mode;
  long long* acc;        // backward: Q23.40 fixed-point gradient accumulator, same layout, or nullptr
};
struct TriTap { int i00, i01, i10, i11; float w00, w01, w10, w11; };   // offsets inside one (res, res) plane + weights

__device__ __forceinline__ TriTap tri_tap(float u, float v, int res) {
  const float fu = fminf(fmaxf((u + 1.0f) * 0.5f * (float)(res - 1), 0.f), (float)(res - 1));
  const float fv = fminf(fmaxf((v + 1.0f) * 0.5f * (float)(res - 1), 0.f), (float)(res - 1));
  const float u0f = floorf(fu), v0f = floorf(fv);
  const int u0 = (int)u0f, v0 = (int)v0f, u1 = min(u0 + 1, res - 1), v1 = min(v0 + 1, res - 1);
  const float tu = fu - u0f, tv = fv - v0f;
  TriTap t;
  t.i00 = v0 * res + u0; t.i01 = v0 * res + u1; t.i10 = v1 * res + u0; t.i11 = v1 * res + u1;
  t.w00 = (1.f - tu) * (1.f - tv); t.w01 = tu * (1.f - tv); t.w10 = (1.f - tu) * tv; t.w11 = tu * tv;
  return t;
}
__device__ __forceinline__ float tri_fetch(const float* plane, const TriTap& t) {
  return fmaf(plane[t.i11], t.w11, fmaf(plane[t.i10], t.w10, fmaf(plane[t.i01], t.w01, plane[t.i00] * t.w00)));
}
// the lane's features 32 mi + frow(r, hi) of one sample, straight into the MFMA B-operand registers
template <int MI>
__device__ __forceinline__ void encode_triplane(const TriCtx& tc, int hi, float x, float y, float z, f32x16 (&E)[MI]) {
  const TriTap tp[3] = {tri_tap(x, y, tc.res), tri_tap(x, z, tc.res), tri_tap(y, z, tc.res)};
  const int64_t ps = (int64_t)tc.res * tc.res, D = tc.mode == NGM_TRI_CONCAT ? 3 * tc.C : tc.C;
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int f = 32 * mi + frow(r, hi);
      float v = 0.f;
      if (f < D) {
        if (tc.mode == NGM_TRI_CONCAT) {
          const int p = f / tc.C, c = f - p * tc.C;
          v = tri_fetch(tc.planes + ((int64_t)p * tc.C + c) * ps, tp[p]);
        } else {
          const float a0 = tri_fetch(tc.planes + (int64_t)f * ps, tp[0]);
          const float a1 = tri_fetch(tc.planes + ((int64_t)tc.C + f) * ps, tp[1]);
          const float a2 = tri_fetch(tc.planes + ((int64_t)2 * tc.C + f) * ps, tp[2]);
          v = tc.mode == NGM_TRI_SUM ? (a0 + a1) + a2 : (a0 * a1) * a2;
        }
      }
      E[mi][r] = v;
    }
}
__device__ __forceinline__ void tri_scatter(long long* plane_acc, const TriTap& t, float g) {
  // 64-bit fixed point (Q23.40): integer atomics are order-independent -> the plane gradient is deterministic
  auto fix = [](float v) { return (long long)((double)v * 1099511627776.0); };
  atomicAdd(reinterpret_cast<unsigned long long*>(plane_acc + t.i00), (unsigned long long)fix(g * t.w00));
  atomicAdd(reinterpret_cast<unsigned long long*>(plane_acc + t.i01), (unsigned long long)fix(g * t.w01));
  atomicAdd(reinterpret_cast<unsigned long long*>(plane_acc + t.i10), (unsigned long long)fix(g * t.w10));
  atomicAdd(reinterpret_cast<unsigned long long*>(plane_acc + t.i11), (unsigned long long)fix(g * t.w11));
}
// d loss / d planes from the lane's d loss / d E (same register map as encode_triplane)
template <int MI>
__device__ __forceinline__ void scatter_triplane_grad(const TriCtx& tc, int hi, float x, float y, float z, const f32x16 (&dE)[MI],
                                                      bool valid) {
  if (!valid) return;
  const TriTap tp[3] = {tri_tap(x, y, tc.res), tri_tap(x, z, tc.res), tri_tap(y, z, tc.res)};
  const int64_t ps = (int64_t)tc.res * tc.res, D = tc.mode == NGM_TRI_CONCAT ? 3 * tc.C : tc.C;
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int f = 32 * mi + frow(r, hi);
      const float g = dE[mi][r];
      if (f >= D || g == 0.f) continue;
      if (tc.mode == NGM_TRI_CONCAT) {
        const int p = f / tc.C, c = f - p * tc.C;
        tri_scatter(tc.acc + ((int64_t)p * tc.C + c) * ps, tp[p], g);
      } else if (tc.mode == NGM_TRI_SUM) {
#pragma unroll
        for (int p = 0; p < 3; ++p) tri_scatter(tc.acc + ((int64_t)p * tc.C + f) * ps, tp[p], g);
      } else {
        float a[3];
#pragma unroll
        for (int p = 0; p < 3; ++p) a[p] = tri_fetch(tc.planes + ((int64_t)p * tc.C + f) * ps, tp[p]);
        tri_scatter(tc.acc + (int64_t)f * ps, tp[0], g * (a[1] * a[2]));
        tri_scatter(tc.acc + ((int64_t)tc.C + f) * ps, tp[1], g * (a[0] * a[2]));
        tri_scatter(tc.acc + ((int64_t)2 * tc.C + f) * ps, tp[2], g * (a[0] * a[1]));
      }
    }
}
__device__ __forceinline__ TriCtx make_tri_ctx(const ngm_field_cfg& fc, const ngm_params& pr, int64_t row, long long* acc) {
  TriCtx tc;
  tc.res = fc.tri_resolution; tc.mode = fc.tri_mode;
  tc.C = fc.tri_mode == NGM_TRI_CONCAT ? fc.dim_enc / 3 : fc.dim_enc;
  tc.planes = (fc.encoding == NGM_ENC_TRIPLANE) ? pr.planes + row * pr.planes_stride : nullptr;
  tc.acc = acc;
  return tc;
}

// One hidden layer on the matrix cores: Y = relu(W X + b), X/Y in C-layout registers, NT sample tiles.
template <int MIN, int MOUT, int NT>
__device__ __forceinline__ void layer_fwd(const float* __restrict__ W, const float* __restrict__ B, int lane,
                                          const f32x16 (&X)[NT][MIN], f32x16 (&Y)[NT][MOUT]) {
  const int io = lane & 31, hi = lane >> 5;
  // accumulators start at 0 (an inline constant of the first MFMA, no register initialisation); the bias is added
  // after the contraction, packed, like the reference's addmm epilogue
#pragma unroll
  for (int mo = 0; mo < MOUT; ++mo)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int r = 0; r < 16; ++r) Y[nt][mo][r] = 0.f;
  // software-pipelined A-fragment fetch: the LDS reads of k-step group g+1 are issued before the MFMAs
  // of group g (4 k-steps per group), so their latency hides under the matrix pipe.
  constexpr int NG = MIN * 4;
  float abuf[2][4][MOUT];
  const float* Wl = W + hi * NGM_WGS + io;
#pragma unroll
  for (int rr = 0; rr < 4; ++rr)
#pragma unroll
    for (int mo = 0; mo < MOUT; ++mo) abuf[0][rr][mo] = Wl[((mo * MIN + 0) * 16 + rr) * 2 * NGM_WGS];
#pragma unroll
  for (int g = 0; g < NG; ++g) {
    if (g + 1 < NG) {
      const int mi1 = (4 * (g + 1)) / 16, r1 = (4 * (g + 1)) % 16;
#pragma unroll
      for (int rr = 0; rr < 4; ++rr)
#pragma unroll
        for (int mo = 0; mo < MOUT; ++mo)
          abuf[(g + 1) & 1][rr][mo] = Wl[((mo * MIN + mi1) * 16 + r1 + rr) * 2 * NGM_WGS];
    }
    __builtin_amdgcn_sched_barrier(0);   // keep the prefetch above this group's MFMAs
    const int mi = (4 * g) / 16, r0 = (4 * g) % 16;
#pragma unroll
    for (int rr = 0; rr < 4; ++rr)
#pragma unroll
      for (int mo = 0; mo < MOUT; ++mo)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) Y[nt][mo] = mfma32(abuf[g & 1][rr][mo], X[nt][mi][r0 + rr], Y[nt][mo]);
    __builtin_amdgcn_sched_barrier(0);
  }
#pragma unroll
  for (int mo = 0; mo < MOUT; ++mo) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const float4 b4 = *reinterpret_cast<const float4*>(B + 32 * mo + 8 * q + 4 * hi);
      const ngm_v2f b01 = {b4.x, b4.y}, b23 = {b4.z, b4.w};
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        const ngm_v2f y01 = ngm_v2f{Y[nt][mo][4 * q], Y[nt][mo][4 * q + 1]} + b01;
        const ngm_v2f y23 = ngm_v2f{Y[nt][mo][4 * q + 2], Y[nt][mo][4 * q + 3]} + b23;
        Y[nt][mo][4 * q] = ngm_relu(y01.x); Y[nt][mo][4 * q + 1] = ngm_relu(y01.y);
        Y[nt][mo][4 * q + 2] = ngm_relu(y23.x); Y[nt][mo][4 * q + 3] = ngm_relu(y23.y);
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------
// fp32 via a three-way bf16 split (opt-in, ngm_field_cfg.matmul_mode = NGM_MATMUL_BF16X3).
// Every fp32 operand is written EXACTLY as hi + mid + lo, three bf16 (8 + 8 + 8 mantissa bits, split by
// truncation so that each residual is exact); the product a b is then the sum of nine bf16 products, of which
// the six with relative weight >= 2^-16 are issued on v_mfma_f32_32x32x16_bf16 (fp32 accumulate): hi hi, hi mid,
// mid hi, mid mid, hi lo, lo hi.  The three dropped terms are below 2^-23 |a b| -- the rounding of one fp32
// operation.  Six bf16 MFMAs of 16 k-steps replace eight fp32 MFMAs of 2 k-steps at 16x the flop rate, and the
// bf16 matrix pipe does not share the fp32 FMA lanes with the VALU (tools/micro/coexec.hip), so the split
// arithmetic overlaps the partner wave's MFMAs.
// K order: k-block kb = 2 mi + b, lane half kh, element e  <->  feature 32 mi + frow(8 b + e, kh), i.e. register
// 8 b + e of input tile mi: the B operand of a k-block is eight consecutive C-layout registers of the previous
// layer's output (packed in place, no data movement between lanes), the weights are stored in the same order.
// ------------------------------------------------------------------------------------------------
typedef __bf16 ngm_bf16x8 __attribute__((ext_vector_type(8)));
typedef uint32_t ngm_u32x4 __attribute__((ext_vector_type(4)));

// x == hi + mid + lo with hi, mid, lo bf16 (bit patterns in the upper halves of h, m, l)
__device__ __forceinline__ void b3_split(float x, uint32_t& h, uint32_t& m, uint32_t& l) {
  h = __float_as_uint(x) & 0xffff0000u;
  const float r1 = x - __uint_as_float(h);            // exact
  m = __float_as_uint(r1) & 0xffff0000u;
  const float r2 = r1 - __uint_as_float(m);           // exact, <= 8 significant bits
  l = __float_as_uint(r2);
}
// two bf16 (upper halves of even / odd) -> one packed word, even element in the low half
__device__ __forceinline__ uint32_t b3_pack(uint32_t even, uint32_t odd) { return __builtin_amdgcn_perm(odd, even, 0x07060302u); }

// One pair of values -> the three packed words (even element in the low half).  Three forms of the same exact
// arithmetic (bit-identical planes for normal numbers, tools/micro/dot2c_split.hip):
//   default       h = x & 0xffff0000, r = x - h, ... : 4 VALU per element + 3 v_perm per pair
//   NGM_SPLIT_DOT2C  the packed hi pair is formed FIRST (one v_perm straight from the fp32 registers), and each residual is
//                 one v_dot2c_f32_bf16 (gfx950): r0 = x0 + hp.lo * (-1) + hp.hi * 0 -- exact, since r is representable --
//                 2 VALU per element + 3 v_perm per pair
//   NGM_SPLIT_PK  the two residual subtractions of a pair as one v_pk_add_f32
__device__ __forceinline__ void b3_split2(float x0, float x1, uint32_t& hp, uint32_t& mp, uint32_t& lp) {
#if defined(NGM_SPLIT_DOT2C)
  typedef __bf16 bf2_ __attribute__((ext_vector_type(2)));
  uint32_t clo = 0x0000bf80u, chi = 0xbf800000u;        // (-1, 0) and (0, -1) as (low, high) bf16
#if NGM_SPLIT_DOT2C == 1
  asm("" : "+v"(clo)); asm("" : "+v"(chi));             // keep them in registers (no inline-constant folding)
#endif
  hp = b3_pack(__float_as_uint(x0), __float_as_uint(x1));
  const float r0 = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf2_, hp), __builtin_bit_cast(bf2_, clo), x0, false);
  const float r1 = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf2_, hp), __builtin_bit_cast(bf2_, chi), x1, false);
  mp = b3_pack(__float_as_uint(r0), __float_as_uint(r1));
  const float q0 = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf2_, mp), __builtin_bit_cast(bf2_, clo), r0, false);
  const float q1 = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf2_, mp), __builtin_bit_cast(bf2_, chi), r1, false);
  lp = b3_pack(__float_as_uint(q0), __float_as_uint(q1));
#elif defined(NGM_SPLIT_PK)
  const uint32_t h0 = __float_as_uint(x0) & 0xffff0000u, h1 = __float_as_uint(x1) & 0xffff0000u;
  const ngm_v2f r = ngm_v2f{x0, x1} - ngm_v2f{__uint_as_float(h0), __uint_as_float(h1)};
  const uint32_t m0 = __float_as_uint(r.x) & 0xffff0000u, m1 = __float_as_uint(r.y) & 0xffff0000u;
  const ngm_v2f q = r - ngm_v2f{__uint_as_float(m0), __uint_as_float(m1)};
  hp = b3_pack(h0, h1); mp = b3_pack(m0, m1); lp = b3_pack(__float_as_uint(q.x), __float_as_uint(q.y));
#else
  uint32_t h0, m0, l0, h1, m1, l1;
  b3_split(x0, h0, m0, l0);
  b3_split(x1, h1, m1, l1);
  hp = b3_pack(h0, h1); mp = b3_pack(m0, m1); lp = b3_pack(l0, l1);
#endif
}

__device__ __forceinline__ void b3_split8(const float (&x)[8], ngm_bf16x8& H, ngm_bf16x8& M, ngm_bf16x8& Lo) {
#ifdef NGM_ABLB_NOSPLIT   // timing ablation of k_field_bwd_b3 (results meaningless): operands straight from the fp32 bits, no split arithmetic
  H = __builtin_bit_cast(ngm_bf16x8, ngm_u32x4{__float_as_uint(x[0]), __float_as_uint(x[1]), __float_as_uint(x[2]), __float_as_uint(x[3])});
  M = __builtin_bit_cast(ngm_bf16x8, ngm_u32x4{__float_as_uint(x[4]), __float_as_uint(x[5]), __float_as_uint(x[6]), __float_as_uint(x[7])});
  Lo = H;
  return;
#endif
  ngm_u32x4 h4, m4, l4;
#pragma unroll
  for (int p = 0; p < 4; ++p) {
#if defined(NGM_SPLIT_DOT2C) || defined(NGM_SPLIT_PK)
    uint32_t hp, mp, lp;
    b3_split2(x[2 * p], x[2 * p + 1], hp, mp, lp);
    h4[p] = hp; m4[p] = mp; l4[p] = lp;
#else
    uint32_t h0, m0, l0, h1, m1, l1;
    b3_split(x[2 * p], h0, m0, l0);
    b3_split(x[2 * p + 1], h1, m1, l1);
    h4[p] = b3_pack(h0, h1); m4[p] = b3_pack(m0, m1); l4[p] = b3_pack(l0, l1);
#endif
  }
  H = __builtin_bit_cast(ngm_bf16x8, h4); M = __builtin_bit_cast(ngm_bf16x8, m4); Lo = __builtin_bit_cast(ngm_bf16x8, l4);
}

// LDS plane store of one hidden layer (16-byte units): [plane 3][mo][kb = 2 MIN][kh 2][io 32]
template <int MIN, int MOUT>
struct B3Planes {
  static constexpr int KB = 2 * MIN;
  static constexpr int PLANE = MOUT * KB * 2 * 32;        // ngm_u32x4 units per plane
  static constexpr int LAYER = 3 * PLANE;
  static constexpr int idx(int p, int mo, int kb, int kh, int io) { return p * PLANE + ((mo * KB + kb) * 2 + kh) * 32 + io; }
};
template <int MI, int MH, int L>
struct B3Lds {                                             // all hidden layers of a field (skip_mode "no")
  static constexpr int off(int l) { return l == 0 ? 0 : B3Planes<MI, MH>::LAYER + (l - 1) * B3Planes<MH, MH>::LAYER; }
  static constexpr int TOTAL = off(L);                     // 16-byte units
};

// build the planes from the fp32 A-fragments FieldStage::commit() wrote (call after its barrier; barrier after this)
template <int MI, int MH, int L>
__device__ __forceinline__ void b3_build_planes(const float* sm, ngm_u32x4* planes) {
  using LY = FieldLds<MI, MH, L, false>;
#pragma unroll
  for (int l = 0; l < L; ++l) {
    const int MIN = (l == 0) ? MI : MH, KB = 2 * MIN;
    const float* W = sm + LY::w_off(l);
    ngm_u32x4* P = planes + B3Lds<MI, MH, L>::off(l);
    const int plane = MH * KB * 2 * 32;
    for (int e4 = threadIdx.x; e4 < plane; e4 += blockDim.x) {
      const int io = e4 & 31, kh = (e4 >> 5) & 1, kb = (e4 >> 6) % KB, mo = (e4 >> 6) / KB;
      const int mi = kb >> 1, b = kb & 1;
      float x[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) x[e] = W[((((mo * MIN + mi) * 16 + 8 * b + e) * 2 + kh) * NGM_WGS) + io];
      ngm_bf16x8 H, M, Lo;
      b3_split8(x, H, M, Lo);
      P[e4] = __builtin_bit_cast(ngm_u32x4, H);
      P[plane + e4] = __builtin_bit_cast(ngm_u32x4, M);
      P[2 * plane + e4] = __builtin_bit_cast(ngm_u32x4, Lo);
    }
  }
}

__device__ __forceinline__ f32x16 mfma_bf16(ngm_bf16x8 a, ngm_bf16x8 b, f32x16 c) {
#ifdef NGM_ABLB_NOMFMA    // timing ablation of k_field_bwd_b3 (results meaningless): one vector instruction that keeps both operands alive
  const ngm_u32x4 ua = __builtin_bit_cast(ngm_u32x4, a), ub = __builtin_bit_cast(ngm_u32x4, b);
  c[0] += __uint_as_float((ua[0] ^ ub[1]) & 0x3fffffffu);
  return c;
#endif
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}

// (the stash record of the training forward; layout comments: act_store below)
struct ActStash {
  float* base;            // NULL: nothing is stored
  int64_t layer_stride;   // floats between layers
  int64_t g0;             // global sample index of lane 0 of this 64-sample step
  int nvalid;             // samples of this step that exist (lanes >= nvalid store nothing)
  int nlayers;            // hidden layers whose output is stashed: layers 0 .. nlayers - 1 (half stash: 1 of 2, the split
                          // backward recomputes the other; RenderFwdArgs::act_layers)
};

// Y = relu(W X + b) like layer_fwd, through the six-product bf16 split
template <int MIN, int MOUT, int NT>
__device__ __forceinline__ void layer_fwd_b3(const ngm_u32x4* __restrict__ P, const float* __restrict__ B, int lane,
                                             const f32x16 (&X)[NT][MIN], f32x16 (&Y)[NT][MOUT]) {
  using PL = B3Planes<MIN, MOUT>;
  const int io = lane & 31, hi = lane >> 5;
  // The accumulators START from the bias (loaded from LDS), not from the inline constant 0: with a constant SrcC the
  // compiler is free to let the destination tile overlap the A / B operand registers of the same v_mfma (observed:
  // `v_mfma_f32_32x32x16_bf16 v[2:17], v[6:9], v[218:221], 0`), and launches of such code were not bitwise
  // reproducible; a tied SrcC = VDST chain cannot be allocated that way.  (Rounding order differs from the fp32 path's
  // bias-last epilogue by one fp32 addition; inside every tolerance.)
#pragma unroll
  for (int mo = 0; mo < MOUT; ++mo)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const float4 b4 = *reinterpret_cast<const float4*>(B + 32 * mo + 8 * q + 4 * hi);
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        Y[nt][mo][4 * q] = b4.x; Y[nt][mo][4 * q + 1] = b4.y; Y[nt][mo][4 * q + 2] = b4.z; Y[nt][mo][4 * q + 3] = b4.w;
      }
    }
#pragma unroll
  for (int kb = 0; kb < PL::KB; ++kb) {
    const int mi = kb >> 1, b = kb & 1;
    // weights of this k-block (three 16-byte reads per output tile), issued before the split so that they travel under it
    ngm_bf16x8 ah[MOUT], am[MOUT], al[MOUT];
#pragma unroll
    for (int mo = 0; mo < MOUT; ++mo) {
      ah[mo] = __builtin_bit_cast(ngm_bf16x8, P[PL::idx(0, mo, kb, hi, io)]);
      am[mo] = __builtin_bit_cast(ngm_bf16x8, P[PL::idx(1, mo, kb, hi, io)]);
      al[mo] = __builtin_bit_cast(ngm_bf16x8, P[PL::idx(2, mo, kb, hi, io)]);
    }
    ngm_bf16x8 bh[NT], bm[NT], bl[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      float x[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) x[e] = X[nt][mi][8 * b + e];
      b3_split8(x, bh[nt], bm[nt], bl[nt]);
    }
    // product-major: consecutive MFMAs go to DIFFERENT accumulators (MOUT * NT independent chains), so no MFMA
    // waits for (or depends on the forwarding of) the one issued right before it; small terms first
#define NGM_B3_PRODUCT(AW, BX)                                                                        \
    _Pragma("unroll") for (int mo = 0; mo < MOUT; ++mo)                                               \
      _Pragma("unroll") for (int nt = 0; nt < NT; ++nt) Y[nt][mo] = mfma_bf16(AW[mo], BX[nt], Y[nt][mo]); \
    __builtin_amdgcn_sched_barrier(0)
    __builtin_amdgcn_sched_barrier(0);
    NGM_B3_PRODUCT(al, bh);
    NGM_B3_PRODUCT(ah, bl);
    NGM_B3_PRODUCT(am, bm);
    NGM_B3_PRODUCT(am, bh);
    NGM_B3_PRODUCT(ah, bm);
    NGM_B3_PRODUCT(ah, bh);
#undef NGM_B3_PRODUCT
  }
  // The ReLU below is inline asm (one v_max_f32, ngm_relu) applied DIRECTLY to MFMA results: the compiler's hazard
  // recognizer does not look into asm statements, so the MFMA -> VALU read wait states (passes + 2 after the last
  // v_mfma; there is no hardware interlock) are supplied here by hand.  Without them the v_max read accumulators the
  // matrix pipe had not written yet whenever the partner wave kept the pipe busy: results differed from launch to
  // launch.  (The fp32 path adds the bias with compiler-visible instructions first, which get their wait states.)
#pragma unroll
  for (int mo = 0; mo < MOUT; ++mo)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      // volatile asm statements keep their order: the wait is issued once, after every MFMA; the empty ones only pin
      // the other accumulators' first VALU use behind it
      if (mo == 0 && nt == 0) asm volatile("s_nop 15\n\ts_nop 7" : "+v"(Y[nt][mo]));
      else asm volatile("" : "+v"(Y[nt][mo]));
    }
#pragma unroll
  for (int mo = 0; mo < MOUT; ++mo)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int r = 0; r < 16; ++r) Y[nt][mo][r] = ngm_relu(Y[nt][mo][r]);
}

// Output layer (4 x H) on the VALU: each lane reduces over ITS 16*MH features; the two lane halves
// of a sample hold disjoint feature sets and are combined by the caller.
template <int MH, int NT>
__device__ __forceinline__ void out_layer_partial(const float* sm_wout, int hi, const f32x16 (&Hh)[NT][MH],
                                                  float (&part)[NT][4]) {
  const float4* w4 = reinterpret_cast<const float4*>(sm_wout);
  // packed over the output channels: (w.x, w.y) and (w.z, w.w) are register pairs straight from the LDS read
  ngm_v2f p01[NT], p23[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) { p01[nt] = ngm_splat2(0.f); p23[nt] = ngm_splat2(0.f); }
#pragma unroll
  for (int mi = 0; mi < MH; ++mi) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float4 w = w4[32 * mi + frow(r, 0) + 4 * hi];
      const ngm_v2f w01 = {w.x, w.y}, w23 = {w.z, w.w};
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        const ngm_v2f h = ngm_splat2(Hh[nt][mi][r]);
        p01[nt] = ngm_fma2(w01, h, p01[nt]);
        p23[nt] = ngm_fma2(w23, h, p23[nt]);
      }
    }
  }
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) { part[nt][0] = p01[nt].x; part[nt][1] = p01[nt].y; part[nt][2] = p23[nt].x; part[nt][3] = p23[nt].y; }
}

// Hidden-activation stash written by the training forward and consumed by the backward kernel (which
// then skips the forward recompute: one third of its MFMA work).  fp32, 32*MH floats per sample and layer,
// tiled so that the forward's stores are fully coalesced:
//     act[layer][tile = g >> 5][chunk c = feature >> 2][position][4 floats],   g = global flat sample index
// i.e. 16-byte granules (4 consecutive features of one sample); the 32 lanes of a C-layout half-wave hold
// one chunk of 32 consecutive samples = 512 contiguous bytes per store instruction and half.
// Sample r = g & 31 of chunk c sits at position r ^ (c & 7) -- the image k_field_bwd_b3 / k_hash_mlp_bwd want in LDS
// (bank-conflict-free column reads), so that their HBM -> LDS transfers are plain linear copies: a transfer whose lanes
// gather a permutation costs the issuing wave ~240 clocks, a linear one ~30 (the permutation moves inside 128-byte
// groups, so the stores here stay whole lines).
#define NGM_ACT_CHUNKS(MH) (8 * (MH))     // 16-byte chunks per sample

// C layout: lane (n = lane & 31, hi) holds, for sample column n of tile nt, features
// 32m + 8g + 4hi + {0..3} in registers 4g..4g+3 -> chunk 8m + 2g + hi, one 16-byte store per (nt, m, g).
// WT: write-through stores (the fused training forward, whose consumer is the next kernel of a short iteration) or non-temporal
// ones (the point evaluation's training forward: gigabytes of stash per call -- written through, that kernel was 7 % slower and
// its backward no faster; same-box A/B, round 6)
template <int MH, int NT, bool WT = true>
__device__ __forceinline__ void act_store(const ActStash& st, int layer, int lane, const f32x16 (&H)[NT][MH]) {
  const int n = lane & 31, hi = lane >> 5;
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    const int s = 32 * nt + n;
    if (s < st.nvalid) {
      const int64_t g = st.g0 + s;
      float* p = st.base + layer * st.layer_stride + (((g >> 5) * NGM_ACT_CHUNKS(MH) + hi) * 32) * 4;
      const int r = (int)(g & 31);
#pragma unroll
      for (int m = 0; m < MH; ++m)
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4)
        {
          typedef float v4f __attribute__((ext_vector_type(4)));
          const v4f val = {H[nt][m][4 * g4], H[nt][m][4 * g4 + 1], H[nt][m][4 * g4 + 2], H[nt][m][4 * g4 + 3]};
          const int pos = r ^ ((2 * g4 + hi) & 7);                       // chunk = 8 m + 2 g4 + hi
#ifdef NGM_STASH_NT    // rounds 1-5: non-temporal stores (the lines stay in the writing XCD's L2 until evicted, dirty)
          constexpr bool wt_ = false;
#else
          constexpr bool wt_ = WT;
#endif
          if constexpr (!wt_) {
            __builtin_nontemporal_store(val, reinterpret_cast<v4f*>(p + (8 * m + 2 * g4) * 128 + 4 * pos));
          } else {
          // Round 6: WRITE-THROUGH stores (sc1).  The stash is read once, by the NEXT kernel, from whichever XCD its workgroup
          // lands on: written through, nothing of it sits dirty in the writer's L2 at the kernel boundary and the backward's
          // transfers are served by the memory side directly.  Same-box A/B at the M1 batch: forward 82.3 -> 80.8 us,
          // BACKWARD 142.5 -> 138.9 us, step 0.2324 -> 0.2266 ms (MI355X_MICROARCH.md, "publish-large": write-through wins for
          // tens of KB per workgroup handed to another kernel).
          asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(reinterpret_cast<v4f*>(p + (8 * m + 2 * g4) * 128 + 4 * pos)), "v"(val));
          }
        }
    }
  }
}

// Full forward MLP for NT tiles: E (encoding, C layout) -> partial outputs.  Keeps the last hidden
// activations in Hlast (needed by the backward kernel for the ReLU mask / output-layer gradient).
// skip connection "add" (models.py:162-169): the encoding is added to the first D units of every hidden layer's
// output; padded encoding features are 0, so whole tiles can be added
template <int MI, int MH, int NT>
__device__ __forceinline__ void skip_add(f32x16 (&H)[NT][MH], const f32x16 (&E)[NT][MI]) {
  if constexpr (MI <= MH) {
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int m = 0; m < MI; ++m)
#pragma unroll
        for (int r = 0; r < 16; ++r) H[nt][m][r] += E[nt][m][r];
  }
}

// SKIP (0 no, 1 add, 2 concat) is a template parameter on purpose: as a run-time flag it kept the encoding registers
// alive through every layer of the default path as well and cost the fused forward 14 us.
template <int MI, int MH, int L, int NT, int SKIP = 0, bool B3 = false, bool WT = true>
__device__ __forceinline__ void mlp_fwd(const float* sm, int lane, const f32x16 (&E)[NT][MI], f32x16 (&Hlast)[NT][MH],
                                        const ActStash* st = nullptr, PhaseClock* pc = nullptr,
                                        const ngm_u32x4* b3w = nullptr) {
  using LY = FieldLds<MI, MH, L, SKIP == 2>;
  static_assert(!B3 || SKIP == 0, "the bf16 split path is compiled for skip_mode no");
  if constexpr (B3) layer_fwd_b3<MI, MH, NT>(b3w + B3Lds<MI, MH, L>::off(0), sm + LY::b_off(0), lane, E, Hlast);
  else layer_fwd<MI, MH, NT>(sm + LY::w_off(0), sm + LY::b_off(0), lane, E, Hlast);
  if constexpr (SKIP == 1) skip_add<MI, MH, NT>(Hlast, E);
  PTICK(pc, 5);
  if (st && st->base) act_store<MH, NT, WT>(*st, 0, lane, Hlast);
  PTICK(pc, 6);
#pragma unroll
  for (int l = 1; l < L; ++l) {
    f32x16 T[NT][MH];
    if constexpr (SKIP == 2) {
      // concat (models.py:159-161): the layer reads cat(hidden, encoding) -- MH + MI input tiles, all in registers
      f32x16 Xc[NT][MH + MI];
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
#pragma unroll
        for (int m = 0; m < MH; ++m) Xc[nt][m] = Hlast[nt][m];
#pragma unroll
        for (int m = 0; m < MI; ++m) Xc[nt][MH + m] = E[nt][m];
      }
      layer_fwd<MH + MI, MH, NT>(sm + LY::w_off(l), sm + LY::b_off(l), lane, Xc, T);
    } else if constexpr (B3) {
      layer_fwd_b3<MH, MH, NT>(b3w + B3Lds<MI, MH, L>::off(l), sm + LY::b_off(l), lane, Hlast, T);
    } else {
      layer_fwd<MH, MH, NT>(sm + LY::w_off(l), sm + LY::b_off(l), lane, Hlast, T);
    }
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int m = 0; m < MH; ++m) Hlast[nt][m] = T[nt][m];
    if constexpr (SKIP == 1) skip_add<MI, MH, NT>(Hlast, E);
    PTICK(pc, 5);
    if (st && st->base && l < st->nlayers) act_store<MH, NT, WT>(*st, l, lane, Hlast);
    PTICK(pc, 6);
  }
}

// Evaluate the field MLP for the 64 samples owned by the 64 lanes of a wave.
// (x,y,z) = this lane's sample in scaled field-local coordinates.  Returns the 4 raw outputs.
// HASH: 0 = encoding from the per-feature table (raw / sin / cos), 1 = permutohedral hash, 2 = triplane (tc)
template <int MI, int MH, int L, bool NEED_COS, int HASH = 0, int SKIP = 0, bool B3 = false, bool WT = true>
__device__ __forceinline__ float4 eval_64(const float* sm, int lane, float x, float y, float z, const HashCtx* hc = nullptr,
                                          const ActStash* st = nullptr, PhaseClock* pc = nullptr,
                                          const ngm_u32x4* b3w = nullptr, const TriCtx* tc = nullptr) {
  using LY = FieldLds<MI, MH, L, SKIP == 2>;
  const int hi = lane >> 5;
  // partner lane (same column j, other half) owns the sample of the other tile
  const float ox = __shfl_xor(x, 32, 64), oy = __shfl_xor(y, 32, 64), oz = __shfl_xor(z, 32, 64);
  f32x16 E[2][MI], dummy[MI];
  if constexpr (HASH == 2) {
    encode_triplane<MI>(*tc, hi, hi ? ox : x, hi ? oy : y, hi ? oz : z, E[0]);
    encode_triplane<MI>(*tc, hi, hi ? x : ox, hi ? y : oy, hi ? z : oz, E[1]);
  } else if constexpr (HASH == 1) {
    static_assert(HASH != 1 || MI == 1, "hash encoding: 2*levels <= 32 features");
    encode_hash(sm + LY::ENCW, *hc, hi, hi ? ox : x, hi ? oy : y, hi ? oz : z, E[0][0]);
    encode_hash(sm + LY::ENCW, *hc, hi, hi ? x : ox, hi ? y : oy, hi ? z : oz, E[1][0]);
    // training: the encoding itself is what the backward cannot cheaply recompute (simplex search + 64 table
    // gathers per sample) -> stash it (128 B per sample); the one hidden layer is recomputed there
    if (st && st->base) act_store<MI, 2, WT>(*st, 0, lane, E);
#ifdef NGM_ABLF_HASHCELLS   // timing ablation (results meaningless): what stashing the lattice search for k_hash_grad would cost the
                            // forward -- 24 bytes (4 x 16-bit slots + 4 fp32 weights) per (sample, level), level-major, written over
                            // the encoding stash's memory (wrapped: the backward then reads garbage)
    if (st && st->base) {
      const int64_t span = (1ll << (63 - __clzll((unsigned long long)(st->layer_stride / 4 - 64)))) - 1;   // float4 slots of the stash area (power of two - 1)
      typedef float v4f __attribute__((ext_vector_type(4)));
      typedef unsigned v2u __attribute__((ext_vector_type(2)));
      v4f* w4 = reinterpret_cast<v4f*>(st->base);
      v2u* i2 = reinterpret_cast<v2u*>(st->base);
#pragma unroll
      for (int nt = 0; nt < 2; ++nt)
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          const int level = 4 * (k >> 1) + 2 * hi + (k & 1);
          const int64_t g = (st->g0 + 32 * nt + (lane & 31)) + (int64_t)level * 524288;
          if (32 * nt + (lane & 31) < st->nvalid) {
            const v4f wv = {E[nt][0][2 * k], E[nt][0][2 * k + 1], x, y};
            const v2u iv = {__float_as_uint(E[nt][0][2 * k]), (unsigned)lane};
            __builtin_nontemporal_store(wv, w4 + (g & span));
            __builtin_nontemporal_store(iv, i2 + ((2 * g + 1) & span));
          }
        }
    }
#endif
  } else {
    if constexpr (!NEED_COS) {
      const ngm_v2f X = {hi ? ox : x, hi ? x : ox}, Y = {hi ? oy : y, hi ? y : oy}, Z = {hi ? oz : z, hi ? z : oz};
      encode_pair_sin<MI>(sm + LY::ENCW, hi, X, Y, Z, E[0], E[1]);
    } else {
      encode_sample<MI, NEED_COS, false>(sm + LY::ENCW, hi, hi ? ox : x, hi ? oy : y, hi ? oz : z, E[0], dummy);
      encode_sample<MI, NEED_COS, false>(sm + LY::ENCW, hi, hi ? x : ox, hi ? y : oy, hi ? z : oz, E[1], dummy);
    }
  }
  PTICK(pc, 4);
  f32x16 Hl[2][MH];
#ifdef NGM_ABLF_NOMFMA   // timing ablation: encoding and output layer without the hidden layers
#pragma unroll
  for (int nt = 0; nt < 2; ++nt)
#pragma unroll
    for (int m = 0; m < MH; ++m) Hl[nt][m] = E[nt][m % MI];
#else
  mlp_fwd<MI, MH, L, 2, SKIP, B3, WT>(sm, lane, E, Hl, HASH == 1 ? nullptr : st, pc, b3w);
#endif
  float part[2][4];
  out_layer_partial<MH, 2>(sm + LY::WOUT, hi, Hl, part);
  if constexpr (SKIP == 2) {          // the output layer reads cat(hidden, encoding) as well
    float pe[2][4];
    out_layer_partial<MI, 2>(sm + LY::WOUT + MH * 32 * 4, hi, E, pe);
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
      for (int c = 0; c < 4; ++c) part[nt][c] += pe[nt][c];
  }
  float o[4];
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    const float send = hi ? part[0][c] : part[1][c];
    const float recv = __shfl_xor(send, 32, 64);
    const float own = hi ? part[1][c] : part[0][c];
    // fixed summation order: low-feature half first
    o[c] = (hi ? (recv + own) : (own + recv)) + sm[LY::BOUT + c];
  }
  PTICK(pc, 7);
  return make_float4(o[0], o[1], o[2], o[3]);
}

// ONE 32-sample tile: the wave step of a batch that has at most 32 samples left (round 6).  The samples are those of lanes
// 0..31; lanes 32..63 hold the other k-half of the same columns, as in tile 0 of eval_64, and every operation on that tile is
// the one eval_64 performs (encode_pair_sin with the tile's point in both slots, the layers and the output layer with NT = 1):
// the same bits for those samples, half the matrix and split work.  A rank of an 8-GPU run trains ~4 fields per iteration --
// one 24-sample ray per wave at the reference's 8 + 16 samples --, where the second tile of eval_64 is pure overhead.
// Sine-only table encodings, skip_mode no (the fused training forward's default instances).
// HASH = 1: the permutohedral encoding of the tile (one encode_hash instead of two: the simplex searches and table gathers are a
// third of the default network's forward), its stash, one hidden-layer pass on fp32 MFMA.
template <int MI, int MH, int L, bool B3, int HASH = 0>
__device__ __forceinline__ float4 eval_32(const float* sm, int lane, float x, float y, float z, const ActStash* st,
                                          const ngm_u32x4* b3w, const HashCtx* hc = nullptr) {
  using LY = FieldLds<MI, MH, L, false>;
  const int hi = lane >> 5;
  const float ox = __shfl_xor(x, 32, 64), oy = __shfl_xor(y, 32, 64), oz = __shfl_xor(z, 32, 64);
  f32x16 E[1][MI];
  const float tx = hi ? ox : x, ty = hi ? oy : y, tz = hi ? oz : z;      // the column's sample (lane & 31)
  if constexpr (HASH == 1) {
    static_assert(HASH != 1 || MI == 1, "hash encoding: 2*levels <= 32 features");
    encode_hash(sm + LY::ENCW, *hc, hi, tx, ty, tz, E[0][0]);
    if (st && st->base) act_store<MI, 1>(*st, 0, lane, E);
  } else {
    f32x16 Eo[MI];
    const ngm_v2f X = {tx, tx}, Y = {ty, ty}, Z = {tz, tz};
    encode_pair_sin<MI>(sm + LY::ENCW, hi, X, Y, Z, E[0], Eo);
  }
  f32x16 Hl[1][MH];
  mlp_fwd<MI, MH, L, 1, 0, B3>(sm, lane, E, Hl, HASH == 1 ? nullptr : st, nullptr, b3w);
  float part[1][4];
  out_layer_partial<MH, 1>(sm + LY::WOUT, hi, Hl, part);
  float o[4];
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    const float recv = __shfl_xor(part[0][c], 32, 64);
    // fixed summation order: low-feature half first (lanes 32..63 end up with the same sums; their samples do not exist)
    o[c] = (hi ? (recv + part[0][c]) : (part[0][c] + recv)) + sm[LY::BOUT + c];
  }
  return make_float4(o[0], o[1], o[2], o[3]);
}

// dispatch helper: logical (D,H,L) -> compiled (MI,MH,L) instantiation
struct FieldShape { int MI, MH, L; };
static inline FieldShape field_shape(const ngm_field_cfg* fc) {
  FieldShape s;
  s.MI = (fc->dim_enc + 31) / 32;
  s.MH = (fc->dim_hidden + 31) / 32;
  s.L = fc->num_layers;
  return s;
}
