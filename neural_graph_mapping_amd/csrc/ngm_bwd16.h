// Shared pieces of the 16-sample-tile backward kernels (ngm_field_bwd16.hip: forward recompute;
// ngm_field_bwd16s.hip: activations read from the forward's stash): LDS weight layout, MFMA tile
// helpers for lanes (j = lane & 15, q = lane >> 4), phase timer.
#pragma once
#include "ngm_field.h"
#include "ngm_launch.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

#define WAVE_SYNC()                                        \
  do {                                                     \
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); \
    __builtin_amdgcn_wave_barrier();                       \
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); \
  } while (0)

// phase timing (debug builds, -DNGM_PHASE_TIMING): TICK(k) adds the cycles since the previous TICK to
// slot k; wave 0 of block 0 reports.  Compiled out otherwise (the counters cost registers and pin
// the instruction schedule, so a build with them is ~50% slower).
#ifdef NGM_PHASE_TIMING
#define TICK_DECL                                                        \
  unsigned long long tacc[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};    \
  unsigned long long tlast = __builtin_readcyclecounter();               \
  const unsigned long long tstart = tlast
#define TICK(k)                                                          \
  do {                                                                   \
    if (a.debug_cycles) {                                                \
      const unsigned long long now_ = __builtin_readcyclecounter();      \
      tacc[k] += now_ - tlast; tlast = __builtin_readcyclecounter();     \
    }                                                                    \
  } while (0)
#define TICK_REPORT                                                               \
  if (a.debug_cycles && blockIdx.x == 0 && threadIdx.x == 0) {                    \
    for (int k = 0; k < 12; ++k) a.debug_cycles[k] = tacc[k];                     \
    a.debug_cycles[12] = __builtin_readcyclecounter() - tstart;                   \
  }
#else
#define TICK_DECL
#define TICK(k)
#define TICK_REPORT
#endif

// ---- HBM -> LDS DMA ---------------------------------------------------------------------------------
// global_load_lds_dwordx4: lane p's 16 bytes land at lds_base + 16 p.  M0 carries the LDS base.  Two address
// forms: 64-bit per-lane pointer, or wave-uniform base (SGPR pair) + 32-bit per-lane byte offset (cheaper:
// one VGPR, no 64-bit VALU arithmetic in the tile loop).
__device__ __forceinline__ void dma16(const void* gsrc, uint32_t lds_base) {
#ifdef NGM_ABLS_NOINP   // timing ablation: no input DMA (results meaningless)
  return;
#endif
  uint32_t keep;
  asm volatile(
      "s_mov_b32 %0, m0\n\t"
      "s_mov_b32 m0, %2\n\t"
      "s_nop 0\n\t"
      "global_load_lds_dwordx4 %1, off\n\t"
      "s_mov_b32 m0, %0"
      : "=&s"(keep)
      : "v"(gsrc), "s"(lds_base)
      : "memory");
}
// cache policy of the streamed (read-once) transfers: non-temporal by default; -DNGM_DMA_HINT='""' / '"sc1"' for A/B runs
#ifndef NGM_DMA_HINT
#define NGM_DMA_HINT "nt"
#endif
__device__ __forceinline__ void dma16_so(const void* sbase, uint32_t voff, uint32_t lds_base) {
#ifdef NGM_ABLS_NODMA   // timing ablation: no activation DMA at all (results meaningless)
  return;
#endif
  uint32_t keep;
  asm volatile(
      "s_mov_b32 %0, m0\n\t"
      "s_mov_b32 m0, %3\n\t"
      "s_nop 0\n\t"
      "global_load_lds_dwordx4 %1, %2 " NGM_DMA_HINT "\n\t"
      "s_mov_b32 m0, %0"
      : "=&s"(keep)
      : "v"(voff), "s"(sbase), "s"(lds_base)
      : "memory");
}
// the same without the non-temporal hint: rows many lanes (and neighbouring tiles) read again
__device__ __forceinline__ void dma16_so_c(const void* sbase, uint32_t voff, uint32_t lds_base) {
  uint32_t keep;
  asm volatile(
      "s_mov_b32 %0, m0\n\t"
      "s_mov_b32 m0, %3\n\t"
      "s_nop 0\n\t"
      "global_load_lds_dwordx4 %1, %2\n\t"
      "s_mov_b32 m0, %0"
      : "=&s"(keep)
      : "v"(voff), "s"(sbase), "s"(lds_base)
      : "memory");
}
#define DMA_WAIT(N) asm volatile("s_waitcnt vmcnt(" #N ")" ::: "memory")

// Per-field bases (wave-uniform) of everything the tile loop streams in; sample indices inside the loop are
// 32-bit and relative to the field (the API only selects this kernel when P * 256 B < 4 GiB).
struct FieldStreams {
  const char* raytab;     // + 32 B * ray-in-field
  const char* dout;       // + 16 B * n
  const char* tpair;      // 16-byte aligned (t,T,t,T) pairs: + 8 B * ((n + par) & ~1)
  const char* act[2];     // tiled stash (ngm_field.h ActStash), base of the field's first 32-sample tile
  uint32_t gb;            // (field's first global sample index) & 31
  uint32_t par;           // parity of the field's first global sample index
};

// Fused compositing backward (k_field_bwd_b3<FC>, k_hash_mlp_bwd<FC>): a wave walks its tiles back to front carrying the
// suffix value Q of the per-ray recursion Q_{k-1} = a_k o_k + (1 - o_k) Q_k.  Its range ends with a tile, not necessarily with
// a ray: this returns Q at the range's upper end, i.e. the recursion over the m samples of the cut ray that lie BEYOND `end`
// (they belong to another wave or workgroup, which will compute them again for its own purposes; m < S, same field).
// Plain loads, once per wave, 64 samples per pass back to front; k = {k_photo, k_depth, k_term} as in the tile passes.
__device__ __forceinline__ float comp_suffix_beyond(const FieldBwdArgs& a, int64_t g0, uint32_t end, int lane, float inv_s,
                                                    float k_photo, float k_depth, float k_term) {
  const int S = a.S;
  const int last = (int)end - 1;
  const int ray = fdiv_idx32(last, inv_s, S);
  const int m = S - 1 - (last - ray * S);                  // samples of that ray at or beyond `end` (wave-uniform)
  if (m <= 0) return 0.f;
  const int64_t gray = g0 / S + ray;                       // global ray index
  const float4 r1 = reinterpret_cast<const float4*>(a.raytab)[2 * gray + 1];
  const float4 q0 = reinterpret_cast<const float4*>(a.rayseed)[2 * gray], q1 = reinterpret_cast<const float4*>(a.rayseed)[2 * gray + 1];
  const float dC0 = k_photo * q0.x, dC1 = k_photo * q0.y, dC2 = k_photo * q0.z, dD = k_depth * q0.w, dT = k_term * q1.x;
  float carry = 0.f;
  for (int p = (m - 1) >> 6; p >= 0; --p) {
    const int o = 64 * p + lane;                           // offset beyond `end`
    const bool valid = o < m;
    const int64_t g = g0 + (int64_t)end + (valid ? o : 0);
    const float4 dd = a.d_out[g];
    const float2 tT = a.stashB[g];
    float dodg;
    const float occ = occ_pointwise_fast(a.rc.geometry_mode, a.rc.geometry_factor, dd.w, &dodg);
    const float ak = dC0 * dd.x + dC1 * dd.y + dC2 * dd.z + dD * (-(r1.z * tT.x)) + dT;
    float A = valid ? ak * occ : 0.f, B = valid ? 1.0f - occ : 1.0f;
    const int kr = valid ? m - 1 - o : 0;                  // later samples of the ray
    seg_rscan_affine64(A, B, kr, lane);
    const float Qend = (valid && kr > 63 - lane) ? carry : 0.f;
    carry = lane_value(fmaf(B, Qend, A), 0);
  }
  return carry;
}

// inputs of tile [n0, n0+16): lane group q fetches piece q of sample j (0: ray origin + dir.x, 1: rest of the
// ray entry, 2: d_out, 3: the aligned stash pair holding t).  One instruction, per-lane 64-bit pointers.
__device__ __forceinline__ void issue_inputs(const FieldStreams& fs, int S, uint32_t n0, uint32_t end, int lane, uint32_t lds) {
  asm volatile("" : "+v"(lane));     // re-derived per call: hoisted out of the tile loop, the per-lane 64-bit stream base (a select
                                     // over lane >> 4) was the value the 512-register kernels spilled, reloaded behind the transfers
  const int j = lane & 15, q = lane >> 4;
  uint32_t n = n0 + j;
  if (n >= end) n = end - 1;                       // clamp: finite data, its gradient contribution is zeroed via d_out
  const uint32_t ray = n / (uint32_t)S;
  const char* src;
  if (q == 0) src = fs.raytab + 32 * (size_t)ray;
  else if (q == 1) src = fs.raytab + 32 * (size_t)ray + 16;
  else if (q == 2) src = fs.dout + 16 * (size_t)n;
  else src = fs.tpair + 8 * (size_t)((n + fs.par) & ~1u);
  dma16(src, lds);
}

// global_load_lds_dword: lane p's 4 bytes land at lds_base + 4 p (per-lane 64-bit pointer)
__device__ __forceinline__ void dma4(const void* gsrc, uint32_t lds_base) {
  uint32_t keep;
  asm volatile(
      "s_mov_b32 %0, m0\n\t"
      "s_mov_b32 m0, %2\n\t"
      "s_nop 0\n\t"
      "global_load_lds_dword %1, off\n\t"
      "s_mov_b32 m0, %0"
      : "=&s"(keep)
      : "v"(gsrc), "s"(lds_base)
      : "memory");
}
// Point mode (ngm_field_eval_bwd_stash: explicit (F,P,3) points, no ray table): inputs of tile [n0, n0+16) -- d_out of sample
// j as piece 2 of the ray-mode layout (every lane group fetches it: all 64 lanes of a transfer move data), and the 48 floats
// of the 16 points as one dword transfer to lds_pts (sample j's coordinates at floats 3 j .. 3 j + 2).
__device__ __forceinline__ void issue_inputs_pts(const char* dout, const char* pts, uint32_t n0, uint32_t end, int lane, uint32_t lds,
                                                 uint32_t lds_pts) {
  asm volatile("" : "+v"(lane));
  const int j = lane & 15;
  uint32_t n = n0 + j;
  if (n >= end) n = end - 1;
  dma16(dout + 16 * (size_t)n, lds);
  const uint32_t d = min((uint32_t)lane, 47u), jj = d / 3u, c = d - 3u * jj;
  uint32_t m = n0 + jj;
  if (m >= end) m = end - 1;
  dma4(pts + 4 * (3 * (size_t)m + c), lds_pts);
}

#define B16_WAVES 8
#define B16_THREADS 512
#define B16_RS 17          // row stride of a 16x16 weight block in LDS ([k-row][out]), odd: transposed reads stay spread

__device__ __forceinline__ f32x4 mfma16(float a, float b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }

// LDS layout (floats).  TI/TH = ceil(D/16), ceil(H/16).
template <int TI, int TH, int L, int RS = B16_RS>
struct Lds16 {
  static constexpr int BLK = 16 * RS;                                     // one 16x16 block
  static constexpr int ENCW = 0;                                              // float4[TI*16]
  static constexpr int w_off(int l) {
    int o = TI * 16 * 4;
    for (int i = 0; i < l; ++i) o += TH * (i == 0 ? TI : TH) * BLK + TH * 16;
    return o;
  }
  static constexpr int b_off(int l) { return w_off(l) + TH * (l == 0 ? TI : TH) * BLK; }
  static constexpr int WOUT = b_off(L - 1) + TH * 16;                          // float4[TH*16]
  static constexpr int BOUT = WOUT + TH * 16 * 4;
  static constexpr int WTOTAL = (BOUT + 4 + 3) & ~3;
  // per-wave staging
  static constexpr int STR_E = 16 * TI + 12;                                   // conflict-friendly (mod 32 == 12)
  static constexpr int STR_H = 16 * TH + 12;
  static constexpr int STR_D = (STR_E > STR_H) ? STR_E : STR_H;
  static constexpr int x_off(int l) { return l == 0 ? 0 : 16 * STR_E + (l - 1) * 16 * STR_H; }
  static constexpr int DBUF = 16 * STR_E + (L - 1) * 16 * STR_H;
  static constexpr int PBUF = DBUF + 16 * STR_D;                               // [16][4]
  static constexpr int OBUF = PBUF + 64;                                       // [16][4]
  static constexpr int WAVE_TOTAL = OBUF + 64;
  static constexpr int TOTAL = WTOTAL + B16_WAVES * WAVE_TOTAL;
};

// weights -> LDS in two phases (B16_THREADS-thread workgroups): issue() starts every global load as straight-line
// code (16-byte loads for the matrices when the rows allow it), commit() writes the LDS image: block (mo, mi) holds
// W[16mo + o][16mi + c] at row krow(c) = 4*(c&3) + (c>>2), col o.  (Loops that loaded and stored element by element
// cost the stash backward ~13 k clocks before its first MFMA.)  Caller must __syncthreads() after commit().
// SWZ: the four 16-byte column groups of LDS row r sit at positions (group ^ (r >> 2)) -- with RS = 16 that makes the
// 16-byte reads of dgrad16v (k_field_bwd16s) bank-conflict free (they were 2-way with the padded rows).
template <int TI, int TH, int L, int RS = B16_RS, bool SWZ = false>
struct FieldStage16 {
  static constexpr int NIT = (TH * 16 * TH * 4 + B16_THREADS - 1) / B16_THREADS;   // 16-byte chunks per thread and layer
  float4 v[L][NIT];
  float4 enc0, enc1, wout;
  float bias[L];
  __device__ __forceinline__ void issue(const ngm_field_cfg& fc, const ngm_params& pr, int64_t row) {
    const int tid = threadIdx.x;
    const int D = fc.dim_enc, H = fc.dim_hidden;
#pragma unroll
    for (int l = 0; l < L; ++l) {
      const int TIN = (l == 0) ? TI : TH, Din = (l == 0) ? D : H;
      const int dt = pr.dtype;
      const float* W = pr.w[l];                          // element offsets from here: the storage may be 16-bit
      const int64_t w0 = row * pr.w_stride[l];
      const int ncol4 = TIN * 4, total4 = TH * 16 * ncol4;
      const uintptr_t wa = reinterpret_cast<uintptr_t>(W) + (uintptr_t)w0 * (dt == NGM_DT_F32 ? 4 : 2);
      const bool vec = ((Din & 3) == 0) && ((wa & (dt == NGM_DT_F32 ? 15 : 7)) == 0);
#pragma unroll
      for (int it = 0; it < NIT; ++it) {
        const int e4 = tid + it * B16_THREADS;
        float4 x = make_float4(0.f, 0.f, 0.f, 0.f);
        if (e4 < total4) {
          const int o = e4 / ncol4, c = 4 * (e4 - o * ncol4);
          if (o < H && c < Din) x = ngm_ldp4(W, w0 + (int64_t)o * Din + c, dt, vec, Din - c);
        }
        v[l][it] = x;
      }
      bias[l] = (tid < H) ? ngm_ldp(pr.b[l], row * pr.b_stride[l] + tid, dt) : 0.f;
    }
    const int dt = pr.dtype;
    wout = make_float4(0.f, 0.f, 0.f, 0.f);
    if (tid < H) {
      const float* W = pr.w[L];
      const int64_t w0 = row * pr.w_stride[L];
      wout = make_float4(ngm_ldp(W, w0 + tid, dt), ngm_ldp(W, w0 + H + tid, dt), ngm_ldp(W, w0 + 2 * H + tid, dt),
                         ngm_ldp(W, w0 + 3 * H + tid, dt));
    }
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
        if (f < n_raw) enc0 = make_float4(f == 0 ? 1.f : 0.f, f == 1 ? 1.f : 0.f, f == 2 ? 1.f : 0.f, NGM_FK_RAW);
        else {
          const int64_t e0 = row * pr.enc_w_stride + (int64_t)(f - n_raw) * 3;
          enc0 = make_float4(ngm_ldp(pr.enc_w, e0, dt), ngm_ldp(pr.enc_w, e0 + 1, dt), ngm_ldp(pr.enc_w, e0 + 2, dt), NGM_FK_SIN);
        }
      } else if (fc.encoding == NGM_ENC_NERF) {
        const int half = 3 * fc.num_octaves;
        const int g = (f < half) ? f : f - half;
        const int d = g / fc.num_octaves, o = g % fc.num_octaves;
        const float m = exp2f((float)(fc.start_octave + o)) * 3.14159265358979323846f;
        enc0 = make_float4(d == 0 ? m : 0.f, d == 1 ? m : 0.f, d == 2 ? m : 0.f, (f < half) ? NGM_FK_SIN : NGM_FK_COS);
      } else enc0 = make_float4(f == 0 ? 1.f : 0.f, f == 1 ? 1.f : 0.f, f == 2 ? 1.f : 0.f, NGM_FK_RAW);
    }
  }
  __device__ __forceinline__ void commit(float* sm, const ngm_field_cfg& fc) const {
    using LY = Lds16<TI, TH, L, RS>;
    const int tid = threadIdx.x;
    if (fc.encoding == NGM_ENC_PERMUTO) {
      if (tid < 16) {
        reinterpret_cast<float4*>(sm + LY::ENCW)[2 * tid] = enc0;
        reinterpret_cast<float4*>(sm + LY::ENCW)[2 * tid + 1] = enc1;
      }
    } else if (tid < TI * 16) {
      reinterpret_cast<float4*>(sm + LY::ENCW)[tid] = enc0;
    }
#pragma unroll
    for (int l = 0; l < L; ++l) {
      const int TIN = (l == 0) ? TI : TH;
      const int ncol4 = TIN * 4, total4 = TH * 16 * ncol4;
      float* dst = sm + LY::w_off(l);
#pragma unroll
      for (int it = 0; it < NIT; ++it) {
        const int e4 = tid + it * B16_THREADS;
        if (e4 < total4) {
          const int o = e4 / ncol4, c = 4 * (e4 - o * ncol4);
          const int mo = o >> 4, ol = o & 15, mi = c >> 4, cl = c & 15;
          float* blk = dst + (mo * TIN + mi) * LY::BLK + (cl >> 2) * RS;          // row 4*j + (cl >> 2) for column cl + j
          const float x[4] = {v[l][it].x, v[l][it].y, v[l][it].z, v[l][it].w};
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const int col = SWZ ? ((((ol >> 2) ^ j) << 2) | (ol & 3)) : ol;       // row >> 2 == j
            blk[4 * j * RS + col] = x[j];
          }
        }
      }
      if (tid < TH * 16) sm[LY::b_off(l) + tid] = bias[l];
    }
    if (tid < TH * 16) reinterpret_cast<float4*>(sm + LY::WOUT)[tid] = wout;
  }
};

// ---- tile helpers (lane (j = lane&15, q = lane>>4)) ---------------------------------------------------
// HW = true (backward with stashed activations): sine and cosine from v_sin_f32 / v_cos_f32.  Those take the
// argument in revolutions and reduce it themselves, so the position is scaled by 1/(2 pi) once per sample and
// the whole software reduction + two polynomials (28 VALU per value) become 3 FMAs + v_fract + 2 quarter-rate
// instructions.  Max abs error ~4e-7 + 1.2e-7 |arg| (tools/micro/hwsin.hip) -- inside the gradient tolerance
// (2e-3 relative), not inside the forward's, which keeps the polynomial.
template <int T, bool NEED_COS, bool WITH_DERIV, bool HW = false>
__device__ __forceinline__ void encode16(const float* sm_encw, int q, float x, float y, float z, f32x4 (&E)[T], f32x4 (&dE)[T]) {
  const float4* tab = reinterpret_cast<const float4*>(sm_encw);
  const float inv2pi = 0.15915494309189535f;
  const float xr = x * inv2pi, yr = y * inv2pi, zr = z * inv2pi;
#pragma unroll
  for (int m = 0; m < T; ++m) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float4 w = tab[16 * m + 4 * q + r];
      float s, c, arg;
      if constexpr (HW) {
        const float rev = __builtin_amdgcn_fractf(fmaf(w.z, zr, fmaf(w.y, yr, w.x * xr)));
        s = __builtin_amdgcn_sinf(rev);
        c = __builtin_amdgcn_cosf(rev);
        arg = (r == 0) ? x : (r == 1) ? y : z;     // only read for raw-coordinate rows, whose w is the unit vector e_r
      } else {
        arg = fmaf(w.z, z, fmaf(w.y, y, w.x * x));
        ngm_sincosf(arg, &s, &c);
      }
      float v = s, d = c;
      if (NEED_COS) { const bool is_cos = (w.w == NGM_FK_COS); v = is_cos ? c : s; d = is_cos ? -s : c; }
      if (m == 0 && r < 3) { const bool raw = (w.w == NGM_FK_RAW); v = raw ? arg : v; d = raw ? 0.f : d; }
      E[m][r] = v;
      if (WITH_DERIV) dE[m][r] = d;
    }
    __builtin_amdgcn_sched_barrier(0);
  }
}

// hash levels of this lane: tile m, pair p -> level 8m + 2q + p (features 16m + 4q + 2p + {0,1})
template <int T>
__device__ __forceinline__ void encode_hash16(const float* sm_lvl, const HashCtx& hc, int q, float x, float y, float z, f32x4 (&E)[T]) {
#pragma unroll
  for (int m = 0; m < T; ++m)
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      const int level = 8 * m + 2 * q + p;
      float f0 = 0.f, f1 = 0.f;
      if (level < hc.nlev) {
        uint32_t idx[4]; float bw[4];
        permuto_simplex(x, y, z, sm_lvl + 8 * level, hc.mask, idx, bw);
        float2 v[4];
        ngm_ldp2x4(hc.tab, (size_t)level * hc.T, idx, hc.dt, v);
#pragma unroll
        for (int r = 0; r < 4; ++r) { f0 = fmaf(v[r].x, bw[r], f0); f1 = fmaf(v[r].y, bw[r], f1); }
      }
      E[m][2 * p] = f0; E[m][2 * p + 1] = f1;
    }
}

template <int T>
__device__ __forceinline__ void store16(float* buf, int stride, int lane, const f32x4 (&V)[T]) {
  const int j = lane & 15, q = lane >> 4;
#pragma unroll
  for (int m = 0; m < T; ++m)
    *reinterpret_cast<float4*>(buf + j * stride + 16 * m + 4 * q) = make_float4(V[m][0], V[m][1], V[m][2], V[m][3]);
}
template <int T>
__device__ __forceinline__ void load16(const float* buf, int stride, int lane, f32x4 (&V)[T]) {
  const int j = lane & 15, q = lane >> 4;
#pragma unroll
  for (int m = 0; m < T; ++m) {
    const float4 v = *reinterpret_cast<const float4*>(buf + j * stride + 16 * m + 4 * q);
    V[m][0] = v.x; V[m][1] = v.y; V[m][2] = v.z; V[m][3] = v.w;
  }
}

// Y = relu(W X + b): k-step (mi, r) uses A = W[16mo + i][16mi + 4q + r] = block(mo,mi)[row 4r + q][i]
template <int TIN, int TOUT, int BLK>
__device__ __forceinline__ void fwd16(const float* __restrict__ W, const float* __restrict__ B, int lane,
                                      const f32x4 (&X)[TIN], f32x4 (&Y)[TOUT]) {
  const int i = lane & 15, q = lane >> 4;
#pragma unroll
  for (int mo = 0; mo < TOUT; ++mo) {
    const float4 b4 = *reinterpret_cast<const float4*>(B + 16 * mo + 4 * q);
    Y[mo][0] = b4.x; Y[mo][1] = b4.y; Y[mo][2] = b4.z; Y[mo][3] = b4.w;
  }
  const float* Wl = W + q * B16_RS + i;
  float abuf[2][TOUT][4];
#pragma unroll
  for (int mo = 0; mo < TOUT; ++mo)
#pragma unroll
    for (int r = 0; r < 4; ++r) abuf[0][mo][r] = Wl[(mo * TIN + 0) * BLK + 4 * r * B16_RS];
#pragma unroll
  for (int mi = 0; mi < TIN; ++mi) {
    if (mi + 1 < TIN) {
#pragma unroll
      for (int mo = 0; mo < TOUT; ++mo)
#pragma unroll
        for (int r = 0; r < 4; ++r) abuf[(mi + 1) & 1][mo][r] = Wl[(mo * TIN + mi + 1) * BLK + 4 * r * B16_RS];
    }
    __builtin_amdgcn_sched_barrier(0);   // keep the prefetch ABOVE this group's MFMAs (the scheduler sinks loads to their use)
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int mo = 0; mo < TOUT; ++mo) Y[mo] = mfma16(abuf[mi & 1][mo][r], X[mi][r], Y[mo]);
    __builtin_amdgcn_sched_barrier(0);
  }
#pragma unroll
  for (int mo = 0; mo < TOUT; ++mo)
#pragma unroll
    for (int r = 0; r < 4; ++r) Y[mo][r] = ngm_relu(Y[mo][r]);
}

// dX = W^T dY: k-step (mo, r) uses A[i][k=q] = W[16mo + 4q + r][16mi + i] = block(mo,mi)[row 4(i&3)+(i>>2)][4q + r]
template <int TIN, int TOUT, int BLK>
__device__ __forceinline__ void dgrad16(const float* __restrict__ W, int lane, const f32x4 (&dY)[TOUT], f32x4 (&dX)[TIN]) {
  const int i = lane & 15, q = lane >> 4;
  const float* Wl = W + (4 * (i & 3) + (i >> 2)) * B16_RS + 4 * q;
#pragma unroll
  for (int mi = 0; mi < TIN; ++mi) dX[mi] = f32x4{0.f, 0.f, 0.f, 0.f};
  float abuf[2][TIN][4];
#pragma unroll
  for (int mi = 0; mi < TIN; ++mi)
#pragma unroll
    for (int r = 0; r < 4; ++r) abuf[0][mi][r] = Wl[(0 * TIN + mi) * BLK + r];
#pragma unroll
  for (int mo = 0; mo < TOUT; ++mo) {
    if (mo + 1 < TOUT) {
#pragma unroll
      for (int mi = 0; mi < TIN; ++mi)
#pragma unroll
        for (int r = 0; r < 4; ++r) abuf[(mo + 1) & 1][mi][r] = Wl[((mo + 1) * TIN + mi) * BLK + r];
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int mi = 0; mi < TIN; ++mi) dX[mi] = mfma16(abuf[mo & 1][mi][r], dY[mo][r], dX[mi]);
    __builtin_amdgcn_sched_barrier(0);
  }
}

// dW[16mo + o][16mi + c] += sum_s dY[o][s] X[c][s]; k-step t covers samples 4t + q
template <int TOUT, int TIN>
__device__ __forceinline__ void wgrad16(const float* __restrict__ dbuf, int dstr, const float* __restrict__ xbuf, int xstr,
                                        int lane, f32x4 (&acc)[TOUT][TIN]) {
  const int i = lane & 15, q = lane >> 4;
  const float* dl = dbuf + q * dstr + i;
  const float* xl = xbuf + q * xstr + i;
  float av[2][TOUT], bv[2][TIN];
#pragma unroll
  for (int mo = 0; mo < TOUT; ++mo) av[0][mo] = dl[16 * mo];
#pragma unroll
  for (int mi = 0; mi < TIN; ++mi) bv[0][mi] = xl[16 * mi];
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    if (t + 1 < 4) {
#pragma unroll
      for (int mo = 0; mo < TOUT; ++mo) av[(t + 1) & 1][mo] = dl[4 * (t + 1) * dstr + 16 * mo];
#pragma unroll
      for (int mi = 0; mi < TIN; ++mi) bv[(t + 1) & 1][mi] = xl[4 * (t + 1) * xstr + 16 * mi];
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int mo = 0; mo < TOUT; ++mo)
#pragma unroll
      for (int mi = 0; mi < TIN; ++mi) acc[mo][mi] = mfma16(av[t & 1][mo], bv[t & 1][mi], acc[mo][mi]);
    __builtin_amdgcn_sched_barrier(0);
  }
}

// lane = feature (T*16 features; the 64/(T*16) lane groups split the 16 samples)
template <int T>
__device__ __forceinline__ float colsum16(const float* buf, int stride, int lane) {
  constexpr int NF = 16 * T, PARTS = (64 % NF == 0) ? 64 / NF : 1, PER = 16 / PARTS;
  if (lane >= NF * PARTS) return 0.f;                       // NF = 48: lanes 48..63 idle
  const int fl = lane % NF, sp = lane / NF;
  float v[PER];
#pragma unroll
  for (int k = 0; k < PER; ++k) v[k] = buf[(sp * PER + k) * stride + fl];
#pragma unroll
  for (int w = PER / 2; w >= 1; w >>= 1)
#pragma unroll
    for (int k = 0; k < w; ++k) v[k] += v[k + w];
  return v[0];
}
template <int T, int NC>
__device__ __forceinline__ void outer16(const float* colbuf, int stride, const float* row4, int lane, float (&acc)[NC]) {
  constexpr int NF = 16 * T, PARTS = (64 % NF == 0) ? 64 / NF : 1, PER = 16 / PARTS;
  if (lane >= NF * PARTS) return;
  const int fl = lane % NF, sp = lane / NF;
  constexpr int BATCH = (PER < 8) ? PER : 8;
#pragma unroll
  for (int k0 = 0; k0 < PER; k0 += BATCH) {
    float h[BATCH]; float4 d[BATCH];
#pragma unroll
    for (int k = 0; k < BATCH; ++k) { const int s = sp * PER + k0 + k; h[k] = colbuf[s * stride + fl]; d[k] = *reinterpret_cast<const float4*>(row4 + 4 * s); }
#pragma unroll
    for (int k = 0; k < BATCH; ++k) {
      acc[0] = fmaf(d[k].x, h[k], acc[0]); acc[1] = fmaf(d[k].y, h[k], acc[1]); acc[2] = fmaf(d[k].z, h[k], acc[2]);
      if (NC > 3) acc[3] = fmaf(d[k].w, h[k], acc[3]);
    }
  }
}
// combine the lane groups that hold the same feature (PARTS > 1)
template <int T>
__device__ __forceinline__ float fold_parts(float v) {
  constexpr int NF = 16 * T;
  if (NF <= 32) v += __shfl_down(v, 32, 64);
  if (NF <= 16) v += __shfl_down(v, 16, 64);
  return v;
}

// Epilogue shared by the 16-sample-tile backward kernels: the 8 waves' accumulators are summed in fixed
// wave order with all threads busy.  Each round every wave parks EPI_CH of its C tiles in LDS
// ([wave][tile][reg][lane], conflict-free), then the 512 threads each add the 8 copies of two elements
// and write the result to the workgroup's partial vector in HBM.  `stage` = LDS scratch of at least
// 8 * 1024 floats; the caller has passed a __syncthreads() after its last use of that memory.
// EPI_CH: C tiles parked per wave and round (stage must hold 8 * EPI_CH * 256 floats); more tiles per round = fewer
// barrier pairs (4 -> 16 took the epilogue of k_field_bwd16s from 8 rounds to 2).
template <int TI, int TH, int L, bool ENC_GRAD, int EPI_CH = 4>
__device__ __forceinline__ void bwd16_epilogue(const FieldBwdArgs& a, float* stage, f32x4 (&acc0)[TH][TI],
                                               f32x4 (&accH)[(L > 1) ? (L - 1) : 1][TH][TH], float (&dbh)[L],
                                               float (&dwo)[4], float (&dwf)[3], float (&dbo)[4]) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int64_t enc_off, w_off[NGM_MAX_LAYERS + 1], b_off[NGM_MAX_LAYERS + 1];
  const int64_t ptot = ngm_param_offsets(&a.fc, &enc_off, w_off, b_off);
  (void)ptot;
  float* dst = a.partials + (int64_t)blockIdx.x * a.p_pad;
  const int D = a.fc.dim_enc, H = a.fc.dim_hidden;
  constexpr int NT0 = TH * TI, NTH = TH * TH, NT = NT0 + (L - 1) * NTH;
  constexpr int EPI_W = EPI_CH * 4 * 64;                   // floats parked per wave and round
#pragma unroll
  for (int t0 = 0; t0 < NT; t0 += EPI_CH) {
#pragma unroll
    for (int tt = 0; tt < EPI_CH; ++tt) {
      const int t = t0 + tt;
      if (t < NT) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float v;
          if (t < NT0) v = acc0[t / TI][t % TI][r];
          else v = accH[(t - NT0) / NTH][((t - NT0) % NTH) / TH][(t - NT0) % TH][r];
          stage[wave * EPI_W + (tt * 4 + r) * 64 + lane] = v;
        }
      }
    }
    __syncthreads();
    for (int e = threadIdx.x; e < EPI_W; e += B16_THREADS) {
      float s0 = 0.f;
#pragma unroll
      for (int w = 0; w < B16_WAVES; ++w) s0 += stage[w * EPI_W + e];
      const int t = t0 + (e >> 8), r = (e >> 6) & 3, jj = e & 15, qq = (e >> 4) & 3;
      if (t < NT) {
        // C fragment: lane (c_local = jj, qq), reg r -> dW[16mo + 4qq + r][16mi + jj]
        int l, mo, mi, din;
        if (t < NT0) { l = 0; mo = t / TI; mi = t % TI; din = D; }
        else { const int u = t - NT0; l = 1 + u / NTH; mo = (u % NTH) / TH; mi = u % TH; din = H; }
        const int o = 16 * mo + 4 * qq + r, c = 16 * mi + jj;
        if (o < H && c < din) dst[w_off[l] + (int64_t)o * din + c] = s0;
      }
    }
    __syncthreads();
  }
  // per-feature vectors: hidden biases, output weights (4 rows), Fourier matrix (3 columns), output bias
#pragma unroll
  for (int l = 0; l < L; ++l) dbh[l] = fold_parts<TH>(dbh[l]);
#pragma unroll
  for (int c = 0; c < 4; ++c) dwo[c] = fold_parts<TH>(dwo[c]);
#pragma unroll
  for (int c = 0; c < 3; ++c) dwf[c] = fold_parts<TI>(dwf[c]);
#pragma unroll
  for (int c = 0; c < 4; ++c) dbo[c] = wave_sum(dbo[c]);
  constexpr int NV = L + 4 + 3 + 4;
  static_assert(NV * 64 <= EPI_W, "per-feature vectors must fit one staging round");
  {
    float* sw = stage + wave * EPI_W;
#pragma unroll
    for (int l = 0; l < L; ++l) sw[l * 64 + lane] = dbh[l];
#pragma unroll
    for (int c = 0; c < 4; ++c) sw[(L + c) * 64 + lane] = dwo[c];
#pragma unroll
    for (int c = 0; c < 3; ++c) sw[(L + 4 + c) * 64 + lane] = dwf[c];
#pragma unroll
    for (int c = 0; c < 4; ++c) sw[(L + 7 + c) * 64 + lane] = dbo[c];
  }
  __syncthreads();
  const bool fourier = ENC_GRAD && a.fc.encoding == NGM_ENC_FOURIER;
  const int n_raw = a.fc.raw_coords ? 3 : 0;
  for (int e = threadIdx.x; e < NV * 64; e += B16_THREADS) {
    float s0 = 0.f;
#pragma unroll
    for (int w = 0; w < B16_WAVES; ++w) s0 += stage[w * EPI_W + e];
    const int k = e >> 6, ln = e & 63;
    if (k < L) { if (ln < H) dst[b_off[k] + ln] = s0; }
    else if (k < L + 4) { if (ln < H) dst[w_off[L] + (int64_t)(k - L) * H + ln] = s0; }
    else if (k < L + 7) { if (fourier && ln < D && ln >= n_raw) dst[enc_off + (int64_t)(ln - n_raw) * 3 + (k - L - 4)] = s0; }
    else if (ln == 0) dst[b_off[L] + (k - L - 7)] = s0;
  }
  if (!fourier)   // the encoding slot of the partial vector (if any) carries no gradient
    for (int64_t p = enc_off + threadIdx.x; p < w_off[0]; p += B16_THREADS) dst[p] = 0.f;
}
