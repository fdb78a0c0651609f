// Backward of encoding + MLP with 16-sample tiles on v_mfma_f32_16x16x4_f32 (gfx950).
//
// Same algorithm as k_field_bwd (ngm_field_bwd.hip) but tiled 16 samples wide so that a workgroup of
// EIGHT waves (two per SIMD) fits: per-wave staging buffers shrink to [16][F+12] floats and the
// activation registers halve, keeping every wave under 256 VGPRs.  With two waves per SIMD the VALU
// phases of one wave (sincos, masks, column sums) and its LDS waits hide under the MFMAs of the other.
//
// Fragment maps of v_mfma_f32_16x16x4_f32 (cdna_hip_programming.md section 3):
//   A: lane l holds A[i = l&15][k = l>>4]     B: lane l holds B[k = l>>4][j = l&15]
//   C/D: lane l, reg r holds C[row = 4*(l>>4) + r][col = l&15]
// Features on M, samples on N: lane (j = l&15, q = l>>4) holds feature 16*m + 4*q + r of sample j in
// register r of tile m -- exactly the B operand of k-step (m, r), so layers chain without data movement.
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

#define B16_WAVES 8
#define B16_THREADS 512
#define B16_RS 17          // row stride of a 16x16 weight block in LDS ([k-row][out]), odd: transposed reads stay spread

__device__ __forceinline__ f32x4 mfma16(float a, float b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }

// LDS layout (floats).  TI/TH = ceil(D/16), ceil(H/16).
template <int TI, int TH, int L>
struct Lds16 {
  static constexpr int BLK = 16 * B16_RS;                                     // one 16x16 block
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

// weights -> LDS: block (mo, mi) holds W[16mo + o][16mi + c] at row krow(c) = 4*(c&3) + (c>>2), col o
template <int TI, int TH, int L>
__device__ __forceinline__ void load_field16(float* sm, const ngm_field_cfg& fc, const ngm_params& pr, int64_t row) {
  using LY = Lds16<TI, TH, L>;
  const int tid = threadIdx.x, nthr = blockDim.x;
  const int D = fc.dim_enc, H = fc.dim_hidden;
  if (fc.encoding == NGM_ENC_PERMUTO) {
    for (int l = tid; l < 16; l += nthr) {
      float4 sc = make_float4(0.f, 0.f, 0.f, 0.f), sh = make_float4(0.f, 0.f, 0.f, 0.f);
      if (l < fc.nr_levels) {
        const float* hs = pr.shift + row * pr.shift_stride + 3 * l;
        sc = make_float4(fc.level_scale[3 * l], fc.level_scale[3 * l + 1], fc.level_scale[3 * l + 2], 0.f);
        sh = make_float4(hs[0], hs[1], hs[2], 0.f);
      }
      reinterpret_cast<float4*>(sm + LY::ENCW)[2 * l] = sc;
      reinterpret_cast<float4*>(sm + LY::ENCW)[2 * l + 1] = sh;
    }
  } else {
    for (int f = tid; f < TI * 16; f += nthr) {
      float4 e = make_float4(0.f, 0.f, 0.f, NGM_FK_ZERO);
      if (f < D) {
        if (fc.encoding == NGM_ENC_FOURIER) {
          const int n_raw = fc.raw_coords ? 3 : 0;
          if (f < n_raw) e = make_float4(f == 0 ? 1.f : 0.f, f == 1 ? 1.f : 0.f, f == 2 ? 1.f : 0.f, NGM_FK_RAW);
          else { const float* w = pr.enc_w + row * pr.enc_w_stride + (int64_t)(f - n_raw) * 3; e = make_float4(w[0], w[1], w[2], NGM_FK_SIN); }
        } else if (fc.encoding == NGM_ENC_NERF) {
          const int half = 3 * fc.num_octaves;
          const int g = (f < half) ? f : f - half;
          const int d = g / fc.num_octaves, o = g % fc.num_octaves;
          const float m = exp2f((float)(fc.start_octave + o)) * 3.14159265358979323846f;
          e = make_float4(d == 0 ? m : 0.f, d == 1 ? m : 0.f, d == 2 ? m : 0.f, (f < half) ? NGM_FK_SIN : NGM_FK_COS);
        } else e = make_float4(f == 0 ? 1.f : 0.f, f == 1 ? 1.f : 0.f, f == 2 ? 1.f : 0.f, NGM_FK_RAW);
      }
      reinterpret_cast<float4*>(sm + LY::ENCW)[f] = e;
    }
  }
#pragma unroll
  for (int l = 0; l < L; ++l) {
    const int TIN = (l == 0) ? TI : TH, Din = (l == 0) ? D : H;
    const float* W = pr.w[l] + row * pr.w_stride[l];
    const float* B = pr.b[l] + row * pr.b_stride[l];
    float* dst = sm + LY::w_off(l);
    const int ncol = TIN * 16, total = TH * 16 * ncol;
    for (int e = tid; e < total; e += nthr) {
      const int o = e / ncol, c = e - o * ncol;
      const float v = (o < H && c < Din) ? W[(int64_t)o * Din + c] : 0.f;
      const int mo = o >> 4, ol = o & 15, mi = c >> 4, cl = c & 15;
      dst[(mo * TIN + mi) * LY::BLK + (4 * (cl & 3) + (cl >> 2)) * B16_RS + ol] = v;
    }
    for (int o = tid; o < TH * 16; o += nthr) sm[LY::b_off(l) + o] = (o < H) ? B[o] : 0.f;
  }
  {
    const float* W = pr.w[L] + row * pr.w_stride[L];
    for (int f = tid; f < TH * 16; f += nthr) {
      float4 w4 = make_float4(0.f, 0.f, 0.f, 0.f);
      if (f < H) w4 = make_float4(W[f], W[H + f], W[2 * H + f], W[3 * H + f]);
      reinterpret_cast<float4*>(sm + LY::WOUT)[f] = w4;
    }
  }
}

// ---- tile helpers (lane (j = lane&15, q = lane>>4)) ---------------------------------------------------
template <int T, bool NEED_COS, bool WITH_DERIV>
__device__ __forceinline__ void encode16(const float* sm_encw, int q, float x, float y, float z, f32x4 (&E)[T], f32x4 (&dE)[T]) {
  const float4* tab = reinterpret_cast<const float4*>(sm_encw);
#pragma unroll
  for (int m = 0; m < T; ++m) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float4 w = tab[16 * m + 4 * q + r];
      const float arg = fmaf(w.z, z, fmaf(w.y, y, w.x * x));
      float s, c;
      ngm_sincosf(arg, &s, &c);
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
        const float2* t = hc.tab + (size_t)level * hc.T;
        float2 v[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = t[idx[r]];
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
    for (int r = 0; r < 4; ++r) Y[mo][r] = fmaxf(Y[mo][r], 0.f);
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

// raw per-sample inputs of one tile column (ray table entry + distance, or the query point) and d_out.
// (An LDS-DMA variant of this prefetch, global_load_lds_dwordx4 into a dead staging tile, was measured
// 70% slower: hipcc then guards every later ds_read with vmcnt(0) and stops interleaving them.)
struct RawIn16 {
  float4 r0, r1, dout;
  float t;
  bool valid;
};

__device__ __forceinline__ RawIn16 fetch_inputs16(const FieldBwdArgs& a, int f, int64_t n, int64_t end, bool ray_mode,
                                                  uint32_t rays_per_field) {
  RawIn16 in;
  in.valid = n < end;
  in.r0 = in.r1 = in.dout = make_float4(0.f, 0.f, 0.f, 0.f);
  in.t = 0.f;
  if (in.valid) {
    const int64_t g = (int64_t)f * a.P + n;
    if (ray_mode) {
      const int64_t ray = (int64_t)f * rays_per_field + (uint32_t)n / (uint32_t)a.S;
      in.r0 = reinterpret_cast<const float4*>(a.raytab)[2 * ray];
      in.r1 = reinterpret_cast<const float4*>(a.raytab)[2 * ray + 1];
      in.t = a.stashB[g].x;
    } else {
      const float* p = a.points + g * 3;
      in.r0 = make_float4(p[0], p[1], p[2], 0.f);
    }
    in.dout = a.d_out[g];
  }
  return in;
}

template <int TI, int TH, int L, bool NEED_COS, bool ENC_GRAD, bool HASH>
__global__ __launch_bounds__(B16_THREADS) void k_field_bwd16(FieldBwdArgs a) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  using LY = Lds16<TI, TH, L>;
  constexpr int BLK = LY::BLK;
  const int f = blockIdx.x % a.F, chunk = blockIdx.x / a.F;
  const int64_t row = a.pr.field_index ? a.pr.field_index[f] : f;
  load_field16<TI, TH, L>(sm, a.fc, a.pr, row);
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int j = lane & 15, q = lane >> 4;
  float* wl = sm + LY::WTOTAL + wave * LY::WAVE_TOTAL;
  float* bufD = wl + LY::DBUF;
  float* pbuf = wl + LY::PBUF;
  float* obuf = wl + LY::OBUF;

  float div, off;
  scale_consts(a.fc.scale_mode, a.fc.field_radius, &div, &off);
  const bool ray_mode = a.points == nullptr;
  const bool posed = a.pos != nullptr;
  float px = 0, py = 0, pz = 0, qw = 1, qx = 0, qy = 0, qz = 0;
  if (!ray_mode && posed) {
    px = a.pos[3 * f]; py = a.pos[3 * f + 1]; pz = a.pos[3 * f + 2];
    qw = a.quat[4 * f]; qx = a.quat[4 * f + 1]; qy = a.quat[4 * f + 2]; qz = a.quat[4 * f + 3];
  }

  f32x4 acc0[TH][TI];
  f32x4 accH[(L > 1) ? (L - 1) : 1][TH][TH];
#pragma unroll
  for (int mo = 0; mo < TH; ++mo) {
#pragma unroll
    for (int mi = 0; mi < TI; ++mi) acc0[mo][mi] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int l = 0; l < L - 1; ++l)
#pragma unroll
      for (int mi = 0; mi < TH; ++mi) accH[l][mo][mi] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  float dbh[L], dwo[4], dbo[4], dwf[3];
#pragma unroll
  for (int l = 0; l < L; ++l) dbh[l] = 0.f;
#pragma unroll
  for (int c = 0; c < 4; ++c) { dwo[c] = 0.f; dbo[c] = 0.f; }
  dwf[0] = dwf[1] = dwf[2] = 0.f;

  const HashCtx hc = make_hash_ctx(a.fc, a.pr, row, nullptr);
  TICK_DECL;
  TICK(0);   // prologue (weights -> LDS)
  const int64_t beg = (int64_t)chunk * a.per_block, end = min(a.P, beg + a.per_block);
  const uint32_t rays_per_field = ray_mode ? (uint32_t)(a.P / a.S) : 0u;
  RawIn16 nxt = fetch_inputs16(a, f, beg + wave * 16 + j, end, ray_mode, rays_per_field);
  for (int64_t base = beg + wave * 16; base < end; base += 16 * B16_WAVES) {
    // software pipeline: this tile's raw inputs were fetched one iteration ago; issue the next tile's
    // loads now so that their HBM latency hides behind this tile's arithmetic
    const RawIn16 cur = nxt;
    const int64_t n = base + j;
    nxt = fetch_inputs16(a, f, n + 16 * B16_WAVES, end, ray_mode, rays_per_field);
    const bool valid = cur.valid;
    float x = 0, y = 0, z = 0;
    float4 dout = make_float4(0.f, 0.f, 0.f, 0.f);
    if (valid) {
      if (ray_mode) {
        const float t = cur.t;
        x = fmaf(t, cur.r0.w, cur.r0.x); y = fmaf(t, cur.r1.x, cur.r0.y); z = fmaf(t, cur.r1.y, cur.r0.z);
      } else {
        Vec3 v{cur.r0.x, cur.r0.y, cur.r0.z};
        if (posed) { v = Vec3{v.x - px, v.y - py, v.z - pz}; v = quat_rotate_inv(qw, qx, qy, qz, v); }
        x = v.x / div + off; y = v.y / div + off; z = v.z / div + off;
      }
      dout = cur.dout;
    }
    TICK(1);   // sample inputs (global loads)
    // ---- forward recompute
    f32x4 E[TI], dEa[TI];
    if constexpr (HASH) encode_hash16<TI>(sm + LY::ENCW, hc, q, x, y, z, E);
    else encode16<TI, NEED_COS, ENC_GRAD>(sm + LY::ENCW, q, x, y, z, E, dEa);
    TICK(2);   // encode (sincos)
    WAVE_SYNC();
    store16<TI>(wl + LY::x_off(0), LY::STR_E, lane, E);
    if (q == 0) {
      *reinterpret_cast<float4*>(pbuf + 4 * j) = make_float4(x, y, z, 0.f);
      *reinterpret_cast<float4*>(obuf + 4 * j) = dout;
    }
    f32x4 Hc[TH];
    fwd16<TI, TH, BLK>(sm + LY::w_off(0), sm + LY::b_off(0), lane, E, Hc);
#pragma unroll
    for (int l = 1; l < L; ++l) {
      store16<TH>(wl + LY::x_off(l), LY::STR_H, lane, Hc);
      f32x4 Hn[TH];
      fwd16<TH, TH, BLK>(sm + LY::w_off(l), sm + LY::b_off(l), lane, Hc, Hn);
#pragma unroll
      for (int m = 0; m < TH; ++m) Hc[m] = Hn[m];
    }
    TICK(3);   // forward layers (MFMA) + staging
    // ---- output layer gradients
    store16<TH>(bufD, LY::STR_D, lane, Hc);
    WAVE_SYNC();
    outer16<TH, 4>(bufD, LY::STR_D, obuf, lane, dwo);
    if (q == 0) { dbo[0] += dout.x; dbo[1] += dout.y; dbo[2] += dout.z; dbo[3] += dout.w; }
    f32x4 dY[TH];
    {
      const float4* w4 = reinterpret_cast<const float4*>(sm + LY::WOUT);
#pragma unroll
      for (int m = 0; m < TH; ++m)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float4 w = w4[16 * m + 4 * q + r];
          const float dh = fmaf(w.w, dout.w, fmaf(w.z, dout.z, fmaf(w.y, dout.y, w.x * dout.x)));
          dY[m][r] = (Hc[m][r] > 0.f) ? dh : 0.f;
        }
    }
    TICK(4);   // output-layer grads (VALU)
#pragma unroll
    for (int l = L - 1; l >= 0; --l) {
      WAVE_SYNC();
      store16<TH>(bufD, LY::STR_D, lane, dY);
      WAVE_SYNC();
      dbh[l] += colsum16<TH>(bufD, LY::STR_D, lane);
      TICK(5);   // stage dY + bias column sums
      if (l == 0) {
        wgrad16<TH, TI>(bufD, LY::STR_D, wl + LY::x_off(0), LY::STR_E, lane, acc0);
        TICK(6); // wgrad MFMA
        if constexpr (HASH) {
          f32x4 dE[TI];
          dgrad16<TI, TH, BLK>(sm + LY::w_off(0), lane, dY, dE);
          if (valid) {
            const int64_t g = (int64_t)f * a.P + n, NP = (int64_t)a.F * a.P;
#pragma unroll
            for (int m = 0; m < TI; ++m)
#pragma unroll
              for (int p = 0; p < 2; ++p) {
                const int level = 8 * m + 2 * q + p;
                if (level < hc.nlev) a.hash_dE[level * NP + g] = make_float2(dE[m][2 * p], dE[m][2 * p + 1]);
              }
            if (q == 0) a.hash_xyz[g] = make_float4(x, y, z, 0.f);
          }
        }
        if (ENC_GRAD) {
          f32x4 dE[TI];
          dgrad16<TI, TH, BLK>(sm + LY::w_off(0), lane, dY, dE);
#pragma unroll
          for (int m = 0; m < TI; ++m)
#pragma unroll
            for (int r = 0; r < 4; ++r) dE[m][r] *= dEa[m][r];
          TICK(7); // dgrad MFMA (+cos scale)
          WAVE_SYNC();
          store16<TI>(bufD, LY::STR_D, lane, dE);
          WAVE_SYNC();
          outer16<TI, 3>(bufD, LY::STR_D, pbuf, lane, dwf);
          TICK(8); // Fourier matrix grads (VALU)
        }
      } else {
        wgrad16<TH, TH>(bufD, LY::STR_D, wl + LY::x_off(l), LY::STR_H, lane, accH[l - 1]);
        TICK(6);
        f32x4 dX[TH], Xl[TH];
        dgrad16<TH, TH, BLK>(sm + LY::w_off(l), lane, dY, dX);
        TICK(7);
        load16<TH>(wl + LY::x_off(l), LY::STR_H, lane, Xl);
#pragma unroll
        for (int m = 0; m < TH; ++m)
#pragma unroll
          for (int r = 0; r < 4; ++r) dY[m][r] = (Xl[m][r] > 0.f) ? dX[m][r] : 0.f;
        TICK(9); // relu mask (reload)
      }
    }
  }
  TICK(10);

  // ---- epilogue: the 8 waves' accumulators are summed in fixed wave order, all threads busy:
  // each round every wave parks EPI_CH of its C tiles in LDS ([wave][tile][reg][lane], conflict-free),
  // then the 512 threads each add the 8 copies of two elements and write the result to the workgroup's
  // partial vector in HBM.
  __syncthreads();
  float* stage = sm + LY::WTOTAL;
  int64_t enc_off, w_off[NGM_MAX_LAYERS + 1], b_off[NGM_MAX_LAYERS + 1];
  const int64_t ptot = ngm_param_offsets(&a.fc, &enc_off, w_off, b_off);
  (void)ptot;
  float* dst = a.partials + (int64_t)blockIdx.x * a.p_pad;
  const int D = a.fc.dim_enc, H = a.fc.dim_hidden;
  constexpr int NT0 = TH * TI, NTH = TH * TH, NT = NT0 + (L - 1) * NTH;
  constexpr int EPI_CH = 4, EPI_W = EPI_CH * 4 * 64;       // floats parked per wave and round
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
  TICK(11);  // epilogue
  TICK_REPORT
}

// ------------------------------------------------------------------------------------------------
template <int TI, int TH, int L>
static int launch_bwd16(const FieldBwdArgs& a, int blocks, hipStream_t st) {
  using LYH = Lds16<TI, TH, L>;
  constexpr int epi = LYH::WTOTAL + B16_WAVES * 4 * 4 * 64;    // epilogue staging (EPI_W per wave)
  const size_t lds = (size_t)(LYH::TOTAL > epi ? LYH::TOTAL : epi) * sizeof(float);
  if (lds > 160 * 1024) return NGM_E_UNSUPPORTED;
#define NGM_LB16(NC, EG, HS)                                                                                                \
  do {                                                                                                                      \
    (void)hipFuncSetAttribute((const void*)k_field_bwd16<TI, TH, L, NC, EG, HS>, hipFuncAttributeMaxDynamicSharedMemorySize, \
                              (int)lds);                                                                                    \
    hipLaunchKernelGGL((k_field_bwd16<TI, TH, L, NC, EG, HS>), dim3(blocks), dim3(B16_THREADS), lds, st, a);                \
  } while (0)
  if (a.fc.encoding == NGM_ENC_PERMUTO) {
    if constexpr (TI <= 2) NGM_LB16(false, false, true);
    else return NGM_E_UNSUPPORTED;
  } else if (a.fc.encoding == NGM_ENC_FOURIER) NGM_LB16(false, true, false);
  else if (a.fc.encoding == NGM_ENC_NERF) NGM_LB16(true, false, false);
  else NGM_LB16(false, false, false);
#undef NGM_LB16
  return 0;
}

// returns NGM_E_UNSUPPORTED when no 16-wide instantiation exists (caller falls back to k_field_bwd)
int ngm_launch_field_bwd16(const FieldBwdArgs& a, int blocks, hipStream_t st) {
  const int TI = (a.fc.dim_enc + 15) / 16, TH = (a.fc.dim_hidden + 15) / 16, L = a.fc.num_layers;
  const bool hash_ok = a.fc.encoding != NGM_ENC_PERMUTO || TI <= 2;
#ifdef NGM_FAST_BUILD
  const bool have = (TI == 4 && TH == 4 && L == 2);
#else
  const bool have = (TI == 4 && TH == 4 && (L == 1 || L == 2)) || (TI == 2 && TH == 2 && (L == 1 || L == 2)) ||
                    (TI == 3 && TH == 3 && L == 1);
#endif
  if (!have || !hash_ok) return NGM_E_UNSUPPORTED;
  NgmProfScope prof_(NGM_K_FIELD_BWD, st);
  if (TI == 4 && TH == 4 && L == 2) return launch_bwd16<4, 4, 2>(a, blocks, st);
#ifndef NGM_FAST_BUILD
  if (TI == 4 && TH == 4 && L == 1) return launch_bwd16<4, 4, 1>(a, blocks, st);
  if (TI == 2 && TH == 2 && L == 1) return launch_bwd16<2, 2, 1>(a, blocks, st);
  if (TI == 2 && TH == 2 && L == 2) return launch_bwd16<2, 2, 2>(a, blocks, st);
  if (TI == 3 && TH == 3 && L == 1) return launch_bwd16<3, 3, 1>(a, blocks, st);
#endif
  return NGM_E_UNSUPPORTED;
}
