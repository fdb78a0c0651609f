// Backward of encoding + MLP with 16-sample tiles on v_mfma_f32_16x16x4_f32 (gfx950).
//
// Same algorithm as k_field_bwd (ngm_field_bwd.hip) but tiled 16 samples wide so that a workgroup of
// EIGHT waves (two per SIMD) fits: per-wave staging buffers shrink to [16][F+12] floats and the
// activation registers halve, keeping every wave under 256 VGPRs.  The second wave per SIMD hides LDS and
// HBM latency.  It does NOT buy MFMA/VALU overlap: on gfx950 fp32 MFMAs and VALU instructions of the
// waves of one SIMD serialise (tools/micro/coexec.hip), so kernel time ~ sum of everything issued; the
// ablation in tools/ablate.sh shows forward recompute, wgrad, dgrad and the VALU phases adding linearly.
// This kernel serves point-mode backward and the shapes k_field_bwd16s does not cover.
//
// Fragment maps of v_mfma_f32_16x16x4_f32 (cdna_hip_programming.md section 3):
//   A: lane l holds A[i = l&15][k = l>>4]     B: lane l holds B[k = l>>4][j = l&15]
//   C/D: lane l, reg r holds C[row = 4*(l>>4) + r][col = l&15]
// Features on M, samples on N: lane (j = l&15, q = l>>4) holds feature 16*m + 4*q + r of sample j in
// register r of tile m -- exactly the B operand of k-step (m, r), so layers chain without data movement.
#include "ngm_bwd16.h"

// raw per-sample inputs of one tile column (ray table entry + distance, or the query point) and d_out.
// (An LDS-DMA variant of this prefetch, global_load_lds_dwordx4 into a dead staging tile, was measured
// 70% slower: hipcc then guards every later ds_read with vmcnt(0) and stops interleaving them.)
struct RawIn16 {
  float4 r0, r1, dout;
  float4 e[2];     // hash encoding of the sample from the forward's stash (HASH with a.act only)
  float t;
  bool valid;
};

// E_STASH: the lane's two 16-byte chunks (features 16m + 4q + {0..3}) of the stashed hash encoding,
// layout act[tile = g >> 5][chunk = feature >> 2][g & 31][4 floats] with 8 chunks per sample
template <bool E_STASH = false>
__device__ __forceinline__ RawIn16 fetch_inputs16(const FieldBwdArgs& a, int f, int64_t n, int64_t end, bool ray_mode,
                                                  uint32_t rays_per_field, int q = 0) {
  RawIn16 in;
  in.valid = n < end;
  in.r0 = in.r1 = in.dout = make_float4(0.f, 0.f, 0.f, 0.f);
  in.e[0] = in.e[1] = make_float4(0.f, 0.f, 0.f, 0.f);
  in.t = 0.f;
  if (in.valid) {
    const int64_t g = (int64_t)f * a.P + n;
    if constexpr (E_STASH) {
      if (a.act) {
        typedef float v4f __attribute__((ext_vector_type(4)));
        const v4f* tile = reinterpret_cast<const v4f*>(a.act) + (g >> 5) * (8 * 32);
        const int r = (int)(g & 31);                      // chunk c holds sample r at position r ^ (c & 7) (ActStash)
        const v4f e0 = __builtin_nontemporal_load(tile + q * 32 + (r ^ q)), e1 = __builtin_nontemporal_load(tile + (4 + q) * 32 + (r ^ (4 + q)));
        in.e[0] = make_float4(e0.x, e0.y, e0.z, e0.w);
        in.e[1] = make_float4(e1.x, e1.y, e1.z, e1.w);
      }
    }
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
  {
    FieldStage16<TI, TH, L> stage;      // every parameter load in flight at once, then the permuting LDS writes
    stage.issue(a.fc, a.pr, row);
    stage.commit(sm, a.fc);
  }
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
  // tile lists as in k_field_bwd16s: the SIMD's older wave (waves 0-3) runs ahead of its partner, so it takes 18/32
  // of the workgroup's tiles ([0, nA) round-robin over waves 0-3, [nA, T) over waves 4-7); static -> deterministic
  const int64_t T = (end - beg + 15) >> 4;
  int64_t nA = ((T * 18 + 31) / 32 + 3) & ~(int64_t)3;
  if (nA > T) nA = T;
  const bool older = wave < 4;
  const int64_t first = beg + 16 * (older ? (int64_t)wave : nA + (wave - 4));
  const int64_t lend = older ? min(end, beg + 16 * nA) : end;
  constexpr int TSTRIDE = 16 * (B16_WAVES / 2);
  RawIn16 nxt = fetch_inputs16<HASH && TI == 2>(a, f, first + j, lend, ray_mode, rays_per_field, q);
  for (int64_t base = first; base < lend; base += TSTRIDE) {
    // software pipeline: this tile's raw inputs were fetched one iteration ago; issue the next tile's
    // loads now so that their HBM latency hides behind this tile's arithmetic
    const RawIn16 cur = nxt;
    const int64_t n = base + j;
    nxt = fetch_inputs16<HASH && TI == 2>(a, f, n + TSTRIDE, lend, ray_mode, rays_per_field, q);
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
    if constexpr (HASH) {
      // ray mode after a training forward: the encoding comes from the forward's stash (no simplex search, no
      // table gathers here); point mode / no stash: recompute
      if (TI == 2 && a.act) {
#pragma unroll
        for (int m = 0; m < TI; ++m) E[m] = f32x4{cur.e[m & 1].x, cur.e[m & 1].y, cur.e[m & 1].z, cur.e[m & 1].w};
      } else encode_hash16<TI>(sm + LY::ENCW, hc, q, x, y, z, E);
    }
#ifdef NGM_ABL_NOVALU
    else {
#pragma unroll
      for (int m = 0; m < TI; ++m) { E[m] = f32x4{x, y, z, x}; dEa[m] = E[m]; }
    }
#else
    else encode16<TI, NEED_COS, ENC_GRAD, true>(sm + LY::ENCW, q, x, y, z, E, dEa);   // hardware sine / cosine, as the forward kernels
#endif
    TICK(2);   // encode (sincos)
    WAVE_SYNC();
    store16<TI>(wl + LY::x_off(0), LY::STR_E, lane, E);
    if (q == 0) {
      *reinterpret_cast<float4*>(pbuf + 4 * j) = make_float4(x, y, z, 0.f);
      *reinterpret_cast<float4*>(obuf + 4 * j) = dout;
    }
    f32x4 Hc[TH];
#ifdef NGM_ABL_NOFWD   // timing ablations (tools/ablate.sh): results are meaningless when any NGM_ABL_* is defined
#pragma unroll
    for (int m = 0; m < TH; ++m) Hc[m] = E[m % TI];
#else
    fwd16<TI, TH, BLK>(sm + LY::w_off(0), sm + LY::b_off(0), lane, E, Hc);
#endif
#pragma unroll
    for (int l = 1; l < L; ++l) {
      store16<TH>(wl + LY::x_off(l), LY::STR_H, lane, Hc);
      f32x4 Hn[TH];
#ifdef NGM_ABL_NOFWD
#pragma unroll
      for (int m = 0; m < TH; ++m) Hn[m] = Hc[m];
#else
      fwd16<TH, TH, BLK>(sm + LY::w_off(l), sm + LY::b_off(l), lane, Hc, Hn);
#endif
#pragma unroll
      for (int m = 0; m < TH; ++m) Hc[m] = Hn[m];
    }
    TICK(3);   // forward layers (MFMA) + staging
    // ---- output layer gradients
    store16<TH>(bufD, LY::STR_D, lane, Hc);
    WAVE_SYNC();
#ifndef NGM_ABL_NOVALU
    outer16<TH, 4>(bufD, LY::STR_D, obuf, lane, dwo);
#endif
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
#ifndef NGM_ABL_NOVALU
      dbh[l] += colsum16<TH>(bufD, LY::STR_D, lane);
#endif
      TICK(5);   // stage dY + bias column sums
      if (l == 0) {
#ifndef NGM_ABL_NOWGRAD
        wgrad16<TH, TI>(bufD, LY::STR_D, wl + LY::x_off(0), LY::STR_E, lane, acc0);
#endif
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
            if (q == 0 && !a.hash_xyz_ready) a.hash_xyz[g] = make_float4(x, y, z, 0.f);
          }
        }
        if (ENC_GRAD) {
          f32x4 dE[TI];
#ifdef NGM_ABL_NODGRAD
#pragma unroll
          for (int m = 0; m < TI; ++m) dE[m] = dY[m % TH];
#else
          dgrad16<TI, TH, BLK>(sm + LY::w_off(0), lane, dY, dE);
#endif
#pragma unroll
          for (int m = 0; m < TI; ++m)
#pragma unroll
            for (int r = 0; r < 4; ++r) dE[m][r] *= dEa[m][r];
          TICK(7); // dgrad MFMA (+cos scale)
          WAVE_SYNC();
          store16<TI>(bufD, LY::STR_D, lane, dE);
          WAVE_SYNC();
#ifndef NGM_ABL_NOVALU
          outer16<TI, 3>(bufD, LY::STR_D, pbuf, lane, dwf);
#endif
          TICK(8); // Fourier matrix grads (VALU)
        }
      } else {
#ifndef NGM_ABL_NOWGRAD
        wgrad16<TH, TH>(bufD, LY::STR_D, wl + LY::x_off(l), LY::STR_H, lane, accH[l - 1]);
#endif
        TICK(6);
        f32x4 dX[TH], Xl[TH];
#ifdef NGM_ABL_NODGRAD
#pragma unroll
        for (int m = 0; m < TH; ++m) dX[m] = dY[m];
#else
        dgrad16<TH, TH, BLK>(sm + LY::w_off(l), lane, dY, dX);
#endif
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

  __syncthreads();
  bwd16_epilogue<TI, TH, L, ENC_GRAD>(a, sm + LY::WTOTAL, acc0, accH, dbh, dwo, dwf, dbo);
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
  if (!have || !hash_ok || a.fc.skip_mode != NGM_SKIP_NO || a.fc.encoding == NGM_ENC_TRIPLANE) return NGM_E_UNSUPPORTED;   // skip connections: 32-sample-tile kernel
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
