// k_field_bwd_b3p: the backward of ngm_field_bwd_b3.hip with the two hidden layers of a tile on TWO waves.
//
// k_field_bwd_b3 keeps both 64x64 weight-gradient accumulators (128 registers) in one wave; what is left of the
// register file does not hold the operand look-ahead a fully software-pipelined tile needs (its Fourier instance
// spills the moment the encoding is moved under the data gradient's MFMAs).  Here a workgroup is two PAIRS of waves
// (still one wave per SIMD):
//   role A (even wave): output layer, layer 1's weight and data gradient, ReLU mask -> dY of layer 0's output,
//                       written in place over layer 1's input tile;
//   role B (odd wave) : one step behind on the same tiles: layer 0's data gradient with the encoding evaluated in
//                       the shadow of its MFMAs, Fourier-matrix gradient, layer 0's weight gradient.
// Each wave owns ONE accumulator set (64 registers), both roles issue 96 MFMAs per tile, and a workgroup barrier per
// step hands the tile over.  Buffers of a pair: the output layer's input tile x2 (DMA lands the next one), layer 1's
// input tile x3 (A works on tile s, B on s - 1, the DMA fills s + 1), inputs x2, points x2, d_out.
// Everything else -- layouts, splits, MFMA blocks, determinism (fixed tile lists, fixed summation order) -- is
// ngm_bwd_b3.h.  Compiled for two hidden layers; one hidden layer keeps k_field_bwd_b3.
#include "ngm_bwd_b3.h"

template <bool EG>
struct LdsB3p {
  static constexpr int PLANES = (1 + (EG ? 1 : 0)) * 3 * PLANE_G * 4;   // floats: W1 (+ W0 when the encoding has a gradient)
  static constexpr int plane_slot(int l) { return EG ? l : l - 1; }
  static constexpr int CONSTS = PLANES;                                  // float4 wout[64], enc[64]
  static constexpr int PAIRS = CONSTS + 512;
  // per pair
  static constexpr int HL = 0;                // 2 tiles
  static constexpr int H1 = 2 * HT;           // 3 tiles
  static constexpr int INB = 5 * HT;          // 2 x float4 [2 halves][4 pieces][16]
  static constexpr int PB = INB + 2 * 512;    // 2 x float4 [32]
  static constexpr int OB = PB + 2 * 128;     // float4 [32]
  static constexpr int PAIR_TOTAL = OB + 128;
  static constexpr int BODY = PAIRS + 2 * PAIR_TOTAL;
  static constexpr int EPI = 4 * 4 * 1024;
  static constexpr int TOTAL = BODY > EPI ? BODY : EPI;
};

// layer 0's data gradient with the encoding evaluated in the shadow of its MFMAs (k-blocks 0..2: 5 + 5 + 6 samples);
// sin -> Eb (k-block major: the weight gradient's operand), cos -> dEa (registers: role B has them)
template <int R0, int N, bool NEED_COS, bool EG>
__device__ __forceinline__ void encode_cols_p(const float4 (&encw)[2], const float* __restrict__ pbuf, int hi, float (&Eb)[2][2][8], float (&dEa)[2][16]) {
  const float inv2pi = 0.15915494309189535f;
  float4 pp[N];
#pragma unroll
  for (int e = 0; e < N; ++e) pp[e] = *reinterpret_cast<const float4*>(pbuf + 4 * (8 * ((R0 + e) >> 2) + 4 * hi + ((R0 + e) & 3)));
#pragma unroll
  for (int e = 0; e < N; ++e) {
    const int r = R0 + e;
    const float4 p = pp[e];
#pragma unroll
    for (int m = 0; m < 2; ++m) {
      const float4 w = encw[m];
      const float arg = fmaf(w.z, p.z, fmaf(w.y, p.y, w.x * p.x));
      const float rev = __builtin_amdgcn_fractf(arg * inv2pi);
      const float sn = __builtin_amdgcn_sinf(rev);
      float v = sn;
      if (NEED_COS) v = (w.w == NGM_FK_COS) ? __builtin_amdgcn_cosf(rev) : sn;
      if (m == 0) v = (w.w == NGM_FK_RAW) ? arg : v;            // raw coordinates are features 0..2
      Eb[r >> 3][m][r & 7] = v;
      if (EG) dEa[m][r] = __builtin_amdgcn_cosf(rev);           // Fourier only; raw rows carry no weight (slot never written)
    }
  }
}
// k-blocks 0..2; the last one is left to the caller (A3, Wb): the pair's transfers ride in the gaps of its MFMAs
template <bool NEED_COS>
__device__ __forceinline__ void dgrad_b3_enc(const ngm_u32x4* __restrict__ P, const RowRegs& R, const PlaneRegs& W0, int lane, f32x16 (&dX)[2],
                                             const float4 (&encw)[2], const float* __restrict__ pbuf, float (&Eb)[2][2][8], float (&dEa)[2][16],
                                             B3Op& A3, PlaneRegs& Wb) {
  PlaneRegs Wa;
  const int hi = lane >> 5;
  const B3Op A0 = b3_rows(R.g[0][0], R.g[0][1]);
  __builtin_amdgcn_sched_barrier(0);
  load_planes(P, 1, lane, Wb);
  const B3Op A1 = b3_rows(R.g[1][0], R.g[1][1]);
  encode_cols_p<0, 5, NEED_COS, true>(encw, pbuf, hi, Eb, dEa);
  dgrad_b3_kb_free(A0, W0, true, dX);
  NGM_INTERLEAVE(12, 9)
  __builtin_amdgcn_sched_barrier(0);
  load_planes(P, 2, lane, Wa);
  const B3Op A2 = b3_rows(R.g[2][0], R.g[2][1]);
  encode_cols_p<5, 5, NEED_COS, true>(encw, pbuf, hi, Eb, dEa);
  dgrad_b3_kb_free(A1, Wb, false, dX);
  NGM_INTERLEAVE(12, 9)
  __builtin_amdgcn_sched_barrier(0);
  load_planes(P, 3, lane, Wb);
  A3 = b3_rows(R.g[3][0], R.g[3][1]);
  encode_cols_p<10, 6, NEED_COS, true>(encw, pbuf, hi, Eb, dEa);
  dgrad_b3_kb_free(A2, Wa, false, dX);
  NGM_INTERLEAVE(12, 10)
  __builtin_amdgcn_sched_barrier(0);
}

// ------------------------------------------------------------------------------------------------
template <bool NEED_COS, bool ENC_GRAD>
__global__ __launch_bounds__(B3B_THREADS) void k_field_bwd_b3p(FieldBwdArgs a) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  using LY = LdsB3p<ENC_GRAD>;
  constexpr int L = 2;
  const int f = blockIdx.x % a.F, chunk = blockIdx.x / a.F;
  const int64_t row = a.pr.field_index ? a.pr.field_index[f] : f;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int pair = wave >> 1;
  const bool roleB = (wave & 1) != 0;
  const int i = lane & 31, hi = lane >> 5;
  ngm_u32x4* planes = reinterpret_cast<ngm_u32x4*>(sm);
  const ngm_u32x4* P1 = planes + LY::plane_slot(1) * 3 * PLANE_G;
  const ngm_u32x4* P0 = planes + LY::plane_slot(0) * 3 * PLANE_G;
  float* pl = sm + LY::PAIRS + pair * LY::PAIR_TOTAL;
  float* obuf = pl + LY::OB;
  const uint32_t pl_lds = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)pl);

  f32x16 acc[2][2];                    // role A: dW of layer 1, role B: dW of layer 0
#pragma unroll
  for (int mo = 0; mo < 2; ++mo)
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mo][mi][r] = 0.f;
  float dbh[L][2], dwo[2][4], dwf[2][3], dbo[4];
#pragma unroll
  for (int m = 0; m < 2; ++m) {
    dbh[0][m] = dbh[1][m] = 0.f;
#pragma unroll
    for (int c = 0; c < 4; ++c) dwo[m][c] = 0.f;
    dwf[m][0] = dwf[m][1] = dwf[m][2] = 0.f;
  }
  dbo[0] = dbo[1] = dbo[2] = dbo[3] = 0.f;

  const uint32_t beg = (uint32_t)chunk * (uint32_t)a.per_block, end = (uint32_t)min(a.P, (int64_t)beg + a.per_block);
  const uint32_t T = (end - beg + 31u) >> 5;                     // tiles of the workgroup; pair p takes p, p + 2, ...
  const uint32_t NP = (T > (uint32_t)pair) ? (T - (uint32_t)pair + 1u) / 2u : 0u;
  const uint32_t steps = (T + 1u) / 2u + 1u;                     // same for all four waves (barriers)
  auto tile_n0 = [&](uint32_t k) { return beg + 32u * ((uint32_t)pair + 2u * k); };
  FieldStreams fs;
  {
    const int64_t g0 = (int64_t)f * a.P;
    fs.raytab = reinterpret_cast<const char*>(a.raytab) + 32 * (g0 / a.S);
    fs.dout = reinterpret_cast<const char*>(a.d_out + g0);
    fs.tpair = reinterpret_cast<const char*>(a.stashB + (g0 & ~(int64_t)1));
    fs.par = (uint32_t)(g0 & 1);
    fs.gb = (uint32_t)(g0 & 31);
    fs.act[0] = reinterpret_cast<const char*>(a.act + (g0 >> 5) * 2048);
    fs.act[1] = reinterpret_cast<const char*>(a.act + a.act_layer_stride + (g0 >> 5) * 2048);
  }
  // first tile of the pair: role A fetches its inputs and the output layer's input tile, role B layer 1's input tile
  // (addresses clamp to the chunk: a pair without tiles fetches valid data it never uses)
  if (!roleB) {
    issue_inputs(fs, a.S, tile_n0(0), end, lane, pl_lds + LY::INB * 4);
    issue_inputs(fs, a.S, tile_n0(0) + 16, end, lane, pl_lds + LY::INB * 4 + 1024);
    issue_tile32(fs.act[1], fs.gb, tile_n0(0), end, lane, pl_lds + LY::HL * 4);
  } else {
    issue_tile32(fs.act[0], fs.gb, tile_n0(0), end, lane, pl_lds + LY::H1 * 4);
  }
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
  }
  if (ENC_GRAD) build_dgrad_planes(a.fc, a.pr, row, 0, planes + LY::plane_slot(0) * 3 * PLANE_G);
  build_dgrad_planes(a.fc, a.pr, row, 1, planes + LY::plane_slot(1) * 3 * PLANE_G);
  DMA_WAIT(0);
  __syncthreads();

  int col[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) col[j] = (i >> 2) * 128 + (i & 3) + 4 * ((4 * hi + j) ^ (i >> 2));
#define COL_OFF(m, r) ((m) * 1024 + col[(r) & 3] + 32 * ((r) >> 2))

#ifdef NGM_B3P_TIMING
  unsigned long long tw_ = 0; const unsigned long long ts_ = __builtin_readcyclecounter();
  unsigned long long tp_[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tl_ = 0;
#define BT(k) do { const unsigned long long n_ = __builtin_readcyclecounter(); tp_[k] += n_ - tl_; tl_ = n_; } while (0)
#define BT0() tl_ = __builtin_readcyclecounter()
#else
#define BT(k)
#define BT0()
#endif
  for (uint32_t s = 0; s < steps; ++s) {
#ifdef NGM_B3P_TIMING
    const unsigned long long t0_ = __builtin_readcyclecounter();
#endif
    if (!roleB) {
      // =========================== role A: tile s of the pair ===========================
      const uint32_t base = tile_n0(s);
      float* HLb = pl + LY::HL + (s & 1u) * HT;
      float* H1b = pl + LY::H1 + (s % 3u) * HT;
      float* inb = pl + LY::INB + (s & 1u) * 512;
      float* pbuf = pl + LY::PB + (s & 1u) * 128;
      if (s < NP) {
        // ---- inputs: lane = sample (both halves compute, half 0 stores)
        {
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
        // ---- output layer, lane = feature
        f32x16 dY[2], Xc[2];
        {
          f32x16 Hc[2];
          const float4 wout[2] = {cwout[i], cwout[32 + i]};
#pragma unroll
          for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int r = 0; r < 16; ++r) Hc[m][r] = HLb[COL_OFF(m, r)];
#pragma unroll
          for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int r = 0; r < 16; ++r) Xc[m][r] = H1b[COL_OFF(m, r)];      // layer 1's input columns: in flight under the arithmetic
#pragma unroll
          for (int half = 0; half < 2; ++half) {
            float4 dO[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) dO[e] = *reinterpret_cast<const float4*>(obuf + 4 * (8 * ((8 * half + e) >> 2) + 4 * hi + (e & 3)));
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
              const int r = 8 * half + e;
              const float4 d = dO[e];
#pragma unroll
              for (int m = 0; m < 2; ++m) {
                const float h = Hc[m][r];
                const float dh = fmaf(wout[m].w, d.w, fmaf(wout[m].z, d.z, fmaf(wout[m].y, d.y, wout[m].x * d.x)));
                const float g = (h > 0.f) ? dh : 0.f;
                dY[m][r] = g;
                dbh[1][m] += g;
                dwo[m][0] = fmaf(d.x, h, dwo[m][0]); dwo[m][1] = fmaf(d.y, h, dwo[m][1]);
                dwo[m][2] = fmaf(d.z, h, dwo[m][2]); dwo[m][3] = fmaf(d.w, h, dwo[m][3]);
              }
            }
            __builtin_amdgcn_sched_barrier(0);
          }
          WAVE_SYNC();
#pragma unroll
          for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int r = 0; r < 16; ++r) HLb[COL_OFF(m, r)] = dY[m][r];
          WAVE_SYNC();
        }
        // ---- layer 1: weight gradient (block 1's splits under block 0's MFMAs), data gradient, mask
        RowRegs R;
        PlaneRegs W0;
        {
          B3Op A0[2] = {b3_regs<0>(dY[0]), b3_regs<0>(dY[1])}, B0[2] = {b3_regs<0>(Xc[0]), b3_regs<0>(Xc[1])};
          __builtin_amdgcn_sched_barrier(0);
          load_rows(HLb, lane, R);
          load_planes(P1, 0, lane, W0);
          B3Op A1[2] = {b3_regs<1>(dY[0]), b3_regs<1>(dY[1])}, B1[2] = {b3_regs<1>(Xc[0]), b3_regs<1>(Xc[1])};
          wgrad_b3_block_free(A0, B0, acc);
          NGM_INTERLEAVE(24, 8)
          __builtin_amdgcn_sched_barrier(0);
          wgrad_b3_block(A1, B1, acc);
        }
        __builtin_amdgcn_sched_barrier(0);
        f32x16 dX[2];
        dgrad_b3(P1, R, W0, lane, dX);
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const float g = (Xc[m][r] > 0.f) ? dX[m][r] : 0.f;
            dY[m][r] = g;
            dbh[0][m] += g;
          }
        WAVE_SYNC();
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
          for (int r = 0; r < 16; ++r) H1b[COL_OFF(m, r)] = dY[m][r];   // role B reads it after the barrier
        WAVE_SYNC();
      }
    } else {
      // =========================== role B: tile s - 1 of the pair ===========================
      const bool has_tile = (s >= 1 && s - 1 < NP);
      // All transfers of the pair for tile s + 1 are issued by role B, the lighter half of a step: layer 1's input tile (its
      // buffer held tile s - 2, which this wave finished a step ago), role A's output-layer tile and inputs (last read at
      // step s - 1); whole tiles aligned with the stash tiles take the scalar-addressed form.  WHERE matters: a wave's LDS
      // instructions crawl while it has transfers it has not waited for (tools/micro/dma_lds.hip; here: the LDS-heavy data
      // gradient right behind 18 transfers took 2 600 clocks instead of 1 500), so they go out in front of the weight
      // gradient -- 48 MFMAs and their operand splits, no LDS instruction until the wait.
      if (!has_tile) {
      {
        const uint32_t nx = tile_n0(s + 1), u0 = nx + fs.gb;
        issue_inputs(fs, a.S, nx, end, lane, pl_lds + (LY::INB + ((s + 1) & 1u) * 512) * 4);
        issue_inputs(fs, a.S, nx + 16, end, lane, pl_lds + (LY::INB + ((s + 1) & 1u) * 512) * 4 + 1024);
        if (((u0 & 31u) == 0u) && (nx + 32u <= end)) {
          issue_tile32_fast(fs.act[0], u0 >> 5, lane, pl_lds + (LY::H1 + ((s + 1) % 3u) * HT) * 4);
          issue_tile32_fast(fs.act[1], u0 >> 5, lane, pl_lds + (LY::HL + ((s + 1) & 1u) * HT) * 4);
        } else {
          issue_tile32(fs.act[0], fs.gb, nx, end, lane, pl_lds + (LY::H1 + ((s + 1) % 3u) * HT) * 4);
          issue_tile32(fs.act[1], fs.gb, nx, end, lane, pl_lds + (LY::HL + ((s + 1) & 1u) * HT) * 4);
        }
      }
      }
      if (has_tile) {
        const uint32_t k = s - 1;
        float* Dtile = pl + LY::H1 + (k % 3u) * HT;
        const float* pbuf = pl + LY::PB + (k & 1u) * 128;
        float dY[2][16];
        BT0();
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
          for (int r = 0; r < 16; ++r) dY[m][r] = Dtile[COL_OFF(m, r)];
        const float4 encw[2] = {cenc[i], cenc[32 + i]};
        float Eb[2][2][8], dEa[2][16];
        f32x16 dE[2];
        if constexpr (ENC_GRAD) {
          RowRegs R0;
          PlaneRegs W00;
          load_rows(Dtile, lane, R0);
          load_planes(P0, 0, lane, W00);
          __builtin_amdgcn_sched_barrier(0);
          B3Op A3;
          PlaneRegs W3;
          BT(0);
          dgrad_b3_enc<NEED_COS>(P0, R0, W00, lane, dE, encw, pbuf, Eb, dEa, A3, W3);
          BT(1);
          dgrad_b3_kb(A3, W3, false, dE);
          __builtin_amdgcn_sched_barrier(0);
          BT(2);
          // Fourier-matrix gradient: d sin(w.x)/d w = cos(w.x) x
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const float4 p = *reinterpret_cast<const float4*>(pbuf + 4 * (8 * (r >> 2) + 4 * hi + (r & 3)));
#pragma unroll
            for (int m = 0; m < 2; ++m) {
              const float g = dE[m][r] * dEa[m][r];
              dwf[m][0] = fmaf(g, p.x, dwf[m][0]); dwf[m][1] = fmaf(g, p.y, dwf[m][1]); dwf[m][2] = fmaf(g, p.z, dwf[m][2]);
            }
          }
        } else {
          encode_cols_p<0, 8, NEED_COS, false>(encw, pbuf, hi, Eb, dEa);
          encode_cols_p<8, 8, NEED_COS, false>(encw, pbuf, hi, Eb, dEa);
        }
        __builtin_amdgcn_sched_barrier(0);
        BT(3);
        WAVE_SYNC();
      {
        const uint32_t nx = tile_n0(s + 1), u0 = nx + fs.gb;
        issue_inputs(fs, a.S, nx, end, lane, pl_lds + (LY::INB + ((s + 1) & 1u) * 512) * 4);
        issue_inputs(fs, a.S, nx + 16, end, lane, pl_lds + (LY::INB + ((s + 1) & 1u) * 512) * 4 + 1024);
        if (((u0 & 31u) == 0u) && (nx + 32u <= end)) {
          issue_tile32_fast(fs.act[0], u0 >> 5, lane, pl_lds + (LY::H1 + ((s + 1) % 3u) * HT) * 4);
          issue_tile32_fast(fs.act[1], u0 >> 5, lane, pl_lds + (LY::HL + ((s + 1) & 1u) * HT) * 4);
        } else {
          issue_tile32(fs.act[0], fs.gb, nx, end, lane, pl_lds + (LY::H1 + ((s + 1) % 3u) * HT) * 4);
          issue_tile32(fs.act[1], fs.gb, nx, end, lane, pl_lds + (LY::HL + ((s + 1) & 1u) * HT) * 4);
        }
      }
        __builtin_amdgcn_sched_barrier(0);
        {
          B3Op A0[2] = {b3_regs<0>(dY[0]), b3_regs<0>(dY[1])}, B0[2] = {b3_arr(Eb[0][0]), b3_arr(Eb[0][1])};
          __builtin_amdgcn_sched_barrier(0);
          BT(4);
          B3Op A1[2] = {b3_regs<1>(dY[0]), b3_regs<1>(dY[1])}, B1[2] = {b3_arr(Eb[1][0]), b3_arr(Eb[1][1])};
          wgrad_b3_block_free(A0, B0, acc);
          NGM_INTERLEAVE(24, 8)
          __builtin_amdgcn_sched_barrier(0);
          BT(5);
          wgrad_b3_block(A1, B1, acc);
        }
        __builtin_amdgcn_sched_barrier(0);
        BT(6);
      }
      DMA_WAIT(0);                   // own transfers: role A reads them after the barrier
    }
#ifdef NGM_B3P_TIMING
    tw_ += __builtin_readcyclecounter() - t0_;
#endif
    __syncthreads();
  }
#ifdef NGM_B3P_TIMING
  if (a.debug_cycles && blockIdx.x == gridDim.x / 2 && lane == 0) {
    a.debug_cycles[wave] = tw_; a.debug_cycles[4 + wave] = __builtin_readcyclecounter() - ts_;
    if (wave == 1) for (int k = 0; k < 7; ++k) a.debug_cycles[8 + k] = tp_[k];
  }
#endif
#undef COL_OFF

  // ---- epilogue: the two waves of a role are summed in fixed order (all of LDS is free now)
  float* stage = sm;
#pragma unroll
  for (int mo = 0; mo < 2; ++mo)
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const f32x16& c = acc[mo][mi];
        *reinterpret_cast<float4*>(stage + (((wave * 4 + (mo * 2 + mi)) * 4 + q) * 64 + lane) * 4) =
            make_float4(c[4 * q], c[4 * q + 1], c[4 * q + 2], c[4 * q + 3]);
      }
  __syncthreads();
  int64_t enc_off, w_off[NGM_MAX_LAYERS + 1], b_off[NGM_MAX_LAYERS + 1];
  (void)ngm_param_offsets(&a.fc, &enc_off, w_off, b_off);
  float* dst = a.partials + (int64_t)blockIdx.x * a.p_pad;
  const int D = a.fc.dim_enc, H = a.fc.dim_hidden;
  for (int e4 = threadIdx.x; e4 < 2 * 4 * 256; e4 += B3B_THREADS) {      // [layer][tile][q][lane]
    const int l = e4 >> 10, rest = e4 & 1023;
    const int w0 = (l == 1) ? 0 : 1;                                     // role A waves (0, 2) hold layer 1, role B (1, 3) layer 0
    const float4 u = *reinterpret_cast<const float4*>(stage + (w0 * 4 * 256 + rest) * 4);
    const float4 v = *reinterpret_cast<const float4*>(stage + ((w0 + 2) * 4 * 256 + rest) * 4);
    const float4 s4 = make_float4(u.x + v.x, u.y + v.y, u.z + v.z, u.w + v.w);
    const int t = rest >> 8, q = (rest >> 6) & 3, ln = rest & 63;
    const int mo = (t >> 1) & 1, mi = t & 1;
    const int o0 = 32 * mo + 8 * q + 4 * (ln >> 5), c = 32 * mi + (ln & 31), din = (l == 0) ? D : H;
    if (c < din) {
      float* d = dst + w_off[l] + (int64_t)o0 * din + c;
      if (o0 < H) d[0] = s4.x;
      if (o0 + 1 < H) d[din] = s4.y;
      if (o0 + 2 < H) d[2 * din] = s4.z;
      if (o0 + 3 < H) d[3 * din] = s4.w;
    }
  }
  __syncthreads();
  // per-feature vectors (a role that does not own one contributes zeros)
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
}

// ------------------------------------------------------------------------------------------------
// returns NGM_E_UNSUPPORTED when this variant does not apply (caller falls back to k_field_bwd_b3 / the fp32-MFMA kernels)
int ngm_launch_field_bwd_b3p(const FieldBwdArgs& a, int blocks, hipStream_t st) {
  const int MI = (a.fc.dim_enc + 31) / 32, MH = (a.fc.dim_hidden + 31) / 32, L = a.fc.num_layers;
  if (!a.act || a.points || a.fc.skip_mode != NGM_SKIP_NO || a.fc.matmul_mode == NGM_MATMUL_F32 || MI != 2 || MH != 2 || L != 2)
    return NGM_E_UNSUPPORTED;
  if (a.fc.encoding != NGM_ENC_FOURIER && a.fc.encoding != NGM_ENC_NERF && a.fc.encoding != NGM_ENC_NONE) return NGM_E_UNSUPPORTED;
  if ((a.P + 64) * 256 >= ((int64_t)1 << 32)) return NGM_E_UNSUPPORTED;   // 32-bit byte offsets inside a field
  NgmProfScope prof_(NGM_K_FIELD_BWD, st);
#define NGM_LBB3P(NC, EG)                                                                                             \
  do {                                                                                                                \
    const size_t lds = (size_t)LdsB3p<EG>::TOTAL * sizeof(float);                                                     \
    if (lds > 160 * 1024) return NGM_E_UNSUPPORTED;                                                                   \
    (void)hipFuncSetAttribute((const void*)k_field_bwd_b3p<NC, EG>, hipFuncAttributeMaxDynamicSharedMemorySize,       \
                              (int)lds);                                                                              \
    hipLaunchKernelGGL((k_field_bwd_b3p<NC, EG>), dim3(blocks), dim3(B3B_THREADS), lds, st, a);                       \
  } while (0)
  if (a.fc.encoding == NGM_ENC_FOURIER) NGM_LBB3P(false, true);
  else if (a.fc.encoding == NGM_ENC_NERF) NGM_LBB3P(true, false);
  else NGM_LBB3P(false, false);
#undef NGM_LBB3P
  return 0;
}
