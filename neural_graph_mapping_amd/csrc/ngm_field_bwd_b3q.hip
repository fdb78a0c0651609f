// k_field_bwd_b3q: the backward of ngm_field_bwd_b3.hip with a tile's two hidden layers on a PAIR of waves THAT SHARE A
// SIMD -- eight waves per workgroup, two per SIMD.
//
// Why (tools/micro/coexec2.hip, round 3): on gfx950 the v_mfma_f32_32x32x16_bf16 stream of one wave is NOT slowed by the
// split-type VALU stream of the other wave of its SIMD (8 MFMAs: 256 clocks alone, 256 with the partner splitting beside it;
// the partner still gets 44 VALU instructions through per 256 clocks), and two VALU-only waves issue at twice the rate
// one wave reaches alone (144 clocks per 44-instruction split each, alone or together).  k_field_bwd_b3 runs ONE wave per
// SIMD (its two 64x64 weight-gradient accumulators take 128 registers and its look-ahead another 250): its phase clocks show
// the matrix pipe idle for half of every tile, under VALU / LDS-only phases nothing else is resident to fill, and the GEMM
// phases at 48 clocks per MFMA instead of 32.  Here
//   role A (waves 0-3): output layer, layer 1's data gradient, ReLU mask -> dY of layer 0's output written in place over
//                       layer 1's input tile, layer 1's weight gradient (64 accumulator registers);
//   role B (waves 4-7): one step behind on the same tiles: layer 0's data gradient, the encoding with its Fourier-matrix
//                       gradient, layer 0's weight gradient (64 accumulator registers);
// pair p = waves (p, p + 4) = the two waves of SIMD p.  Both roles issue 96 MFMAs and ~1 000 VALU instructions per tile, so
// whenever one is in a VALU / LDS phase the other's MFMAs have the pipe, and each hides the other's latencies: nothing is
// software-pipelined, no look-ahead registers, no scheduling directives.
// Per pair in LDS: the output layer's input tile, layer 1's input tile x2 (A works on tile s, B on s - 1), positions x3,
// d_out rows; 26 KB x 4 pairs + 48 KB of data-gradient weight planes = 154 KB.  A workgroup barrier per step hands the
// tiles over.  Every role does its LDS work first, issues the next tile's HBM -> LDS transfers where its buffers are
// free, and ends the step with its LDS-free weight gradient (48 MFMAs) -- a wave's LDS instructions crawl while it has
// transfers it has not waited for (tools/micro/dma_lds.hip).
// Positions and d_out rows come straight from k_stash_bwd's outputs (no lane = sample phase, no ray-table traffic).
// Deterministic: fixed tile lists, fixed summation order in the epilogue.  Arithmetic, layouts, splits, MFMA blocks:
// ngm_bwd_b3.h, shared with k_field_bwd_b3, which remains for one hidden layer and as the reference for this kernel.
#include "ngm_bwd_b3.h"

#define B3Q_WAVES 8
#define B3Q_THREADS 512
#define B3Q_PAIRS 4

template <bool EG>
struct LdsB3q {
  static constexpr int PLANES = (1 + (EG ? 1 : 0)) * 3 * PLANE_G * 4;   // floats: W1 (+ W0 when the encoding has a gradient)
  static constexpr int plane_slot(int l) { return EG ? l : l - 1; }
  static constexpr int CONSTS = PLANES;                                  // float4 wout[64], enc[64]
  static constexpr int PAIRS = CONSTS + 512;
  static constexpr int HL = 0;                // the output layer's input tile
  static constexpr int H1 = HT;               // layer 1's input tile x2
  static constexpr int PB = 3 * HT;           // positions x3: float4 [32]
  static constexpr int OB = PB + 3 * 128;     // d_out rows: float4 [32]
  static constexpr int PAIR_TOTAL = OB + 128;
  static constexpr int BODY = PAIRS + B3Q_PAIRS * PAIR_TOTAL;
  static constexpr int EPI = B3Q_WAVES * 4 * 1024;
  static constexpr int TOTAL = BODY > EPI ? BODY : EPI;
};

// dW[32 mo + .][.] += the 16 samples of one k-block: A = dY of output rows mo (split by the caller just before), B = both
// input tiles.  12 MFMAs on two accumulators, product-major.
__device__ __forceinline__ void wgrad_b3_half(const B3Op& A, const B3Op (&Bx)[2], f32x16 (&acc)[2]) {
#define NGM_WGH(PA, PB_) _Pragma("unroll") for (int mi = 0; mi < 2; ++mi) acc[mi] = mfma_bf16(A.PA, Bx[mi].PB_, acc[mi])
  NGM_WGH(l, h); NGM_WGH(h, l); NGM_WGH(m, m); NGM_WGH(m, h); NGM_WGH(h, m); NGM_WGH(h, h);
#undef NGM_WGH
}

struct RowK { float4 g0, g1; };
__device__ __forceinline__ void load_row_k(const float* __restrict__ tile, int kb, int rn, int rkh, RowK& R) {
  const int c0 = 4 * kb + 2 * rkh;
  R.g0 = *reinterpret_cast<const float4*>(tile + tile_chunk(c0, rn));
  R.g1 = *reinterpret_cast<const float4*>(tile + tile_chunk(c0 + 1, rn));
}
// dX^T = dY^T W over the four k-blocks: rows of dY from the LDS tile, weight planes from LDS; k-block kb + 1's 8 reads are
// issued before k-block kb's split + 12 MFMAs (one block of look-ahead: 32 registers)
__device__ __forceinline__ void dgrad_b3q(const ngm_u32x4* __restrict__ P, const float* __restrict__ tile, int lane, int rn, int rkh,
                                          f32x16 (&dX)[2]) {
  RowK Ra, Rb;
  PlaneRegs Wa, Wb;
  load_row_k(tile, 0, rn, rkh, Ra); load_planes(P, 0, lane, Wa);
  load_row_k(tile, 1, rn, rkh, Rb); load_planes(P, 1, lane, Wb);
  __builtin_amdgcn_sched_barrier(0);
  { const B3Op A = b3_rows(Ra.g0, Ra.g1); dgrad_b3_kb_free(A, Wa, true, dX); }
  load_row_k(tile, 2, rn, rkh, Ra); load_planes(P, 2, lane, Wa);
  __builtin_amdgcn_sched_barrier(0);
  { const B3Op A = b3_rows(Rb.g0, Rb.g1); dgrad_b3_kb_free(A, Wb, false, dX); }
  load_row_k(tile, 3, rn, rkh, Rb); load_planes(P, 3, lane, Wb);
  __builtin_amdgcn_sched_barrier(0);
  { const B3Op A = b3_rows(Ra.g0, Ra.g1); dgrad_b3_kb_free(A, Wa, false, dX); }
  __builtin_amdgcn_sched_barrier(0);
  { const B3Op A = b3_rows(Rb.g0, Rb.g1); dgrad_b3_kb_free(A, Wb, false, dX); }
}

// the same without look-ahead (role A: its registers are taken; the partner wave hides the LDS latency)
__device__ __forceinline__ void dgrad_b3q_lean(const ngm_u32x4* __restrict__ P, const float* __restrict__ tile, int lane, int rn, int rkh,
                                               f32x16 (&dX)[2]) {
#pragma unroll
  for (int kb = 0; kb < 4; ++kb) {
    RowK R;
    PlaneRegs Wk;
    load_row_k(tile, kb, rn, rkh, R);
    load_planes(P, kb, lane, Wk);
    const B3Op A = b3_rows(R.g0, R.g1);
    dgrad_b3_kb_free(A, Wk, kb == 0, dX);
    __builtin_amdgcn_sched_barrier(0);
  }
}

template <bool NEED_COS, bool ENC_GRAD>
__global__ __launch_bounds__(B3Q_THREADS) void k_field_bwd_b3q(FieldBwdArgs a) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  using LY = LdsB3q<ENC_GRAD>;
  constexpr int L = 2;
  const int f = blockIdx.x % a.F, chunk = blockIdx.x / a.F;
  const int64_t row = a.pr.field_index ? a.pr.field_index[f] : f;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int pair = wave & 3;
  const bool roleB = wave >= 4;
  const int i = lane & 31, hi = lane >> 5;
  ngm_u32x4* planes = reinterpret_cast<ngm_u32x4*>(sm);
  const ngm_u32x4* P1 = planes + LY::plane_slot(1) * 3 * PLANE_G;
  const ngm_u32x4* P0 = planes + LY::plane_slot(0) * 3 * PLANE_G;
  float* pl = sm + LY::PAIRS + pair * LY::PAIR_TOTAL;
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
  const uint32_t T = (end - beg + 31u) >> 5;                     // tiles of the workgroup; pair p takes p, p + 4, ...
  const uint32_t NP = (T > (uint32_t)pair) ? (T - (uint32_t)pair + 3u) / 4u : 0u;
  const uint32_t steps = (T + 3u) / 4u + 1u;                     // same for all eight waves (barriers)
  auto tile_n0 = [&](uint32_t k) { return beg + 32u * ((uint32_t)pair + 4u * k); };
  const int64_t g0 = (int64_t)f * a.P;
  const uint32_t gb = (uint32_t)(g0 & 31);
  const char* act0 = reinterpret_cast<const char*>(a.act + (g0 >> 5) * 2048);
  const char* act1 = reinterpret_cast<const char*>(a.act + a.act_layer_stride + (g0 >> 5) * 2048);
  const char* dout = reinterpret_cast<const char*>(a.d_out + g0);
  const char* xyz = reinterpret_cast<const char*>(a.hash_xyz + g0);
  // 32 rows of 16 bytes (d_out / positions of one tile, clamped to the chunk): the lower half-wave moves them
  auto issue_rows = [&](const char* src, uint32_t n0, uint32_t lds, int lane) __attribute__((always_inline)) {
    if (lane < 32) {
      uint32_t n = n0 + (uint32_t)lane;
      if (n >= end) n = end - 1;
      dma16(src + 16 * (size_t)n, lds);
    }
  };
  auto issue_tile = [&](const char* src, uint32_t n0, uint32_t lds, int lane) __attribute__((always_inline)) {
    const uint32_t u0 = n0 + gb;
    if (((u0 & 31u) == 0u) && (n0 + 32u <= end)) issue_tile32_fast(src, u0 >> 5, lane, lds);   // whole, aligned: linear copies
    else issue_tile32(src, gb, n0, end, lane, lds);
  };
  // first tile of the pair: role A fetches the output layer's input tile and d_out, role B layer 1's input tile and the
  // positions (addresses clamp to the chunk: a pair without tiles fetches valid data it never uses)
  if (!roleB) {
    issue_tile(act1, tile_n0(0), pl_lds + LY::HL * 4, lane);
    issue_rows(dout, tile_n0(0), pl_lds + LY::OB * 4, lane);
  } else {
    issue_tile(act0, tile_n0(0), pl_lds + LY::H1 * 4, lane);
    issue_rows(xyz, tile_n0(0), pl_lds + LY::PB * 4, lane);
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
  if (threadIdx.x < B3B_THREADS) {     // build_dgrad_planes strides by B3B_THREADS
    if (ENC_GRAD) build_dgrad_planes(a.fc, a.pr, row, 0, planes + LY::plane_slot(0) * 3 * PLANE_G);
    build_dgrad_planes(a.fc, a.pr, row, 1, planes + LY::plane_slot(1) * 3 * PLANE_G);
  }
  DMA_WAIT(0);
  __syncthreads();

#define COL_OFF(m, r) ((m) * 1024 + col[(r) & 3] + 32 * ((r) >> 2))
#ifdef NGM_B3Q_PRIO   // experiment: role A is the longer role (role B idles at the step barrier): let it win VALU arbitration
  if (!roleB) __builtin_amdgcn_s_setprio(NGM_B3Q_PRIO);
#endif
  const int lane0 = lane;
#ifdef NGM_B3Q_TIMING   // per-wave clocks: slots 0..5 phases of the role, 6 transfer wait, 7 barrier wait, 8 total (waves 0 and 4 of the middle block)
  unsigned long long tq_[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}, tl_ = __builtin_readcyclecounter();
  const unsigned long long ts_ = tl_;
#define QT(k) do { const unsigned long long n_ = __builtin_readcyclecounter(); tq_[k] += n_ - tl_; tl_ = n_; } while (0)
#else
#define QT(k)
#endif
  for (uint32_t s = 0; s < steps; ++s) {
    // Every lane-dependent LDS / HBM offset is re-derived per step from an opaque copy of the lane index (a dozen VALU):
    // hoisted out of this loop as invariants they occupied ~60 registers and were spilled to scratch, and a scratch
    // reload waits on vmcnt, i.e. on the transfers in flight.
    int lane = lane0;
    asm volatile("" : "+v"(lane));
    const int i = lane & 31, hi = lane >> 5;
    int col[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) col[j] = (i >> 2) * 128 + (i & 3) + 4 * ((4 * hi + j) ^ (i >> 2));
    const int rn = i, rkh = hi;                                     // row reads: lane = sample rn, k-half rkh
    if (!roleB) {
      // =========================== role A: tile s of the pair ===========================
      if (s < NP) {
        const uint32_t base = tile_n0(s);
        float* HLb = pl + LY::HL;
        float* H1b = pl + LY::H1 + (s & 1u) * HT;
        float* obuf = pl + LY::OB;
        if (base + 32u > end) {          // last, partial tile of the chunk: rows past the end carry no gradient
          if (lane < 32 && base + (uint32_t)lane >= end) {
            float z = 0.f;
            asm volatile("" : "+v"(z));
            *reinterpret_cast<float4*>(obuf + 4 * lane) = make_float4(z, z, z, z);
          }
          WAVE_SYNC();
        }
        QT(0);
        // ---- output layer, lane = feature: dY = relu'(H) * (Wout^T d_out); output-weight and bias gradients
        f32x16 dY[2];
        {
          f32x16 Hc[2];
          const float4 wout[2] = {cwout[i], cwout[32 + i]};
#pragma unroll
          for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int r = 0; r < 16; ++r) Hc[m][r] = HLb[COL_OFF(m, r)];
          if (hi == 0) {
            const float4 dd = *reinterpret_cast<const float4*>(obuf + 4 * i);
            dbo[0] += dd.x; dbo[1] += dd.y; dbo[2] += dd.z; dbo[3] += dd.w;
          }
#pragma unroll
          for (int half = 0; half < 2; ++half) {
            float4 dO[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) dO[e] = *reinterpret_cast<const float4*>(obuf + 4 * (8 * ((8 * half + e) >> 2) + 4 * hi + (e & 3)));
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
          // pin the per-feature sums HERE: left free, the compiler sinks their FMAs to the end of the step and keeps (spills)
          // the d_out rows and activations they read
#pragma unroll
          for (int m = 0; m < 2; ++m) {
            asm volatile("" : "+v"(dbh[1][m]));
#pragma unroll
            for (int c = 0; c < 4; ++c) asm volatile("" : "+v"(dwo[m][c]));
          }
          WAVE_SYNC();
#pragma unroll
          for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int r = 0; r < 16; ++r) HLb[COL_OFF(m, r)] = dY[m][r];
          WAVE_SYNC();
        }
        QT(1);
        // ---- layer 1's weight gradient first: dY and the layer's input columns are in registers only here
        {
          f32x16 Xc[2];
#pragma unroll
          for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int r = 0; r < 16; ++r) Xc[m][r] = H1b[COL_OFF(m, r)];
#pragma unroll
          for (int b = 0; b < 2; ++b) {
            B3Op Bx[2];
            if (b == 0) { Bx[0] = b3_regs<0>(Xc[0]); Bx[1] = b3_regs<0>(Xc[1]); }
            else { Bx[0] = b3_regs<1>(Xc[0]); Bx[1] = b3_regs<1>(Xc[1]); }
#pragma unroll
            for (int mo = 0; mo < 2; ++mo) {         // one output-row half at a time: 36 operand registers instead of 48
              const B3Op A = (b == 0) ? b3_regs<0>(dY[mo]) : b3_regs<1>(dY[mo]);
              wgrad_b3_half(A, Bx, acc[mo]);
              __builtin_amdgcn_sched_barrier(0);
            }
          }
        }
        QT(2);
        // ---- layer 1: data gradient dX^T = dY^T W1 (rows of dY from the tile, k-block by k-block)
        f32x16 dX[2];
        dgrad_b3q_lean(P1, HLb, lane, rn, rkh, dX);
        QT(3);
        // ---- ReLU mask of layer 0's output (re-read: cheaper than 32 registers held across two GEMMs), dY in place
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const float x = H1b[COL_OFF(m, r)];
            const float g = (x > 0.f) ? dX[m][r] : 0.f;
            dX[m][r] = g;
            dbh[0][m] += g;
          }
        asm volatile("" : "+v"(dbh[0][0]), "+v"(dbh[0][1]));
        WAVE_SYNC();
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
          for (int r = 0; r < 16; ++r) H1b[COL_OFF(m, r)] = dX[m][r];      // role B reads it after the barrier
        WAVE_SYNC();
        // ---- the next tile's transfers LAST: nothing of this role may follow them but the wait (an LDS instruction or a
        // scratch reload behind un-waited transfers waits for them); the partner role has the SIMD meanwhile
        if (s + 1 < NP) {
          issue_tile(act1, tile_n0(s + 1), pl_lds + LY::HL * 4, lane);
          issue_rows(dout, tile_n0(s + 1), pl_lds + LY::OB * 4, lane);
        }
        QT(4);
      }
    } else {
      // =========================== role B: tile s - 1 of the pair ===========================
      const bool has_tile = (s >= 1 && s - 1 < NP);
      if (has_tile) {
        const uint32_t k = s - 1;
        const float* Dtile = pl + LY::H1 + (k & 1u) * HT;
        const float* pbuf = pl + LY::PB + (k % 3u) * 128;
        QT(0);
        f32x16 dE[2];
        if constexpr (ENC_GRAD) {
          dgrad_b3q(P0, Dtile, lane, rn, rkh, dE);
        }
        QT(1);
        // ---- encoding (lane = feature) and, Fourier only, the gradient of its matrix: d sin(w.x)/d w = cos(w.x) x
        const float4 encw[2] = {cenc[i], cenc[32 + i]};
        float Eb[2][2][8];
#pragma unroll
        for (int b = 0; b < 2; ++b) {
          const float inv2pi = 0.15915494309189535f;
          float4 pp[8];
#pragma unroll
          for (int e = 0; e < 8; ++e) pp[e] = *reinterpret_cast<const float4*>(pbuf + 4 * (8 * ((8 * b + e) >> 2) + 4 * hi + (e & 3)));
#pragma unroll
          for (int e = 0; e < 8; ++e) {
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
              Eb[b][m][e] = v;
              if (ENC_GRAD) {      // raw rows carry no weight (their slot of the partial vector is never written)
                const float g = dE[m][8 * b + e] * __builtin_amdgcn_cosf(rev);
                dwf[m][0] = fmaf(g, p.x, dwf[m][0]); dwf[m][1] = fmaf(g, p.y, dwf[m][1]); dwf[m][2] = fmaf(g, p.z, dwf[m][2]);
              }
            }
          }
          __builtin_amdgcn_sched_barrier(0);
        }
        if (ENC_GRAD) {
#pragma unroll
          for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int c = 0; c < 3; ++c) asm volatile("" : "+v"(dwf[m][c]));
        }
        QT(2);
        f32x16 dY[2];                    // dY of layer 0's output, lane = feature: the weight gradient's A operand
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
          for (int r = 0; r < 16; ++r) dY[m][r] = Dtile[COL_OFF(m, r)];
        WAVE_SYNC();
        // ---- the next tile's transfers: the buffer of tile s + 1 is the one just read (tile s - 1)
        if (s + 1 < NP) {
          issue_tile(act0, tile_n0(s + 1), pl_lds + (LY::H1 + ((s + 1) & 1u) * HT) * 4, lane);
          issue_rows(xyz, tile_n0(s + 1), pl_lds + (LY::PB + ((s + 1) % 3u) * 128) * 4, lane);
        }
        __builtin_amdgcn_sched_barrier(0);
        QT(3);
#pragma unroll
        for (int b = 0; b < 2; ++b) {
          B3Op Bx[2];
          Bx[0] = b3_arr(Eb[b][0]); Bx[1] = b3_arr(Eb[b][1]);
#pragma unroll
          for (int mo = 0; mo < 2; ++mo) {
            const B3Op A = (b == 0) ? b3_regs<0>(dY[mo]) : b3_regs<1>(dY[mo]);
            wgrad_b3_half(A, Bx, acc[mo]);
            __builtin_amdgcn_sched_barrier(0);
          }
        }
      } else if (s + 1 < NP) {           // first step: nothing to differentiate yet, only the transfers of tile 1
        issue_tile(act0, tile_n0(s + 1), pl_lds + (LY::H1 + ((s + 1) & 1u) * HT) * 4, lane);
        issue_rows(xyz, tile_n0(s + 1), pl_lds + (LY::PB + ((s + 1) % 3u) * 128) * 4, lane);
      }
    }
    QT(5);
    DMA_WAIT(0);                         // own transfers: the partner role reads them after the barrier
    QT(6);
    __syncthreads();
    QT(7);
  }
#ifdef NGM_B3Q_TIMING
  if (a.debug_cycles && blockIdx.x == gridDim.x / 2 && lane0 == 0 && (wave == 0 || wave == 4)) {
    tq_[8] = __builtin_readcyclecounter() - ts_;
    for (int k = 0; k < 9; ++k) a.debug_cycles[(wave ? 16 : 0) + k] = tq_[k];
  }
#endif
#undef COL_OFF

  // ---- epilogue: the four waves of a role are summed in fixed pair order (all of LDS is free now)
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
  for (int e4 = threadIdx.x; e4 < 2 * 4 * 256; e4 += B3Q_THREADS) {      // [layer][tile][q][lane]
    const int l = e4 >> 10, rest = e4 & 1023;
    const int w0 = (l == 1) ? 0 : 4;                                     // role A waves 0-3 hold layer 1, role B waves 4-7 layer 0
    float4 s4 = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int p = 0; p < B3Q_PAIRS; ++p) {                                // fixed order: deterministic
      const float4 v = *reinterpret_cast<const float4*>(stage + ((w0 + p) * 4 * 256 + rest) * 4);
      s4.x += v.x; s4.y += v.y; s4.z += v.z; s4.w += v.w;
    }
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
  for (int e = threadIdx.x; e < NV * 32; e += B3Q_THREADS) {
    const int k = e >> 5, ii = e & 31;
    float s0 = 0.f;
#pragma unroll
    for (int w = 0; w < B3Q_WAVES; ++w) s0 += stage[(w * NV + k) * 64 + ii] + stage[(w * NV + k) * 64 + 32 + ii];
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
    for (int64_t p = enc_off + threadIdx.x; p < w_off[0]; p += B3Q_THREADS) dst[p] = 0.f;
}

// ------------------------------------------------------------------------------------------------
// returns NGM_E_UNSUPPORTED when this variant does not apply (caller falls back to k_field_bwd_b3 / the fp32-MFMA kernels)
int ngm_launch_field_bwd_b3q(const FieldBwdArgs& a, int blocks, hipStream_t st) {
  const int MI = (a.fc.dim_enc + 31) / 32, MH = (a.fc.dim_hidden + 31) / 32, L = a.fc.num_layers;
  if (!a.act || a.points || a.fc.skip_mode != NGM_SKIP_NO || a.fc.matmul_mode == NGM_MATMUL_F32 || MI != 2 || MH != 2 || L != 2)
    return NGM_E_UNSUPPORTED;
  if (a.fc.encoding != NGM_ENC_FOURIER && a.fc.encoding != NGM_ENC_NERF && a.fc.encoding != NGM_ENC_NONE) return NGM_E_UNSUPPORTED;
  if (!a.hash_xyz || !a.hash_xyz_ready) return NGM_E_UNSUPPORTED;         // positions come from k_stash_bwd
  if ((a.P + 64) * 256 >= ((int64_t)1 << 32)) return NGM_E_UNSUPPORTED;   // 32-bit byte offsets inside a field
  NgmProfScope prof_(NGM_K_FIELD_BWD, st);
#define NGM_LBB3Q(NC, EG)                                                                                             \
  do {                                                                                                                \
    const size_t lds = (size_t)LdsB3q<EG>::TOTAL * sizeof(float);                                                     \
    if (lds > 160 * 1024) return NGM_E_UNSUPPORTED;                                                                   \
    (void)hipFuncSetAttribute((const void*)k_field_bwd_b3q<NC, EG>, hipFuncAttributeMaxDynamicSharedMemorySize,       \
                              (int)lds);                                                                              \
    hipLaunchKernelGGL((k_field_bwd_b3q<NC, EG>), dim3(blocks), dim3(B3Q_THREADS), lds, st, a);                       \
  } while (0)
  if (a.fc.encoding == NGM_ENC_FOURIER) NGM_LBB3Q(false, true);
  else if (a.fc.encoding == NGM_ENC_NERF) NGM_LBB3Q(true, false);
  else NGM_LBB3Q(false, false);
#undef NGM_LBB3Q
  return 0;
}
