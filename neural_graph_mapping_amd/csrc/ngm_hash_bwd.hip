// k_hash_mlp_bwd: backward of the reference's DEFAULT network (config/neural_graph_map.yaml:6-20: permutohedral hash
// encoding, 16 levels x 2 features, one hidden layer of 32 units) in the fused training step, on the bf16 matrix pipe with
// the exact three-way split of ngm_field.h (fp32-exact products, fp32 accumulate) -- the arithmetic of k_field_bwd_b3
// for 32-wide layers, selected by the same ngm_field_cfg.matmul_mode.
//
// The generic kernel that served this shape (k_field_bwd16: 16-sample tiles, fp32 MFMA, every operand through LDS
// staging tiles, five wave barriers per tile) took 66 us for 7x less arithmetic than the Fourier network's backward.
// Here a tile is 32 samples on one wave and 36 MFMAs:
//   * the encoding E (32 features per sample) comes from the forward's stash (no simplex search, no table gathers) by
//     HBM -> LDS DMA, in the [16-byte chunk][sample ^ (chunk & 7)] image that serves both orientations an MFMA operand
//     can ask for: rows (lane = sample, two ds_read_b128 per k-block) and columns (lane = feature, ds_read_b32,
//     bank-conflict free);
//   * H^T = relu(E W0^T + b) is recomputed TRANSPOSED (A = rows of E, B = planes of W0): its C fragment has lane = hidden
//     unit, registers = 16 of the samples -- the orientation in which the output-layer gradient, the ReLU mask, the bias
//     and output-weight gradients are lane-local and in which dY is directly the weight gradient's A operand (no stash of
//     H: 12 MFMAs are cheaper than 2 x 67 MB of HBM traffic);
//   * dY goes through a 4 KB scratch tile once to be read as rows, the B operand of dE^T = W0^T dY^T, whose C fragment has
//     lane = sample, registers = features: each lane stores its sample's (level, 2 features) pairs as float2 into the
//     level-major dL/dE array k_hash_grad reads -- 256 contiguous bytes per half-wave and level;
//   * the scaled sample positions k_hash_grad needs are written by k_stash_bwd, which has the ray entry and the distance
//     in registers anyway (bit-identical: the same fmaf of the same operands as the forward), not here.
// Registers: one 32x32 accumulator (16) + ~120 -> two waves per SIMD (8 per workgroup), so the hardware overlaps one
// wave's splits with the other's MFMAs and nothing has to be software-pipelined.  Deterministic: fixed tile lists, fixed
// summation order in the epilogue.
//
// Supported: permutohedral encoding with dim_enc <= 32, one hidden layer of <= 32 units, skip_mode no, ray mode with the
// forward's encoding stash, matmul_mode auto / bf16x3.  Everything else keeps k_field_bwd16.
#include "ngm_bwd_b3.h"

#define HB_WAVES 8
#define HB_THREADS 512
#define HB_PG 128                 // 16-byte granules of one weight plane: [kb 2][kh 2][n 32]

struct LdsHB {
  static constexpr int PLANES = 2 * 3 * HB_PG * 4;          // floats: planes of W0 for the recompute, then for the data gradient
  static constexpr int CONSTS = PLANES;                      // float4 wout[32], float b0[32]
  static constexpr int WAVES = CONSTS + 192;
  static constexpr int ET = 0, DT = 1024, OB = 2048;         // per wave: encoding tile, dY scratch tile, d_out rows (64 x float4: the
  static constexpr int WAVE_TOTAL = 2048 + 768;              // DMA instruction moves 64 x 16 bytes); fused compositing: three such
                                                             // blocks [stash row | (t, T) pair], [ray entry 0 | 1], [loss seeds 0 | 1]
  static constexpr int BODY = WAVES + HB_WAVES * WAVE_TOTAL;
  static constexpr int EPI = HB_WAVES * 1024 + HB_WAVES * 9 * 64;
  static constexpr int TOTAL = BODY > EPI ? BODY : EPI;
};

// one 32-feature encoding tile, samples [n0, n0 + 32) of the field (clamped to end - 1): 4 transfers of two chunks each
__device__ __forceinline__ void hb_issue_tile(const char* sbase, uint32_t gb, uint32_t n0, uint32_t end, int lane, uint32_t lds_tile) {
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const uint32_t c = 2 * k + (lane >> 5);
    uint32_t n = n0 + (((uint32_t)lane & 31u) ^ (c & 7u));
    if (n >= end) n = end - 1;
    const uint32_t u = n + gb;
    dma16_so(sbase, (((u >> 5) * 8u + c) * 32u + ((u & 31u) ^ (c & 7u))) * 16u, lds_tile + k * 1024);
  }
}

// FC: the compositing backward of k_stash_bwd fused in (FieldBwdArgs::fused_comp, as in k_field_bwd_b3): d_out carries the
// forward's (colour, geometry) stash, every wave walks a contiguous ray-aligned range of tiles back to front carrying the
// suffix value of the per-ray recursion, and writes the scaled sample positions k_hash_grad searches the simplices of.
template <bool FC>
__global__ __launch_bounds__(HB_THREADS) void k_hash_mlp_bwd(FieldBwdArgs a) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  using LY = LdsHB;
  const int f = blockIdx.x % a.F, chunk = blockIdx.x / a.F;
  const int64_t row = a.pr.field_index ? a.pr.field_index[f] : f;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int i = lane & 31, hi = lane >> 5;
  const int D = a.fc.dim_enc, H = a.fc.dim_hidden;
  ngm_u32x4* Pf = reinterpret_cast<ngm_u32x4*>(sm);          // B operand of the recompute: lane (o, kh) holds W0[o][16 kb + 8 kh + e]
  ngm_u32x4* Pd = Pf + 3 * HB_PG;                              // A operand of the data gradient: lane (i, kh) holds W0[16 kb + 8 kh + e][i]
  float4* cwout = reinterpret_cast<float4*>(sm + LY::CONSTS);
  float* cb0 = sm + LY::CONSTS + 128;
  float* wl = sm + LY::WAVES + wave * LY::WAVE_TOTAL;
  float* Et = wl + LY::ET;
  float* Dt = wl + LY::DT;
  float* ob = wl + LY::OB;
  const uint32_t wl_lds = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)wl);

  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  float dbh = 0.f, dwo[4] = {0.f, 0.f, 0.f, 0.f}, dbo[4] = {0.f, 0.f, 0.f, 0.f};

  const uint32_t beg = (uint32_t)chunk * (uint32_t)a.per_block, bend = (uint32_t)min(a.P, (int64_t)beg + a.per_block);
  constexpr uint32_t TSTRIDE = 32 * HB_WAVES;
  uint32_t first, end, ntiles;
  int32_t tstep;
  if constexpr (FC) {
    const uint32_t wr = (uint32_t)a.per_block / HB_WAVES, wb = min(bend, beg + (uint32_t)wave * wr);
    end = min(bend, wb + wr);
    ntiles = (end - wb + 31u) >> 5;
    first = wb + 32u * (ntiles - 1u);
    tstep = -32;
  } else {
    end = bend;
    first = beg + 32u * (uint32_t)wave;
    ntiles = first < end ? (end - first + TSTRIDE - 1u) / TSTRIDE : 0u;
    tstep = (int32_t)TSTRIDE;
  }
  const int64_t g0 = (int64_t)f * a.P;
  const uint32_t gb = (uint32_t)(g0 & 31);
  const char* act = reinterpret_cast<const char*>(a.act + (g0 >> 5) * 1024);
  const char* dout = reinterpret_cast<const char*>(a.d_out + g0);
  const char* tpair = reinterpret_cast<const char*>(a.stashB + (g0 & ~(int64_t)1));
  const uint32_t par = (uint32_t)(g0 & 1);
  const char* raytab = reinterpret_cast<const char*>(a.raytab) + 32 * (g0 / a.S);
  const char* seeds = FC ? reinterpret_cast<const char*>(a.rayseed) + 32 * (g0 / a.S) : nullptr;
  const float inv_s = 1.0f / (float)a.S;
  auto issue = [&](uint32_t n0) __attribute__((always_inline)) {
    uint32_t n = n0 + (uint32_t)i;                                     // both halves fetch the 32 rows: the transfer is 64 x 16 bytes
    if (n >= end) n = end - 1;
    if constexpr (FC) {
      // wave-uniform bases + 32-bit lane offsets (per-lane 64-bit pointers kept across the loop were spilled, and a scratch
      // reload between two transfers waits for the first to land)
      const uint32_t ray = (uint32_t)fdiv_idx32((int)n, inv_s, a.S);
      dma16(hi ? tpair + 8 * (size_t)((n + par) & ~1u) : dout + 16 * (size_t)n, wl_lds + LY::OB * 4);
      dma16_so_c(raytab, 32u * ray + 16u * (uint32_t)hi, wl_lds + LY::OB * 4 + 1024);
      dma16_so_c(seeds, 32u * ray + 16u * (uint32_t)hi, wl_lds + LY::OB * 4 + 2048);
    } else {
      dma16(dout + 16 * (size_t)n, wl_lds + LY::OB * 4);
    }
    const uint32_t u0 = n0 + gb;
    if (((u0 & 31u) == 0u) && (n0 + 32u <= end)) {                     // whole tile, aligned with the stash tiles: one linear 4 KB copy
      const char* b0 = act + (size_t)__builtin_amdgcn_readfirstlane(u0 >> 5) * 4096;
      dma16_x4(b0, (uint32_t)lane * 16u, wl_lds + LY::ET * 4);
    } else {
      hb_issue_tile(act, gb, n0, end, lane, wl_lds + LY::ET * 4);
    }
  };
  if (ntiles) issue(first);
  // FC: this thread's share of the forward's loss partials, in flight together with the weights below (round 6: summed in a
  // loop of its own after the planes were built, the prologue was two more memory round trips long)
  constexpr int PMAX = 20;
  float pv_[FC ? PMAX : 1];
  if constexpr (FC) {
    if (a.loss_partials && threadIdx.x < 256) {
      const int slot = threadIdx.x & 15, part = threadIdx.x >> 4;
#pragma unroll
      for (int q = 0; q < PMAX; ++q) {
        const int b_ = part + 16 * q;
        pv_[q] = (b_ < a.n_partials) ? a.loss_partials[(int64_t)b_ * NGM_NUM_LOSS_SUMS + slot] : 0.f;
      }
    }
  }
  // weight planes + per-unit constants while the first tile travels
  {
    const float* W = a.pr.w[0];
    const int64_t w0 = row * a.pr.w_stride[0];
    if (threadIdx.x < 2 * HB_PG) {
      const bool dg = threadIdx.x >= HB_PG;
      const int g = threadIdx.x & (HB_PG - 1);
      const int n = g & 31, kh = (g >> 5) & 1, kb = g >> 6;
      float x[8];
      int off[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int k = 16 * kb + 8 * kh + e;
        const int o = dg ? k : n, c = dg ? n : k;
        off[e] = (o < H && c < D) ? o * D + c : 0;
      }
      ngm_ldp_gather<8>(W, w0, off, a.pr.dtype, x);                 // all eight loads in flight together
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int k = 16 * kb + 8 * kh + e;
        const int o = dg ? k : n, c = dg ? n : k;
        x[e] = (o < H && c < D) ? x[e] : 0.f;
      }
      ngm_bf16x8 h, m, lo;
      b3_split8(x, h, m, lo);
      ngm_u32x4* P = dg ? Pd : Pf;
      P[g] = __builtin_bit_cast(ngm_u32x4, h);
      P[HB_PG + g] = __builtin_bit_cast(ngm_u32x4, m);
      P[2 * HB_PG + g] = __builtin_bit_cast(ngm_u32x4, lo);
    } else if (threadIdx.x < 2 * HB_PG + 32) {
      const int ft = threadIdx.x - 2 * HB_PG;
      const float* Wo = a.pr.w[1];
      const int64_t wo0 = row * a.pr.w_stride[1];
      cwout[ft] = (ft < H) ? make_float4(ngm_ldp(Wo, wo0 + ft, a.pr.dtype), ngm_ldp(Wo, wo0 + H + ft, a.pr.dtype),
                                         ngm_ldp(Wo, wo0 + 2 * H + ft, a.pr.dtype), ngm_ldp(Wo, wo0 + 3 * H + ft, a.pr.dtype))
                           : make_float4(0.f, 0.f, 0.f, 0.f);
      cb0[ft] = (ft < H) ? ngm_ldp(a.pr.b[0], row * a.pr.b_stride[0] + ft, a.pr.dtype) : 0.f;
    }
  }
  // FC: the global loss normalisers, as in k_stash_bwd / k_field_bwd_b3
  __shared__ float s_red[FC ? 16 : 1][17];
  __shared__ float s_sums[NGM_NUM_LOSS_SUMS];
  __shared__ __attribute__((aligned(16))) float s_k[8];
  if constexpr (FC) {
    if (a.loss_partials && threadIdx.x < 256) {
      const int slot = threadIdx.x & 15, part = threadIdx.x >> 4;
      float sacc = 0.f;
#pragma unroll
      for (int q = 0; q < PMAX; ++q) sacc += pv_[q];               // + 0 beyond the last partial: the same sum in the same order
      for (int b = part + 16 * PMAX; b < a.n_partials; b += 16) sacc += a.loss_partials[(int64_t)b * NGM_NUM_LOSS_SUMS + slot];
      s_red[part][slot] = sacc;
    }
  }
  __syncthreads();
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
      if (a.rc.photometric_mode == NGM_PHOTO_L2) k_photo = 2.0f * k_photo;
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
  float carryQ = 0.f;
  if constexpr (FC) {
    if (ntiles) carryQ = comp_suffix_beyond(a, g0, end, lane, inv_s, s_k[0], s_k[1], s_k[2]);
  }

  // lane-constant LDS offsets (floats): element (feature i, sample frow(r, hi)) of a tile sits at
  //   (i >> 2) * 128 + (i & 3) + 4 * ((8 (r >> 2) + 4 hi + (r & 3)) ^ (i >> 2))
  int col[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) col[j] = (i >> 2) * 128 + (i & 3) + 4 * ((4 * hi + j) ^ (i >> 2));
#define COL_OFF(r) (col[(r) & 3] + 32 * ((r) >> 2))
  const float4 wout = cwout[i];
  const float bias = cb0[i];
  const int64_t NP = (int64_t)a.F * a.P;
  const int nlev = a.fc.nr_levels;

  DMA_WAIT(0);
  WAVE_SYNC();
  for (uint32_t it = 0; it < ntiles; ++it) {
    const uint32_t base = first + (uint32_t)((int32_t)it * tstep);
    const uint32_t nxt = base + (uint32_t)tstep;
    const bool more = it + 1u < ntiles;
    if constexpr (FC) {
      // ---- k_stash_bwd's arithmetic on this tile, lane = sample (lanes 32..63 mirror 0..31 and are the identity of the
      // scan); the stash rows in `ob` are replaced by dL/d(raw outputs), which is what the rest of the tile reads there
      const float4* blk = reinterpret_cast<const float4*>(ob);
      const float4 dd = blk[i], sp = blk[32 + i], r0 = blk[64 + i], r1 = blk[96 + i], q0 = blk[128 + i], q1 = blk[160 + i];
      const uint32_t n = base + (uint32_t)i;
      const bool valid = n < end;
      const uint32_t nc = valid ? n : end - 1;
      const bool odd = ((nc + par) & 1u) != 0u;
      const float t = odd ? sp.z : sp.x, T = odd ? sp.w : sp.y;
      const int rayi = fdiv_idx32((int)nc, inv_s, a.S);
      const int k = (int)nc - rayi * a.S, kr = a.S - 1 - k;
      const float4 kn = *reinterpret_cast<const float4*>(s_k);
      const float k_photo = kn.x, k_depth = kn.y, k_term = kn.z, k_fs = kn.w, k_ts = s_k[4];
      const float dzc = r1.z, gt = r1.w, geom = dd.w;
      const float dC0 = k_photo * q0.x, dC1 = k_photo * q0.y, dC2 = k_photo * q0.z, dD = k_depth * q0.w, dT = k_term * q1.x;
      const float depth = -(dzc * t);
      float dodg = 0.f;
      const float occ = occ_pointwise_fast(a.rc.geometry_mode, a.rc.geometry_factor, geom, &dodg);
      const float ak = dC0 * dd.x + dC1 * dd.y + dC2 * dd.z + dD * depth + dT;
      const bool live = valid && hi == 0;
      float A = live ? ak * occ : 0.f, B = live ? 1.0f - occ : 1.0f;
      seg_rscan_affine32(A, B, live ? kr : 0, lane);
      const float Qend = (live && kr > 31 - i) ? carryQ : 0.f;
      const float nA = lane_next(A, 0.f), nB = lane_next(B, 1.f);
      const float Qk = (kr >= 1) ? fmaf(nB, Qend, nA) : Qend;
      carryQ = lane_value(fmaf(B, Qend, A), 0);
      const float tau = a.rc.truncation_distance, cf = a.rc.color_factor;
      const float w = occ * T;
      float dg = T * (ak - Qk) * dodg;
      const float thr = (gt - tau) * (gt != 0.0f ? 1.0f : 0.0f);
      if (t < thr) dg += k_fs * (geom * tau - tau) * tau;
      const float dl = gt - t;
      if (fabsf(dl) < tau && gt != 0.0f) dg += k_ts * (geom * tau - dl) * tau;
      if (a.rc.overwrite_behind_camera && dzc * t > 0.f) dg = 0.f;
      const float4 dv = valid ? make_float4(cf * w * dC0, cf * w * dC1, cf * w * dC2, dg) : make_float4(0.f, 0.f, 0.f, 0.f);
      WAVE_SYNC();                               // every lane has read its rows
      if (hi == 0) {
        *reinterpret_cast<float4*>(ob + 4 * i) = dv;
        if (valid) {                             // the position k_hash_grad searches the simplex of (the forward's own fmaf)
          typedef float v4f __attribute__((ext_vector_type(4)));
          const v4f pxyz = {fmaf(t, r0.w, r0.x), fmaf(t, r1.x, r0.y), fmaf(t, r1.y, r0.z), 0.f};
          ngm_store_wt(reinterpret_cast<ngm_v4f_*>(a.hash_xyz + g0 + n), pxyz);      // written through: read by k_hash_grad (another launch)
        }
      }
      WAVE_SYNC();
    } else if (base + 32u > end) {              // last, partial tile: rows past the end carry no gradient
      if (base + (uint32_t)lane >= end && lane < 32) {
        float z = 0.f;
        asm volatile("" : "+v"(z));              // materialised here (hoisted out of the loop it was spilled to scratch)
        *reinterpret_cast<float4*>(ob + 4 * lane) = make_float4(z, z, z, z);
      }
      WAVE_SYNC();
    }
    // ---- the rows of E and the first k-block's planes; the rest is fetched where its latency hides behind arithmetic
    float4 er[2][2];
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
      const int c0 = 4 * kb + 2 * hi;
      er[kb][0] = *reinterpret_cast<const float4*>(Et + tile_chunk(c0, i));
      er[kb][1] = *reinterpret_cast<const float4*>(Et + tile_chunk(c0 + 1, i));
    }
    ngm_u32x4 wf[2][3];
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int p = 0; p < 3; ++p) wf[kb][p] = Pf[p * HB_PG + (kb * 2 + hi) * 32 + i];
    const float4 drow = *reinterpret_cast<const float4*>(ob + 4 * i);
    dbo[0] += drow.x; dbo[1] += drow.y; dbo[2] += drow.z; dbo[3] += drow.w;
    // ---- H^T = E W0^T + b, transposed: lane = hidden unit, register r <-> sample frow(r, hi)
    f32x16 Hc;
#pragma unroll
    for (int r = 0; r < 16; ++r) Hc[r] = bias;
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
      const B3Op A = b3_rows(er[kb][0], er[kb][1]);
      const ngm_bf16x8 wh = __builtin_bit_cast(ngm_bf16x8, wf[kb][0]), wm = __builtin_bit_cast(ngm_bf16x8, wf[kb][1]),
                       wlo = __builtin_bit_cast(ngm_bf16x8, wf[kb][2]);
      Hc = mfma_bf16(A.l, wh, Hc);
      Hc = mfma_bf16(A.h, wlo, Hc);
      Hc = mfma_bf16(A.m, wm, Hc);
      Hc = mfma_bf16(A.m, wh, Hc);
      Hc = mfma_bf16(A.h, wm, Hc);
      Hc = mfma_bf16(A.h, wh, Hc);
    }
    // ---- output layer, lane = hidden unit: dY = relu'(.) (Wout^T d_out); output-weight and hidden-bias gradients
    float dY[16];
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      float4 dO[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) dO[e] = *reinterpret_cast<const float4*>(ob + 4 * (8 * ((8 * half + e) >> 2) + 4 * hi + (e & 3)));
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int r = 8 * half + e;
        const float4 d = dO[e];
        const float pre = Hc[r];
        const float h = fmaxf(pre, 0.f);
        const float dh = fmaf(wout.w, d.w, fmaf(wout.z, d.z, fmaf(wout.y, d.y, wout.x * d.x)));
        const float g = (pre > 0.f) ? dh : 0.f;
        dY[r] = g;
        dbh += g;
        dwo[0] = fmaf(d.x, h, dwo[0]); dwo[1] = fmaf(d.y, h, dwo[1]); dwo[2] = fmaf(d.z, h, dwo[2]); dwo[3] = fmaf(d.w, h, dwo[3]);
      }
    }
    // ---- dY through the scratch tile: written as columns, read as rows
#pragma unroll
    for (int r = 0; r < 16; ++r) Dt[COL_OFF(r)] = dY[r];
    WAVE_SYNC();
    float4 dr[2][2];
    ngm_u32x4 wd[2][3];
    float Ec[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) Ec[r] = Et[COL_OFF(r)];
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
      const int c0 = 4 * kb + 2 * hi;
      dr[kb][0] = *reinterpret_cast<const float4*>(Dt + tile_chunk(c0, i));
      dr[kb][1] = *reinterpret_cast<const float4*>(Dt + tile_chunk(c0 + 1, i));
#pragma unroll
      for (int p = 0; p < 3; ++p) wd[kb][p] = Pd[p * HB_PG + (kb * 2 + hi) * 32 + i];
    }
    WAVE_SYNC();
    // ---- the next tile's transfers: every landing buffer has been read, no LDS instruction follows until the wait
    if (more) issue(nxt);
    __builtin_amdgcn_sched_barrier(0);
    // ---- weight gradient: dW0[o][i] += sum_s dY[s][o] E[s][i], both operands lane = feature
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      float ya[8], xa[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) { ya[e] = dY[8 * b + e]; xa[e] = Ec[8 * b + e]; }
      const B3Op A = b3_arr(ya), Bx = b3_arr(xa);
      acc = mfma_bf16(A.l, Bx.h, acc);
      acc = mfma_bf16(A.h, Bx.l, acc);
      acc = mfma_bf16(A.m, Bx.m, acc);
      acc = mfma_bf16(A.m, Bx.h, acc);
      acc = mfma_bf16(A.h, Bx.m, acc);
      acc = mfma_bf16(A.h, Bx.h, acc);
    }
    // ---- data gradient dE^T[i][s] = sum_o W0[o][i] dY[s][o]: lane = sample, register r <-> feature frow(r, hi)
    f32x16 dE;
#pragma unroll
    for (int r = 0; r < 16; ++r) dE[r] = 0.f;
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
      const B3Op Bd = b3_rows(dr[kb][0], dr[kb][1]);
      const ngm_bf16x8 wh = __builtin_bit_cast(ngm_bf16x8, wd[kb][0]), wm = __builtin_bit_cast(ngm_bf16x8, wd[kb][1]),
                       wlo = __builtin_bit_cast(ngm_bf16x8, wd[kb][2]);
      dE = mfma_bf16(wlo, Bd.h, dE);
      dE = mfma_bf16(wh, Bd.l, dE);
      dE = mfma_bf16(wm, Bd.m, dE);
      dE = mfma_bf16(wm, Bd.h, dE);
      dE = mfma_bf16(wh, Bd.m, dE);
      dE = mfma_bf16(wh, Bd.h, dE);
    }
    if (base + (uint32_t)i < end) {
      // lane-dependent part of the address once (sample, the half's first level); the eight (q, p) offsets are wave-uniform
      int li = i, lh = hi;
      asm volatile("" : "+v"(li), "+v"(lh));       // re-derived per tile: hoisted, the 64-bit lane part of the address was
                                                    // spilled and its reload waited for the transfers issued above
      float2* dst = a.hash_dE + (g0 + base + li) + (int64_t)(2 * lh) * NP;
#pragma unroll
      for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int p = 0; p < 2; ++p) {
          const int lv = 4 * q + p;                        // level = lv + 2 hi: features 2 level, 2 level + 1 = frow(4 q + 2 p, hi), + 1
          if (lv + 2 * hi < nlev) {
            typedef float v2f __attribute__((ext_vector_type(2)));
            const v2f v = {dE[4 * q + 2 * p], dE[4 * q + 2 * p + 1]};
            ngm_store_wt(reinterpret_cast<ngm_v2f_*>(dst + (int64_t)lv * NP), v);      // read once, by k_hash_grad: written through (k_hash_grad 80.5 -> 76.7 us)
          }
        }
    }
    DMA_WAIT(0);
    WAVE_SYNC();
  }
#undef COL_OFF
  __syncthreads();

  // ---- epilogue: the eight waves' accumulators in fixed wave order (all of LDS is free now)
  float* stage = sm;
#pragma unroll
  for (int q = 0; q < 4; ++q)
    *reinterpret_cast<float4*>(stage + ((wave * 4 + q) * 64 + lane) * 4) = make_float4(acc[4 * q], acc[4 * q + 1], acc[4 * q + 2], acc[4 * q + 3]);
  float* vec = stage + HB_WAVES * 1024 + wave * 9 * 64;
  vec[lane] = dbh;
#pragma unroll
  for (int c = 0; c < 4; ++c) vec[(1 + c) * 64 + lane] = dwo[c];
#pragma unroll
  for (int c = 0; c < 4; ++c) vec[(5 + c) * 64 + lane] = wave_sum(dbo[c]);
  __syncthreads();
  int64_t enc_off, w_off[NGM_MAX_LAYERS + 1], b_off[NGM_MAX_LAYERS + 1];
  (void)ngm_param_offsets(&a.fc, &enc_off, w_off, b_off);
  float* dstp = a.partials + (int64_t)blockIdx.x * a.p_pad;
  if (threadIdx.x < 256) {
    const int e4 = threadIdx.x;
    float4 s4 = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int w = 0; w < HB_WAVES; ++w) {
      const float4 v = *reinterpret_cast<const float4*>(stage + (w * 256 + e4) * 4);
      s4.x += v.x; s4.y += v.y; s4.z += v.z; s4.w += v.w;
    }
    const int q = e4 >> 6, ln = e4 & 63;
    const int o0 = 8 * q + 4 * (ln >> 5), c = ln & 31;        // rows frow(4 q + j, hi) = o0 + j
    if (c < D) {
      float* d = dstp + w_off[0] + (int64_t)o0 * D + c;
      if (o0 < H) d[0] = s4.x;
      if (o0 + 1 < H) d[D] = s4.y;
      if (o0 + 2 < H) d[2 * D] = s4.z;
      if (o0 + 3 < H) d[3 * D] = s4.w;
    }
  } else for (int e = threadIdx.x - 256; e < 9 * 32; e += HB_THREADS - 256) {
    const int k = e >> 5, ii = e & 31;
    const float* v0 = stage + HB_WAVES * 1024;
    float s0 = 0.f;
#pragma unroll
    for (int w = 0; w < HB_WAVES; ++w) s0 += v0[(w * 9 + k) * 64 + ii] + v0[(w * 9 + k) * 64 + 32 + ii];
    if (k == 0) { if (ii < H) dstp[b_off[0] + ii] = s0; }
    else if (k < 5) { if (ii < H) dstp[w_off[1] + (int64_t)(k - 1) * H + ii] = s0; }
    else if (ii == 0) dstp[b_off[1] + (k - 5)] = 0.25f * s0;     // both halves of a wave added the same 32 rows, and wave_sum put
                                                                 // the wave's total into both lanes read here
  }
}

bool ngm_hash_mlp_bwd_applies(const FieldBwdArgs& a) {
  if (a.fc.encoding != NGM_ENC_PERMUTO || !a.act || a.points || !a.raytab || !a.stashB || a.fc.skip_mode != NGM_SKIP_NO ||
      a.fc.matmul_mode == NGM_MATMUL_F32 || a.fc.num_layers != 1 || a.fc.dim_enc > 32 || a.fc.dim_hidden > 32 || a.fc.dim_enc <= 16 ||
      a.fc.dim_out != 4 || !a.hash_dE)
    return false;
  return (a.P + 64) * 128 < ((int64_t)1 << 32);     // 32-bit byte offsets inside a field
}
// returns NGM_E_UNSUPPORTED when this kernel does not apply (caller falls back to k_field_bwd16)
int ngm_launch_hash_mlp_bwd(const FieldBwdArgs& a, int blocks, hipStream_t st) {
  if (!ngm_hash_mlp_bwd_applies(a)) return NGM_E_UNSUPPORTED;
  if (a.fused_comp ? (!a.rayseed || !a.hash_xyz || a.per_block % (HB_WAVES * 32)) : !a.hash_xyz_ready) return NGM_E_UNSUPPORTED;
  NgmProfScope prof_(NGM_K_FIELD_BWD, st);
  const size_t lds = (size_t)LdsHB::TOTAL * sizeof(float);
  if (a.fused_comp) {
    (void)hipFuncSetAttribute((const void*)k_hash_mlp_bwd<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(k_hash_mlp_bwd<true>, dim3(blocks), dim3(HB_THREADS), lds, st, a);
  } else {
    (void)hipFuncSetAttribute((const void*)k_hash_mlp_bwd<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(k_hash_mlp_bwd<false>, dim3(blocks), dim3(HB_THREADS), lds, st, a);
  }
  return 0;
}
