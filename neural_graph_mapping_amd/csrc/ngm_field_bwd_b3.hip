// Backward of encoding + MLP for the fused training step on the bf16 matrix pipe: every fp32 operand written
// exactly as hi + mid + lo (three bf16), six products on v_mfma_f32_32x32x16_bf16, fp32 accumulate -- the arithmetic
// of layer_fwd_b3 (ngm_field.h), selected by the same ngm_field_cfg.matmul_mode.  gfx950, 32-sample tiles, 4 waves
// per workgroup (one per SIMD: the whole register file and 27 KB of LDS per wave).
//
// Why a kernel of its own and not a mode of k_field_bwd16s: an MFMA operand holds its contraction index in the
// registers of a lane.  The data gradient contracts over FEATURES (operand lane = sample), the weight gradient over
// SAMPLES (operand lane = feature), so every dY is needed in both orientations.  Here
//   * the data gradient is computed TRANSPOSED, dX^T[s][i] = sum_o dY^T[s][o] W[o][i]: rows = the tile's 32 samples,
//     columns = features, so its C fragment (lane = feature i, registers = 16 of the samples) IS the weight-gradient
//     operand layout of the next layer down -- no data movement for that use;
//   * the other orientation (lane = sample, eight consecutive features per k-block) comes from LDS: the activation
//     tile lands there by DMA in [16-byte chunk][sample] order, each lane reads its feature's column (one ds_read_b32
//     per value), and dY is written back IN PLACE over the activation it was masked with (same addresses: no
//     hazard between lanes), then read as rows (two ds_read_b128 per k-block);
//   * everything per feature -- ReLU masks, bias / output-weight / Fourier-matrix gradients -- is lane-local in
//     the "lane = feature" orientation: sums over the tile's samples are register sums, no LDS reductions.
// "lane = feature" layout: lane (i = lane & 31, hi = lane >> 5), tile m: feature 32 m + i, register r <-> sample
// frow(r, hi) = 8 (r >> 2) + 4 hi + (r & 3) of the tile (the C layout of a 32x32 MFMA with samples on the rows).
// A weight-gradient k-block b takes registers 8b..8b+7 of both operands.
//
// LDS tile [chunk c = feature >> 2][position][4 floats], sample s of chunk c at position s ^ (c & 7): the column read
// of a half-wave (32 features = 8 chunks, one sample) then touches 32 different banks, the row read is a permutation
// of 512 contiguous bytes.  The DMA (global_load_lds_dwordx4, lane p -> LDS base + 16 p) applies the permutation on
// the source side.
//
// Supported: dim_enc and dim_hidden in 33..64, 1-2 hidden layers, Fourier / NeRF / no encoding, skip_mode no, ray mode
// with the forward's activation stash.  Everything else (and matmul_mode f32) keeps the fp32-MFMA kernels.
#include "ngm_bwd_b3.h"

// ------------------------------------------------------------------------------------------------
// HS: H2^T = H1 W1^T (48 MFMAs, the structure of a data gradient: A = rows of the H1 tile, B = W1's forward planes) with the
// tile's ENCODING evaluated in the shadows of those MFMAs.  The two streams are independent (positions vs. H1 rows), and this
// is the one kernel instance with the registers to keep the 32 sines (+ 32 cosines, Fourier) until layer 0 wants them.
// Scheduling by construction, not by sched_group_barrier: every MFMA is followed by one slice of vector work -- one (sample,
// feature-tile) pair of the encoding: ~9 instructions, or one pair of the next k-block's operand split: 11 -- and a full
// scheduling fence; each slice's results are pinned to their slice by an empty volatile asm (their uses are a phase away, and
// instruction selection otherwise sinks the pure sine / cosine down to them, out from under the MFMAs).
template <bool NEED_COS, bool ENC_GRAD>
__device__ __forceinline__ void recompute_with_encoding(const ngm_u32x4* __restrict__ P, const float* __restrict__ tile,
                                                        const float* __restrict__ pb, const float4 (&encw)[2], int lane,
                                                        f32x16 (&Hc)[2], float (&Eb)[2][2][8], float (&Cb)[2][2][8]) {
  const int hi = lane >> 5;
  const float inv2pi = 0.15915494309189535f;
  const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  RowRegs R;
  PlaneRegs W[2];
  float4 pq[2][4];                   // positions of the quarter in work / the next one
  load_rows(tile, lane, R);
  load_planes(P, 0, lane, W[0]);
  auto ldq = [&](int q) __attribute__((always_inline)) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int b = q >> 1, e = 4 * (q & 1) + j;
      pq[q & 1][j] = *reinterpret_cast<const float4*>(pb + 4 * (8 * ((8 * b + e) >> 2) + 4 * hi + (e & 3)));
    }
  };
  ldq(0);
  __builtin_amdgcn_sched_barrier(0);
  B3Op A = b3_rows(R.g[0][0], R.g[0][1]);
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int kb = 0; kb < 4; ++kb) {
    const PlaneRegs& Wk = W[kb & 1];
    if (kb < 3) { load_planes(P, kb + 1, lane, W[(kb + 1) & 1]); ldq(kb + 1); }
    uint32_t nh[4], nm[4], nl[4];
    auto enc = [&](int idx) __attribute__((always_inline)) {           // element idx of quarter kb: sample j, feature tile m
      const int j = idx >> 1, m = idx & 1, b = kb >> 1, e = 4 * (kb & 1) + j;
      float4 p = pq[kb & 1][j];
      asm volatile("" : "+v"(p.x));                                       // ... and its input: not before this slice either
      const float4 w = encw[m];
      const float arg = fmaf(w.z, p.z, fmaf(w.y, p.y, w.x * p.x));
      const float rev = __builtin_amdgcn_fractf(arg * inv2pi);
      const float sn = __builtin_amdgcn_sinf(rev);
      float v = sn;
      if (NEED_COS) v = (w.w == NGM_FK_COS) ? __builtin_amdgcn_cosf(rev) : sn;
      if (m == 0) v = (w.w == NGM_FK_RAW) ? arg : v;                    // raw coordinates are features 0..2
      asm volatile("" : "+v"(v));
      Eb[b][m][e] = v;
      if (ENC_GRAD) {
        float c = __builtin_amdgcn_cosf(rev);
        asm volatile("" : "+v"(c));
        Cb[b][m][e] = c;
      }
    };
    auto spl = [&](int pr) __attribute__((always_inline)) {           // pair pr of the next k-block's eight row values
      if (kb < 3) {
        const float4 g = R.g[kb < 3 ? kb + 1 : 3][pr >> 1];
        float x0 = (pr & 1) ? g.z : g.x, x1 = (pr & 1) ? g.w : g.y;
        asm volatile("" : "+v"(x0), "+v"(x1));
        b3_split2(x0, x1, nh[pr], nm[pr], nl[pr]);
        asm volatile("" : "+v"(nh[pr]), "+v"(nm[pr]), "+v"(nl[pr]));
      }
    };
#define NGM_HS_STEP(PA, PW, NT_, Z, WORK)                                                                       \
    asm volatile("" : "+v"(A.PA));            /* the MFMA not before its step (its operands are ready long before) */ \
    Hc[NT_] = mfma_bf16(A.PA, __builtin_bit_cast(ngm_bf16x8, Wk.PW[NT_]), (Z) ? zero : Hc[NT_]);                \
    WORK;                                                                                                       \
    __builtin_amdgcn_sched_barrier(0);
    NGM_HS_STEP(l, h, 0, kb == 0, enc(0))
    NGM_HS_STEP(l, h, 1, kb == 0, spl(0))
    NGM_HS_STEP(h, l, 0, false, enc(1))
    NGM_HS_STEP(h, l, 1, false, enc(2))
    NGM_HS_STEP(m, m, 0, false, spl(1))
    NGM_HS_STEP(m, m, 1, false, enc(3))
    NGM_HS_STEP(m, h, 0, false, enc(4))
    NGM_HS_STEP(m, h, 1, false, spl(2))
    NGM_HS_STEP(h, m, 0, false, enc(5))
    NGM_HS_STEP(h, m, 1, false, enc(6))
    NGM_HS_STEP(h, h, 0, false, spl(3))
    NGM_HS_STEP(h, h, 1, false, enc(7))
#undef NGM_HS_STEP
    if (kb < 3) {
      A.h = __builtin_bit_cast(ngm_bf16x8, ngm_u32x4{nh[0], nh[1], nh[2], nh[3]});
      A.m = __builtin_bit_cast(ngm_bf16x8, ngm_u32x4{nm[0], nm[1], nm[2], nm[3]});
      A.l = __builtin_bit_cast(ngm_bf16x8, ngm_u32x4{nl[0], nl[1], nl[2], nl[3]});
    }
  }
}

// ------------------------------------------------------------------------------------------------
// FC: the compositing backward of k_stash_bwd fused into the input phase (FieldBwdArgs::fused_comp): the d_out stream then
// carries the forward's (colour, geometry) stash, every wave walks a contiguous ray-aligned range of tiles BACK TO FRONT
// and carries the suffix value Q of the per-ray recursion Q_{k-1} = a_k o_k + (1 - o_k) Q_k from tile to tile.
// HS: half stash (FieldBwdArgs::act_half, L = 2): only layer 0's output was stashed; the output layer's input is recomputed
// from it per tile (LdsB3b).
template <int L, bool NEED_COS, bool ENC_GRAD, bool FC = false, bool HS = false>
__global__ __launch_bounds__(B3B_THREADS) void k_field_bwd_b3(FieldBwdArgs a) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  using LY = LdsB3b<L, ENC_GRAD, HS>;
#ifdef NGM_PROLOGUE_TIMING   // clocks since kernel entry at the prologue's milestones (middle block, thread 0) -> debug_cycles[0..7]
  unsigned long long ptk_[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  const unsigned long long pt0_ = __builtin_readcyclecounter();
#define PTK(i) ptk_[i] = __builtin_readcyclecounter() - pt0_
#else
#define PTK(i)
#endif
  const int f = blockIdx.x % a.F, chunk = blockIdx.x / a.F;
  const int64_t row = a.pr.field_index ? a.pr.field_index[f] : f;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int i = lane & 31, hi = lane >> 5;
  PTK(0);
  ngm_u32x4* planes = reinterpret_cast<ngm_u32x4*>(sm);
  float* wl = sm + LY::PLANES + wave * LY::WAVE_TOTAL;
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
  // point mode (FC = false only): explicit points instead of ray table + distances; the field's pose for the world -> local map
  const bool pts = !FC && a.points != nullptr;
  const char* pts_base = pts ? reinterpret_cast<const char*>(a.points + (int64_t)f * a.P * 3) : nullptr;
  __shared__ float s_pose[12];
  if (pts && threadIdx.x == 0) {
    float div, off;
    scale_consts(a.fc.scale_mode, a.fc.field_radius, &div, &off);
    const bool posed = a.pos != nullptr;
    s_pose[0] = posed ? a.pos[3 * f] : 0.f; s_pose[1] = posed ? a.pos[3 * f + 1] : 0.f; s_pose[2] = posed ? a.pos[3 * f + 2] : 0.f;
    s_pose[3] = posed ? a.quat[4 * f] : 1.f; s_pose[4] = posed ? a.quat[4 * f + 1] : 0.f; s_pose[5] = posed ? a.quat[4 * f + 2] : 0.f;
    s_pose[6] = posed ? a.quat[4 * f + 3] : 0.f; s_pose[7] = div; s_pose[8] = off; s_pose[9] = posed ? 1.f : 0.f;
  }
  {
    const int64_t g0 = (int64_t)f * a.P;
    fs.raytab = pts ? nullptr : reinterpret_cast<const char*>(a.raytab) + 32 * (g0 / a.S);
    fs.dout = reinterpret_cast<const char*>(a.d_out + g0);
    fs.tpair = pts ? nullptr : reinterpret_cast<const char*>(a.stashB + (g0 & ~(int64_t)1));
    fs.par = (uint32_t)(g0 & 1);
    fs.gb = (uint32_t)(g0 & 31);
    fs.act[0] = reinterpret_cast<const char*>(a.act + (g0 >> 5) * 2048);
    fs.act[1] = reinterpret_cast<const char*>(a.act + a.act_layer_stride + (g0 >> 5) * 2048);
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
    } else if (pts) {
      issue_inputs_pts(fs.dout, pts_base, first, end, lane, wl_lds + LY::INB * 4, wl_lds + LY::INB * 4 + 2048);
      issue_inputs_pts(fs.dout, pts_base, first + 16, end, lane, wl_lds + LY::INB * 4 + 1024, wl_lds + LY::INB * 4 + 2048 + 256);
    } else {
      issue_inputs(fs, a.S, first, end, lane, wl_lds + LY::INB * 4);
      issue_inputs(fs, a.S, first + 16, end, lane, wl_lds + LY::INB * 4 + 1024);
    }
    if constexpr (!HS) issue_tile32(fs.act[L - 1], fs.gb, first, end, lane, wl_lds + LY::HL * 4);
    if (L == 2) issue_tile32(fs.act[0], fs.gb, first, end, lane, wl_lds + LY::H1 * 4);
  }
  // per-feature constants (output-layer column, encoding row): LDS, re-read by the phase that needs them -- as
  // loop-long register residents they were spilled to scratch, and a scratch reload waits on vmcnt, i.e. on the DMA.
  // Round 6: every global load of the prologue -- constants, the forward's loss partials, the weights of all plane sets -- is
  // ISSUED here, behind the first tile's transfers, and consumed below: one memory round trip instead of ~9 in a row.
  float4* cwout = reinterpret_cast<float4*>(sm + LY::CONSTS);
  float4* cenc = cwout + 64;
  __shared__ float s_red[FC ? 16 : 1][17];
  __shared__ float s_sums[NGM_NUM_LOSS_SUMS];
  constexpr int NSETS = (ENC_GRAD ? 1 : 0) + (L == 2 ? 1 : 0) + (HS ? 1 : 0);       // plane sets, in build order
  constexpr int NPX = NSETS > 0 ? NSETS * 16 : 1;
  float px_[NPX];                      // 2 granules x 8 weights per set
  {
    int off_[NPX];
    int si = 0;
    if (ENC_GRAD) { planes_offsets<false>(a.fc, 0, 0, *reinterpret_cast<int(*)[8]>(off_ + 16 * si)); planes_offsets<false>(a.fc, 0, 1, *reinterpret_cast<int(*)[8]>(off_ + 16 * si + 8)); ++si; }
    if (L == 2) { planes_offsets<false>(a.fc, 1, 0, *reinterpret_cast<int(*)[8]>(off_ + 16 * si)); planes_offsets<false>(a.fc, 1, 1, *reinterpret_cast<int(*)[8]>(off_ + 16 * si + 8)); ++si; }
    if (HS) { planes_offsets<true>(a.fc, 1, 0, *reinterpret_cast<int(*)[8]>(off_ + 16 * si)); planes_offsets<true>(a.fc, 1, 1, *reinterpret_cast<int(*)[8]>(off_ + 16 * si + 8)); ++si; }
    // one storage-type branch around all of them; layer 0's and layer 1's weights are different tensors: per-set base
    si = 0;
    if (ENC_GRAD) { ngm_ldp_gather<16>(a.pr.w[0], row * a.pr.w_stride[0], *reinterpret_cast<int(*)[16]>(off_ + 16 * si), a.pr.dtype, *reinterpret_cast<float(*)[16]>(px_ + 16 * si)); ++si; }
    if (L == 2) { ngm_ldp_gather<16>(a.pr.w[1], row * a.pr.w_stride[1], *reinterpret_cast<int(*)[16]>(off_ + 16 * si), a.pr.dtype, *reinterpret_cast<float(*)[16]>(px_ + 16 * si)); ++si; }
    if (HS) { ngm_ldp_gather<16>(a.pr.w[1], row * a.pr.w_stride[1], *reinterpret_cast<int(*)[16]>(off_ + 16 * si), a.pr.dtype, *reinterpret_cast<float(*)[16]>(px_ + 16 * si)); ++si; }
  }
  // the forward's loss partials: this thread's share (slot, every 16th partial), at most PMAX of them in registers
  constexpr int PMAX = 20;
  float pv_[FC ? PMAX : 1];
  const int lslot = threadIdx.x & 15, lpart = threadIdx.x >> 4;
  if constexpr (FC) {
    if (a.loss_partials) {
#pragma unroll
      for (int q = 0; q < PMAX; ++q) {
        const int b_ = lpart + 16 * q;
        pv_[q] = (b_ < a.n_partials) ? a.loss_partials[(int64_t)b_ * NGM_NUM_LOSS_SUMS + lslot] : 0.f;
      }
    }
  }
  float4 cw_ = make_float4(0.f, 0.f, 0.f, 0.f), ce_ = make_float4(0.f, 0.f, 0.f, NGM_FK_ZERO);
  float cb_ = 0.f;
  if (threadIdx.x < 64) {
    const int ft = threadIdx.x, H = a.fc.dim_hidden;
    const float* W = a.pr.w[L];
    const int64_t w0 = row * a.pr.w_stride[L];
    if (ft < H) cw_ = make_float4(ngm_ldp(W, w0 + ft, a.pr.dtype), ngm_ldp(W, w0 + H + ft, a.pr.dtype),
                                  ngm_ldp(W, w0 + 2 * H + ft, a.pr.dtype), ngm_ldp(W, w0 + 3 * H + ft, a.pr.dtype));
    ce_ = enc_row_of(a.fc, a.pr, row, ft);
    if constexpr (HS) cb_ = (ft < H) ? ngm_ldp(a.pr.b[1], row * a.pr.b_stride[1] + ft, a.pr.dtype) : 0.f;
  }
  // ---- consumption
  if (threadIdx.x < 64) {
    cwout[threadIdx.x] = cw_;
    cenc[threadIdx.x] = ce_;
    if constexpr (HS) sm[LY::CONSTS + 512 + threadIdx.x] = cb_;
  }
  // FC: the global loss normalisers, as in k_stash_bwd (from the all-reduced sums, or summed here by every workgroup from
  // the forward's per-workgroup partials in k_loss_reduce's fixed order: identical everywhere, deterministic)
  if constexpr (FC) {
    if (a.loss_partials) {
      float s_ = 0.f;
#pragma unroll
      for (int q = 0; q < PMAX; ++q) s_ += pv_[q];           // + 0 beyond the last partial: the same sum, the same order
      for (int b_ = lpart + 16 * PMAX; b_ < a.n_partials; b_ += 16) s_ += a.loss_partials[(int64_t)b_ * NGM_NUM_LOSS_SUMS + lslot];
      s_red[lpart][lslot] = s_;
    }
  }
  {
    int si = 0;
    if (ENC_GRAD) { planes_commit<false>(a.fc, 0, 0, px_ + 16 * si, planes + LY::plane_slot(0) * 3 * PLANE_G); planes_commit<false>(a.fc, 0, 1, px_ + 16 * si + 8, planes + LY::plane_slot(0) * 3 * PLANE_G); ++si; }
    if (L == 2) { planes_commit<false>(a.fc, 1, 0, px_ + 16 * si, planes + LY::plane_slot(1) * 3 * PLANE_G); planes_commit<false>(a.fc, 1, 1, px_ + 16 * si + 8, planes + LY::plane_slot(1) * 3 * PLANE_G); ++si; }
    if (HS) { planes_commit<true>(a.fc, 1, 0, px_ + 16 * si, planes + LY::fwd_slot() * 3 * PLANE_G); planes_commit<true>(a.fc, 1, 1, px_ + 16 * si + 8, planes + LY::fwd_slot() * 3 * PLANE_G); ++si; }
  }
  PTK(1);
  __syncthreads();
  PTK(2);
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
  PTK(3);
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

  PTK(4);
  DMA_WAIT(0);
  PTK(5);
  TICK_DECL;
  TICK(0);
  for (uint32_t it = 0; it < ntiles; ++it) {
    const uint32_t base = first + (uint32_t)((int32_t)it * tstep);
    const uint32_t nxt = base + (uint32_t)tstep;
    const bool more = it + 1u < ntiles;
    float* HLb = wl + LY::HL;
    float* H1b = wl + LY::H1;
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
#ifdef NGM_ABLB_NOCOMP    // timing ablation (results meaningless): no compositing backward, the stash row is the gradient
        {
          WAVE_SYNC();
          *reinterpret_cast<float4*>((hi ? pbuf : pbuf2) + 4 * j) = make_float4(x, y, z, 0.f);
          *reinterpret_cast<float4*>((hi ? obuf : obuf2) + 4 * j) = valid ? dd : make_float4(0.f, 0.f, 0.f, 0.f);
          WAVE_SYNC();
        }
        if (false) {
#else
        {
#endif
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
      float x = fmaf(t, r0.w, r0.x), y = fmaf(t, r1.x, r0.y), z = fmaf(t, r1.y, r0.z);
      if (pts) {                       // the forward's own map world -> scaled field-local (k_field_points_fwd): same bits
        const float* pw = inb + 512 + 64 * h + 3 * jj;
        const Vec3 v = scaled_local_point(Vec3{pw[0], pw[1], pw[2]}, s_pose[9] != 0.f, s_pose[0], s_pose[1], s_pose[2], s_pose[3],
                                          s_pose[4], s_pose[5], s_pose[6], s_pose[7], s_pose[8]);
        x = v.x; y = v.y; z = v.z;
      }
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
    // ---- output layer, lane = feature: dY = relu'(H) * (Wout^T d_out); output-weight and bias gradients.
    // Every LDS read of a phase is issued at its top (sched_barrier keeps it there): one wave per SIMD, so the latency
    // is hidden by this wave's own arithmetic or not at all.
    f32x16 dY[2], Xc[2];
    float Eb[2][2][8];               // the tile's encoding, weight-gradient operand layout (lane = feature, 8 samples per k-block)
    float Cb[2][2][8];               // HS + Fourier: cos of the same arguments, for the Fourier-matrix gradient
    {
      f32x16 Hc[2];
      const float4 wout[2] = {cwout[i], cwout[32 + i]};
      float4 dOa[2][8];              // both halves' d_out rows up front, as for the positions below
      if constexpr (HS) {
        // recompute H2 = relu(W1 H1 + b1) transposed: A = the H1 tile's rows (lane = sample), B = W1's forward planes, C fragment =
        // lane = output feature, registers = samples: the "lane = feature" layout the rest of this phase works in.  Same block
        // structure as a data gradient (row splits of k-block k + 1 under the MFMAs of k-block k).
        const ngm_u32x4* fwdP = planes + LY::fwd_slot() * 3 * PLANE_G;
#ifndef NGM_HS_NO_EARLY_ENC
        // ... with the ENCODING of this tile in the shadows of its 48 MFMAs: sine (and, Fourier, cosine) of the tile's 32 x 64
        // (sample, feature) pairs need only the positions, the recompute only the H1 rows -- two independent streams, and this
        // kernel instance has the registers to keep the 32 + 32 results until layer 0 wants them (the full-stash kernel does not:
        // docs/DESIGN_NOTEBOOK.md 7.1).  A quarter of the pairs per k-block; the vector phase that is left behind layer 0's data
        // gradient is the Fourier-matrix gradient alone.
        {
          const float4 encw[2] = {cenc[i], cenc[32 + i]};
          recompute_with_encoding<NEED_COS, ENC_GRAD>(fwdP, H1b, pb_c, encw, lane, Hc, Eb, Cb);
        }
#else
        {
          RowRegs Rh;
          PlaneRegs Wf;
          load_rows(H1b, lane, Rh);
          load_planes(fwdP, 0, lane, Wf);
          __builtin_amdgcn_sched_barrier(0);
          dgrad_b3(fwdP, Rh, Wf, lane, Hc);
        }
#endif
        // layer 1's input columns (weight gradient operand, ReLU mask of layer 0) and the d_out rows: in flight under the bias / ReLU
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
          for (int r = 0; r < 16; ++r) Xc[m][r] = H1b[COL_OFF(m, r)];
#pragma unroll
        for (int e = 0; e < 8; ++e) dOa[0][e] = *reinterpret_cast<const float4*>(ob_c + 4 * (8 * (e >> 2) + 4 * hi + (e & 3)));
        const float b1v[2] = {sm[LY::CONSTS + 512 + i], sm[LY::CONSTS + 512 + 32 + i]};
        {
          const float4 a0 = accl[0], a1 = accl[64];
          const float2 a2 = reinterpret_cast<const float2*>(accl + 128)[0];
          dwo[0][0] = a0.x; dwo[0][1] = a0.y; dwo[0][2] = a0.z; dwo[0][3] = a0.w;
          dwo[1][0] = a1.x; dwo[1][1] = a1.y; dwo[1][2] = a1.z; dwo[1][3] = a1.w;
          dbh[L - 1][0] = a2.x; dbh[L - 1][1] = a2.y;
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
          for (int r = 0; r < 16; ++r) Hc[m][r] = fmaxf(Hc[m][r] + b1v[m], 0.f);
      } else {
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
          for (int r = 0; r < 16; ++r) Hc[m][r] = HLb[COL_OFF(m, r)];
        if (L == 2) {                  // layer 1's input columns: in flight under the output layer's arithmetic (one wave per SIMD:
                                       // a load issued where it is needed is a stall of a full LDS round trip)
#pragma unroll
          for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int r = 0; r < 16; ++r) Xc[m][r] = H1b[COL_OFF(m, r)];
        }
#pragma unroll
        for (int half = 0; half < 2; ++half)
#pragma unroll
          for (int e = 0; e < 8; ++e) dOa[half][e] = *reinterpret_cast<const float4*>(ob_c + 4 * (8 * ((8 * half + e) >> 2) + 4 * hi + (e & 3)));
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int half = 0; half < 2; ++half) {
        if constexpr (HS) {            // the second half's rows: in flight under the first half's arithmetic (register pressure)
          if (half == 0) {
#pragma unroll
            for (int e = 0; e < 8; ++e) dOa[1][e] = *reinterpret_cast<const float4*>(ob_c + 4 * (8 * ((8 + e) >> 2) + 4 * hi + (e & 3)));
          }
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const int r = 8 * half + e;
          const float4 d = dOa[half][e];
#pragma unroll
          for (int m = 0; m < 2; ++m) {
            const float h = Hc[m][r];
#ifdef NGM_ABLB_NOOUT     // timing ablation (results meaningless): no output-layer arithmetic
            dY[m][r] = h + d.x;
            continue;
#endif
            const float dh = fmaf(wout[m].w, d.w, fmaf(wout[m].z, d.z, fmaf(wout[m].y, d.y, wout[m].x * d.x)));
            const float g = (h > 0.f) ? dh : 0.f;
            dY[m][r] = g;
            dbh[L - 1][m] += g;
            dwo[m][0] = fmaf(d.x, h, dwo[m][0]); dwo[m][1] = fmaf(d.y, h, dwo[m][1]);
            dwo[m][2] = fmaf(d.z, h, dwo[m][2]); dwo[m][3] = fmaf(d.w, h, dwo[m][3]);
          }
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      WAVE_SYNC();
      if constexpr (HS) {
        accl[0] = make_float4(dwo[0][0], dwo[0][1], dwo[0][2], dwo[0][3]);
        accl[64] = make_float4(dwo[1][0], dwo[1][1], dwo[1][2], dwo[1][3]);
        reinterpret_cast<float2*>(accl + 128)[0] = make_float2(dbh[L - 1][0], dbh[L - 1][1]);
      }
#pragma unroll
      for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int r = 0; r < 16; ++r) HLb[COL_OFF(m, r)] = dY[m][r];
      WAVE_SYNC();
    }
    TICK(4);
    float* Dtile = HLb;              // tile holding, as rows, dY of the layer whose input gradients come next
    if constexpr (L == 2) {
      RowRegs R;
      PlaneRegs W0;
      {
        B3Op A0[2] = {b3_regs<0>(dY[0]), b3_regs<0>(dY[1])}, B0[2] = {b3_regs<0>(Xc[0]), b3_regs<0>(Xc[1])};
        __builtin_amdgcn_sched_barrier(0);
        load_rows(HLb, lane, R);       // consumed after the weight gradient
        load_planes(planes + LY::plane_slot(1) * 3 * PLANE_G, 0, lane, W0);
        B3Op A1[2] = {b3_regs<1>(dY[0]), b3_regs<1>(dY[1])}, B1[2] = {b3_regs<1>(Xc[0]), b3_regs<1>(Xc[1])};
        wgrad_b3_block_free(A0, B0, acc[1]);
        NGM_INTERLEAVE(24, 8)
        __builtin_amdgcn_sched_barrier(0);
        wgrad_b3_block(A1, B1, acc[1]);
      }
      __builtin_amdgcn_sched_barrier(0);
      TICK(6);
      f32x16 dX[2];
      dgrad_b3(planes + LY::plane_slot(1) * 3 * PLANE_G, R, W0, lane, dX);
      TICK(7);
      if constexpr (HS) {
        const float2 b0 = reinterpret_cast<const float2*>(accl + 128)[1];
        dbh[0][0] = b0.x; dbh[0][1] = b0.y;
      }
#pragma unroll
      for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float g = (Xc[m][r] > 0.f) ? dX[m][r] : 0.f;
          dY[m][r] = g;
          dbh[0][m] += g;
        }
      WAVE_SYNC();
      if constexpr (HS) reinterpret_cast<float2*>(accl + 128)[1] = make_float2(dbh[0][0], dbh[0][1]);
#pragma unroll
      for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int r = 0; r < 16; ++r) H1b[COL_OFF(m, r)] = dY[m][r];
      WAVE_SYNC();
      Dtile = H1b;
      TICK(9);
    }
    // ---- layer 0.  Data gradient first (only dY's rows are needed), so that cos(.) never has to be kept: the encoding
    // is then evaluated k-block by k-block, each value feeding the Fourier-matrix gradient and the weight-gradient operand
    f32x16 dE[2];
    if constexpr (ENC_GRAD) {
      RowRegs R0;
      PlaneRegs W00;
      load_rows(Dtile, lane, R0);
      load_planes(planes + LY::plane_slot(0) * 3 * PLANE_G, 0, lane, W00);
      __builtin_amdgcn_sched_barrier(0);
      dgrad_b3(planes + LY::plane_slot(0) * 3 * PLANE_G, R0, W00, lane, dE);
    }
    TICK(8);
    const float4 encw[2] = {cenc[i], cenc[32 + i]};
    float4 ppa[2][8];                // both blocks' positions up front: the second block's read latency hides under the first's arithmetic
#ifndef NGM_HS_NO_EARLY_ENC
    constexpr bool EARLY = HS;       // the encoding was evaluated under the recompute's MFMAs
#else
    constexpr bool EARLY = false;
#endif
    if constexpr (!EARLY || ENC_GRAD) {
#pragma unroll
      for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int e = 0; e < 8; ++e) ppa[b][e] = *reinterpret_cast<const float4*>(pb_c + 4 * (8 * ((8 * b + e) >> 2) + 4 * hi + (e & 3)));
    }
    if constexpr (HS && ENC_GRAD) {
      const float4 f0 = accl[192], f1 = accl[256];
      dwf[0][0] = f0.x; dwf[0][1] = f0.y; dwf[0][2] = f0.z;
      dwf[1][0] = f1.x; dwf[1][1] = f1.y; dwf[1][2] = f1.z;
    }
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (EARLY) {
      if constexpr (ENC_GRAD) {      // Fourier-matrix gradient: d sin(w.x)/d w = cos(w.x) x
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            const float4 p = ppa[b][e];
#pragma unroll
            for (int m = 0; m < 2; ++m) {
              const float g = dE[m][8 * b + e] * Cb[b][m][e];
              dwf[m][0] = fmaf(g, p.x, dwf[m][0]); dwf[m][1] = fmaf(g, p.y, dwf[m][1]); dwf[m][2] = fmaf(g, p.z, dwf[m][2]);
            }
          }
        __builtin_amdgcn_sched_barrier(0);
      }
    } else {
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      const float inv2pi = 0.15915494309189535f;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float4 p = ppa[b][e];
#pragma unroll
        for (int m = 0; m < 2; ++m) {
          const float4 w = encw[m];
#ifdef NGM_ABLB_NOENC     // timing ablation (results meaningless): no encoding arithmetic, no Fourier-matrix gradient
          Eb[b][m][e] = p.x + w.x;
          if (ENC_GRAD) dwf[m][0] += dE[m][8 * b + e];
          continue;
#endif
          const float arg = fmaf(w.z, p.z, fmaf(w.y, p.y, w.x * p.x));
          const float rev = __builtin_amdgcn_fractf(arg * inv2pi);
          const float sn = __builtin_amdgcn_sinf(rev);
          float v = sn;
          if (NEED_COS) v = (w.w == NGM_FK_COS) ? __builtin_amdgcn_cosf(rev) : sn;
          if (m == 0) v = (w.w == NGM_FK_RAW) ? arg : v;            // raw coordinates are features 0..2
          Eb[b][m][e] = v;
          if (ENC_GRAD) {    // Fourier only: d sin(w.x)/d w = cos(w.x) x; raw rows carry no weight (their slot is never written)
            const float g = dE[m][8 * b + e] * __builtin_amdgcn_cosf(rev);
            dwf[m][0] = fmaf(g, p.x, dwf[m][0]); dwf[m][1] = fmaf(g, p.y, dwf[m][1]); dwf[m][2] = fmaf(g, p.z, dwf[m][2]);
          }
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    }
    if constexpr (HS && ENC_GRAD) {
      accl[192] = make_float4(dwf[0][0], dwf[0][1], dwf[0][2], 0.f);
      accl[256] = make_float4(dwf[1][0], dwf[1][1], dwf[1][2], 0.f);
    }
    // ---- next tile's transfers + this tile's last matrix work.  On gfx950 a wave's LDS instructions crawl while it
    // has HBM -> LDS transfers it has not waited for (tools/micro/dma_lds.hip: 8 ds_read_b128 behind 8 transfers cost
    // 1400 clocks instead of 250, whether the data has long arrived or not), and with one wave per SIMD nobody fills
    // such holes.  So the transfers are issued where every landing buffer is free and NO LDS instruction follows until
    // the wait: under the layer-0 weight gradient (48 MFMAs + operand splits, registers only).
    WAVE_SYNC();
    TICK(3);
    if constexpr (FC) {
      // the small inputs of the tile after next: its landing buffer (0 for even tiles, 1 for odd ones) was consumed by the
      // pass at the top of this tile (even) or of the previous one (odd)
      if (it + 2u < ntiles) issue_small(base - 64u, (int)(it & 1u));
    }
    if (more) {
      if constexpr (!FC) {
        if (pts) {
          issue_inputs_pts(fs.dout, pts_base, nxt, end, lane, wl_lds + LY::INB * 4, wl_lds + LY::INB * 4 + 2048);
          issue_inputs_pts(fs.dout, pts_base, nxt + 16, end, lane, wl_lds + LY::INB * 4 + 1024, wl_lds + LY::INB * 4 + 2048 + 256);
        } else {
          issue_inputs(fs, a.S, nxt, end, lane, wl_lds + LY::INB * 4);
          issue_inputs(fs, a.S, nxt + 16, end, lane, wl_lds + LY::INB * 4 + 1024);
        }
      }
      const uint32_t u0 = nxt + fs.gb;
      if (((u0 & 31u) == 0u) && (nxt + 32u <= end)) {      // whole tile, aligned with the stash tiles: scalar addressing
        if constexpr (!HS) issue_tile32_fast(fs.act[L - 1], u0 >> 5, lane, wl_lds + LY::HL * 4);
        if (L == 2) issue_tile32_fast(fs.act[0], u0 >> 5, lane, wl_lds + LY::H1 * 4);
      } else {
        if constexpr (!HS) issue_tile32(fs.act[L - 1], fs.gb, nxt, end, lane, wl_lds + LY::HL * 4);
        if (L == 2) issue_tile32(fs.act[0], fs.gb, nxt, end, lane, wl_lds + LY::H1 * 4);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    TICK(5);
    {
      B3Op A0[2] = {b3_regs<0>(dY[0]), b3_regs<0>(dY[1])}, B0[2] = {b3_arr(Eb[0][0]), b3_arr(Eb[0][1])};
      __builtin_amdgcn_sched_barrier(0);
      B3Op A1[2] = {b3_regs<1>(dY[0]), b3_regs<1>(dY[1])}, B1[2] = {b3_arr(Eb[1][0]), b3_arr(Eb[1][1])};
      wgrad_b3_block_free(A0, B0, acc[0]);
      NGM_INTERLEAVE(24, 8)
      __builtin_amdgcn_sched_barrier(0);
      wgrad_b3_block(A1, B1, acc[0]);
    }
    __builtin_amdgcn_sched_barrier(0);
    TICK(10);
    DMA_WAIT(0);
    TICK(2);
    WAVE_SYNC();
  }
#undef COL_OFF
  PTK(6);
  if constexpr (HS) {                 // the per-feature sums back into registers before the staging area takes all of LDS
    const float4 a0 = accl[0], a1 = accl[64], a2 = accl[128], f0 = accl[192], f1 = accl[256];
    dwo[0][0] = a0.x; dwo[0][1] = a0.y; dwo[0][2] = a0.z; dwo[0][3] = a0.w;
    dwo[1][0] = a1.x; dwo[1][1] = a1.y; dwo[1][2] = a1.z; dwo[1][3] = a1.w;
    dbh[L - 1][0] = a2.x; dbh[L - 1][1] = a2.y; dbh[0][0] = a2.z; dbh[0][1] = a2.w;
    dwf[0][0] = f0.x; dwf[0][1] = f0.y; dwf[0][2] = f0.z;
    dwf[1][0] = f1.x; dwf[1][1] = f1.y; dwf[1][2] = f1.z;
  }
  __syncthreads();

  // ---- epilogue: the four waves' accumulators are summed in fixed wave order (all of LDS is free now).
  // Round 6: accumulator tiles AND per-feature vectors are staged before ONE barrier (they were two staging rounds with a
  // barrier each), and the summing passes issue every LDS read of a thread before the first add (they were 8 + 3 dependent
  // iterations): ~11 k clocks per workgroup -> see profiles/r06_small_batch_clocks.txt.
  float* stage = sm;
  constexpr int NT = LY::NT;
  constexpr int NV = 2 * L + 8 + 6 + 4;
  static_assert(LY::EPI_ACC + B3B_WAVES * NV * 64 <= LY::TOTAL, "the per-feature vectors are staged behind the accumulator tiles");
  float* stagev = sm + LY::EPI_ACC;
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
  // per-feature vectors: lane (i, hi) holds the partial sums of feature 32 m + i over its half of the samples
  {
    float* sw = stagev + wave * NV * 64;
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
  int64_t enc_off, w_off[NGM_MAX_LAYERS + 1], b_off[NGM_MAX_LAYERS + 1];
  (void)ngm_param_offsets(&a.fc, &enc_off, w_off, b_off);
  float* dst = a.partials + (int64_t)blockIdx.x * a.p_pad;
  const int D = a.fc.dim_enc, H = a.fc.dim_hidden;
  {
    static_assert(NT * 256 % B3B_THREADS == 0, "whole passes");
    constexpr int NIT = NT * 256 / B3B_THREADS;
    float4 v4[NIT][B3B_WAVES];
#pragma unroll
    for (int it = 0; it < NIT; ++it)
#pragma unroll
      for (int w = 0; w < B3B_WAVES; ++w)
        v4[it][w] = *reinterpret_cast<const float4*>(stage + (w * NT * 256 + it * B3B_THREADS + threadIdx.x) * 4);
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int e4 = it * B3B_THREADS + threadIdx.x;
      float4 s4 = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
      for (int w = 0; w < B3B_WAVES; ++w) {                           // fixed wave order: deterministic
        s4.x += v4[it][w].x; s4.y += v4[it][w].y; s4.z += v4[it][w].z; s4.w += v4[it][w].w;
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
  }
  const bool fourier = ENC_GRAD && a.fc.encoding == NGM_ENC_FOURIER;
  const int n_raw = a.fc.raw_coords ? 3 : 0;
  {
    constexpr int NIV = (NV * 32 + B3B_THREADS - 1) / B3B_THREADS;
    float sv[NIV];
#pragma unroll
    for (int it = 0; it < NIV; ++it) {
      const int e = it * B3B_THREADS + threadIdx.x;
      const int k = min(e >> 5, NV - 1), ii = e & 31;
      float s0 = 0.f;
#pragma unroll
      for (int w = 0; w < B3B_WAVES; ++w) s0 += stagev[(w * NV + k) * 64 + ii] + stagev[(w * NV + k) * 64 + 32 + ii];
      sv[it] = s0;
    }
#pragma unroll
    for (int it = 0; it < NIV; ++it) {
      const int e = it * B3B_THREADS + threadIdx.x;
      if (e >= NV * 32) break;
      const int k = e >> 5, ii = e & 31;
      const float s0 = sv[it];
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
  }
  if (!fourier)   // the encoding slot of the partial vector (if any) carries no gradient
    for (int64_t p = enc_off + threadIdx.x; p < w_off[0]; p += B3B_THREADS) dst[p] = 0.f;
  TICK(11);
  TICK_REPORT
#ifdef NGM_PROLOGUE_TIMING
  PTK(7);
  if (a.debug_cycles && blockIdx.x == gridDim.x / 2 && threadIdx.x == 0) {
    for (int k = 0; k < 8; ++k) a.debug_cycles[k] = ptk_[k];
    for (int k = 8; k < 13; ++k) a.debug_cycles[k] = 0;
    a.debug_cycles[12] = ptk_[7];
  }
#endif
#undef PTK
}

// ------------------------------------------------------------------------------------------------
template <int L, bool HS>
static int launch_bwd_b3(const FieldBwdArgs& a, int blocks, hipStream_t st) {
#define NGM_LBB3(NC, EG)                                                                                              \
  do {                                                                                                                \
    const size_t lds = (size_t)LdsB3b<L, EG, HS>::TOTAL * sizeof(float);                                              \
    if (lds > 160 * 1024) return NGM_E_UNSUPPORTED;                                                                   \
    if (a.fused_comp) {                                                                                               \
      (void)hipFuncSetAttribute((const void*)k_field_bwd_b3<L, NC, EG, true, HS>, hipFuncAttributeMaxDynamicSharedMemorySize, \
                                (int)lds);                                                                            \
      hipLaunchKernelGGL((k_field_bwd_b3<L, NC, EG, true, HS>), dim3(blocks), dim3(B3B_THREADS), lds, st, a);         \
    } else {                                                                                                          \
      (void)hipFuncSetAttribute((const void*)k_field_bwd_b3<L, NC, EG, false, HS>, hipFuncAttributeMaxDynamicSharedMemorySize, \
                                (int)lds);                                                                            \
      hipLaunchKernelGGL((k_field_bwd_b3<L, NC, EG, false, HS>), dim3(blocks), dim3(B3B_THREADS), lds, st, a);        \
    }                                                                                                                 \
  } while (0)
  if (a.fc.encoding == NGM_ENC_FOURIER) NGM_LBB3(false, true);
  else if (a.fc.encoding == NGM_ENC_NERF) NGM_LBB3(true, false);
  else NGM_LBB3(false, false);
#undef NGM_LBB3
  return 0;
}

bool ngm_field_bwd_b3_applies(const FieldBwdArgs& a) {
  const int MI = (a.fc.dim_enc + 31) / 32, MH = (a.fc.dim_hidden + 31) / 32, L = a.fc.num_layers;
  if (!a.act || (a.points && (a.fused_comp || a.act_half)) || a.fc.skip_mode != NGM_SKIP_NO || a.fc.matmul_mode == NGM_MATMUL_F32 || MI != 2 || MH != 2 || L < 1 || L > 2)
    return false;
  if (a.fc.encoding != NGM_ENC_FOURIER && a.fc.encoding != NGM_ENC_NERF && a.fc.encoding != NGM_ENC_NONE) return false;
  if ((a.P + 64) * 256 >= ((int64_t)1 << 32)) return false;   // 32-bit byte offsets inside a field
#ifdef NGM_FAST_BUILD
  if (L != 2) return false;
#endif
  return true;
}
// returns NGM_E_UNSUPPORTED when this variant does not apply (caller falls back to the fp32-MFMA kernels)
int ngm_launch_field_bwd_b3(const FieldBwdArgs& a, int blocks, hipStream_t st) {
  const int L = a.fc.num_layers;
  if (!ngm_field_bwd_b3_applies(a)) return NGM_E_UNSUPPORTED;
  if (a.fused_comp && (a.per_block % (B3B_WAVES * 32) || !a.rayseed)) return NGM_E_INVALID;
  NgmProfScope prof_(NGM_K_FIELD_BWD, st);
  if (a.act_half && L != 2) return NGM_E_INVALID;
  if (L == 2) return a.act_half ? launch_bwd_b3<2, true>(a, blocks, st) : launch_bwd_b3<2, false>(a, blocks, st);
#ifndef NGM_FAST_BUILD
  if (L == 1) return launch_bwd_b3<1, false>(a, blocks, st);
#endif
  return NGM_E_UNSUPPORTED;
}
