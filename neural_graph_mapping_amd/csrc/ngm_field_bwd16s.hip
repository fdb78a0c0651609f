// Backward of encoding + MLP for the fused training step, activations taken from the forward's stash
// (gfx950, 16-sample tiles on v_mfma_f32_16x16x4_f32, 8 waves per workgroup).
//
// k_field_bwd16 (ngm_field_bwd16.hip) recomputes the hidden layers of every sample before it can
// differentiate them: one third of its matrix-core work.  On gfx950 fp32 MFMA and VALU instructions of
// the waves of one SIMD do not overlap (tools/micro/coexec.hip: a wave of v_mfma_f32_16x16x4_f32 next to
// a wave of v_fma_f32 takes the SUM of their solo times), so a SIMD's time is the sum of everything
// it issues and the recompute cannot hide behind anything.  The training forward therefore writes the
// post-ReLU hidden activations (64 floats per sample and layer, ActStash in ngm_field.h) and this
// kernel reads them back: 256 instead of 384 MFMAs per 16-sample tile.
//
// Data movement: the activation rows, the ray-table entry, d_out and t of the NEXT tile travel HBM -> LDS
// with global_load_lds_dwordx4 while the current tile is being differentiated (no staging registers;
// issued from inline asm so that hipcc does not guard every later ds_read with vmcnt(0); the consumer
// waits with explicit counted s_waitcnt).  LDS tiles use a blocked layout the DMA can fill linearly:
//   tile[16 samples][64 features] = 4 blocks of (4 rows x 64 floats) + 8 floats of padding per block;
//   one DMA instruction (64 lanes x 16 B = 1 KiB) fills one block from 4 consecutive stash rows.
// The weight-gradient MFMAs take k-step t = samples {4b + t : b = 0..3}, one row from every block, so
// the 8-float block padding spreads the four rows over all 32 banks (2 lanes per bank = the minimum).
//
// Supported: dim_enc and dim_hidden in 49..64 (4 tiles each), 1-2 hidden layers, Fourier / NeRF / no
// encoding, ray mode.  Everything else keeps the recompute kernel.
#include "ngm_bwd16.h"

// (An earlier build re-derived every lane-dependent LDS offset inside each phase from an opaque copy of the lane id, to
// keep loop-invariant offsets out of registers while the kernel was spilling.  With the spills gone -- aligned weight
// rows, late encoding -- letting the compiler hoist them is 4 % faster.)

// Blocked, swizzled tile: sample n = 4 b + r lives in block b (stride 272 floats = 16 banks), row r (64 floats), and
// its 16-byte chunk c = feature / 4 sits at chunk position c ^ 2r.  A 32-lane half-wave of every access pattern then
// touches 32 different banks: the wgrad operand reads (one row of two blocks: bank offset 16 apart), the C-layout b128
// accesses (4 rows of a block differ in chunk position) and the lane = feature reads (a permutation of one row).
#define BK_STRIDE 272
#define BK_TILE (4 * BK_STRIDE)           // one 16 x 64 tile
__device__ __forceinline__ int bk_idx(int n, int ft) {
  return (n >> 2) * BK_STRIDE + (n & 3) * 64 + ((((ft >> 2) ^ (2 * (n & 3)))) << 2) + (ft & 3);
}

// Row stride of the 16x16 weight blocks in this kernel: only the data-gradient MFMAs read them, four consecutive
// outputs at a time (one ds_read_b128 = four k-steps).  Unpadded rows of 16 floats with the four 16-byte groups of row r
// stored at (group ^ (r >> 2)): each of the instruction's four 16-lane service groups then touches 16 distinct 16-byte
// slots (simulated with the lane groups of MI355X_MICROARCH.md, LDS) -- padded rows of 20 floats were 2-way
// conflicted (SQ_LDS_BANK_CONFLICT 24 % of the LDS cycles of the kernel came from this and the tile stores).
#define S16_RS 16

template <int L>
struct Lds16s {
  using W = Lds16<4, 4, L, S16_RS>;       // weight region
  static constexpr int XE = 0;            // encoding tile
  static constexpr int act(int l) { return l * BK_TILE; }   // input tile of layer l (l = 1..L); act(L) doubles as dY staging
  static constexpr int PBUF = (L + 1) * BK_TILE;             // [16][4] xyz
  static constexpr int OBUF = PBUF + 64;                     // [16][4] d_out
  static constexpr int INBUF = OBUF + 64;                    // input DMA landing buffer, float4 [4][16]
  static constexpr int WAVE_TOTAL = INBUF + 256;
  static constexpr int EPI = B16_WAVES * 4 * 4 * 64;         // epilogue staging
  static constexpr int BODY = B16_WAVES * WAVE_TOTAL;
  static constexpr int TOTAL = W::WTOTAL + (BODY > EPI ? BODY : EPI);
};

// ---- blocked-tile helpers, lane (j = lane & 15, q = lane >> 4) --------------------------------------
__device__ __forceinline__ void store16b(float* buf, int lane, const f32x4 (&V)[4]) {
  const int j = lane & 15, q = lane >> 4;
#pragma unroll
  for (int m = 0; m < 4; ++m) *reinterpret_cast<float4*>(buf + bk_idx(j, 16 * m + 4 * q)) = make_float4(V[m][0], V[m][1], V[m][2], V[m][3]);
}
__device__ __forceinline__ void load16b(const float* buf, int lane, f32x4 (&V)[4]) {
  const int j = lane & 15, q = lane >> 4;
#pragma unroll
  for (int m = 0; m < 4; ++m) {
    const float4 v = *reinterpret_cast<const float4*>(buf + bk_idx(j, 16 * m + 4 * q));
    V[m][0] = v.x; V[m][1] = v.y; V[m][2] = v.z; V[m][3] = v.w;
  }
}

// Weight-gradient accumulate with the accumulator pinned to the AGPR half of the register file.  The
// 2 x 64 accumulator registers live for the whole kernel; left to the allocator (VGPR-form MFMA, one
// 256-register pool) it spills a third of them inside the tile loop, every spill store serialised
// behind its MFMA.  "+a" makes them AGPRs: 128 AGPRs for the accumulators, 128 VGPRs for the rest.
// No software wait states are needed: an accumulator is only consumed as SrcC of the same opcode
// (back-to-back accumulate) and by the epilogue, a barrier away.
__device__ __forceinline__ void mfma16_agpr(f32x4& acc, float a, float b) {
#ifndef NGM_S_AGPRFORM
  acc = mfma16(a, b, acc);
#else
  asm("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+a"(acc) : "v"(a), "v"(b));
#endif
}

// Data-gradient MFMAs in explicit VGPR form.  Once a function uses AGPRs hipcc selects the AGPR form for
// every builtin MFMA, and dX's 16 destination registers would then compete with the 128 accumulator
// AGPRs (it spilled 64 of them per tile).  The compiler cannot see into the asm, so the MFMA -> VALU read
// wait states (11 for an 8-pass MFMA) are supplied by hand: mfma_results_ready() after the last one.
__device__ __forceinline__ f32x4 mfma16_v0(float a, float b) {
  f32x4 d;
  asm("v_mfma_f32_16x16x4_f32 %0, %1, %2, 0" : "=v"(d) : "v"(a), "v"(b));
  return d;
}
__device__ __forceinline__ void mfma16_v(f32x4& acc, float a, float b) {
  asm("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b));
}
__device__ __forceinline__ void mfma_results_ready(f32x4 (&d)[4]) {
  asm volatile("s_nop 15" : "+v"(d[0]), "+v"(d[1]), "+v"(d[2]), "+v"(d[3]));
}

// dX = W^T dY: k-step (mo, r) uses A[i][k=q] = W[16mo + 4q + r][16mi + i] = block(mo,mi)[row 4(i&3)+(i>>2)][4q + r]
template <int BLK>
__device__ __forceinline__ void dgrad16v(const float* __restrict__ W, int lane, const f32x4 (&dY)[4], f32x4 (&dX)[4]) {
  const int i = lane & 15, q = lane >> 4;
  const float* Wl = W + (4 * (i & 3) + (i >> 2)) * S16_RS + 4 * (q ^ (i & 3));   // row = 4 (i & 3) + (i >> 2): swizzle by row >> 2
  // one 16-byte read = the A operands of the four k-steps (mo, r = 0..3) of tile mi; double buffered over mo
  float4 abuf[2][4];
#pragma unroll
  for (int mi = 0; mi < 4; ++mi) abuf[0][mi] = *reinterpret_cast<const float4*>(Wl + mi * BLK);
#pragma unroll
  for (int mo = 0; mo < 4; ++mo) {
    if (mo + 1 < 4) {
#pragma unroll
      for (int mi = 0; mi < 4; ++mi) abuf[(mo + 1) & 1][mi] = *reinterpret_cast<const float4*>(Wl + ((mo + 1) * 4 + mi) * BLK);
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int mi = 0; mi < 4; ++mi) {
        const float4 a4 = abuf[mo & 1][mi];
        const float av = (r == 0) ? a4.x : (r == 1) ? a4.y : (r == 2) ? a4.z : a4.w;
        if (mo == 0 && r == 0) dX[mi] = mfma16_v0(av, dY[0][0]);
        else mfma16_v(dX[mi], av, dY[mo][r]);
      }
    __builtin_amdgcn_sched_barrier(0);
  }
  mfma_results_ready(dX);
}

// dW[16mo + o][16mi + c] += sum_s dY[s][o] X[s][c]; k-step t takes samples 4b + t (b = lane >> 4)
__device__ __forceinline__ void wgrad16b(const float* __restrict__ dbuf, const float* __restrict__ xbuf, int lane,
                                         f32x4 (&acc)[4][4]) {
  const int i = lane & 15, q = lane >> 4;
  // row t of block q, feature 16 m + i: chunk position (4 m + (i >> 2)) ^ 2 t = 4 (m ^ (t >> 1)) + ((i >> 2) ^ 2 (t & 1))
  const int e0 = q * BK_STRIDE + ((i >> 2) << 2) + (i & 3), e1 = q * BK_STRIDE + (((i >> 2) ^ 2) << 2) + (i & 3);
  const float* dl[2] = {dbuf + e0, dbuf + e1};
  const float* xl[2] = {xbuf + e0, xbuf + e1};
  float av[2][4], bv[2][4];
#pragma unroll
  for (int m = 0; m < 4; ++m) { av[0][m] = dl[0][16 * m]; bv[0][m] = xl[0][16 * m]; }
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    if (t + 1 < 4) {
      const int t1 = t + 1;
#pragma unroll
      for (int m = 0; m < 4; ++m) {
        av[t1 & 1][m] = dl[t1 & 1][64 * t1 + 16 * (m ^ (t1 >> 1))];
        bv[t1 & 1][m] = xl[t1 & 1][64 * t1 + 16 * (m ^ (t1 >> 1))];
      }
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int mo = 0; mo < 4; ++mo)
#pragma unroll
      for (int mi = 0; mi < 4; ++mi) mfma16_agpr(acc[mo][mi], av[t & 1][mo], bv[t & 1][mi]);
    __builtin_amdgcn_sched_barrier(0);
  }
}

// lane = feature: sum over the tile's 16 samples
__device__ __forceinline__ float colsum16b(const float* buf, int lane) {
  float v[16];
#pragma unroll
  for (int n = 0; n < 16; ++n) v[n] = buf[bk_idx(n, lane)];
#pragma unroll
  for (int w = 8; w >= 1; w >>= 1)
#pragma unroll
    for (int k = 0; k < w; ++k) v[k] += v[k + w];
  return v[0];
}
// lane = feature: acc[c] += sum_s col[s][lane] * row4[s][c]
template <int NC>
__device__ __forceinline__ void outer16b(const float* colbuf, const float* row4, int lane, float (&acc)[NC]) {
#pragma unroll
  for (int k0 = 0; k0 < 16; k0 += 4) {
    float h[4]; float4 d[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) { h[k] = colbuf[bk_idx(k0 + k, lane)]; d[k] = *reinterpret_cast<const float4*>(row4 + 4 * (k0 + k)); }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      acc[0] = fmaf(d[k].x, h[k], acc[0]);
      acc[1] = fmaf(d[k].y, h[k], acc[1]);
      acc[2] = fmaf(d[k].z, h[k], acc[2]);
      if (NC > 3) acc[NC - 1] = fmaf(d[k].w, h[k], acc[NC - 1]);
    }
    // pin this batch's FMAs here (they otherwise sink below the later batches' loads and every d row stays live)
    if (NC > 3) asm volatile("" : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[NC - 1]));
    else asm volatile("" : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]));
    __builtin_amdgcn_sched_barrier(0);
  }
}

// activation tile (stash slot l-1 = input of layer l) for samples [n0, n0+16): 4 instructions, one block each;
// lane p fetches granule (sample 4b + (p >> 4), chunk p & 15) of the tiled stash
__device__ __forceinline__ void issue_act(const char* sbase, uint32_t gb, uint32_t n0, uint32_t end, int lane, uint32_t lds_tile) {
#pragma unroll
  for (int b = 0; b < 4; ++b) {
    uint32_t n = n0 + 4 * b + (lane >> 4);
    if (n >= end) n = end - 1;
    const uint32_t u = n + gb;
    // LDS position lane -> (row lane >> 4, chunk position lane & 15) holds logical chunk (lane & 15) ^ 2 row
    const uint32_t c = (uint32_t)((lane & 15) ^ (2 * (lane >> 4)));              // logical chunk of this LDS position
    dma16_so(sbase, (((u >> 5) * 16u + c) * 32u + ((u & 31u) ^ (c & 7u))) * 16u, lds_tile + b * BK_STRIDE * 4);   // stash position: ngm_field.h ActStash
  }
}

#ifndef NGM_OLDER_SHARE
#define NGM_OLDER_SHARE 18u     // 32nds of a workgroup's tiles that go to its four older waves (see the tile lists below)
#endif

// per-wave event timeline of the middle workgroup (debug builds, -DNGM_BWD_TIMELINE; a.debug_cycles must hold
// 16 + 8 * 64 words): entry, after the prologue barrier, every tile start, loop end, kernel end
#ifdef NGM_BWD_TIMELINE
#define BTL_DECL                                                                                                    \
  unsigned long long* btl = (a.debug_cycles && blockIdx.x == gridDim.x / 2 && (threadIdx.x & 63) == 0)              \
                                ? a.debug_cycles + 16 + 64 * (threadIdx.x >> 6) : nullptr;                        \
  int btl_n = 0;                                                                                                    \
  const unsigned long long btl_t0 = __builtin_readcyclecounter()
#define BTL(k)                                                                                                      \
  do {                                                                                                              \
    if (btl && btl_n < 64) btl[btl_n] = ((unsigned long long)(k) << 48) | (__builtin_readcyclecounter() - btl_t0);  \
    ++btl_n;                                                                                                        \
  } while (0)
#else
#define BTL_DECL
#define BTL(k)
#endif

// ------------------------------------------------------------------------------------------------
template <int L, bool NEED_COS, bool ENC_GRAD>
__global__ __launch_bounds__(B16_THREADS) void k_field_bwd16s(FieldBwdArgs a) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  BTL_DECL;
  using LY = Lds16s<L>;
  using LW = typename LY::W;
  constexpr int BLK = LW::BLK;
  const int f = blockIdx.x % a.F, chunk = blockIdx.x / a.F;
  const int64_t row = a.pr.field_index ? a.pr.field_index[f] : f;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int j = lane & 15, q = lane >> 4;
  float* wl = sm + LW::WTOTAL + wave * LY::WAVE_TOTAL;
  float* bufE = wl + LY::XE;
  float* bufD = wl + LY::act(L);
  float* pbuf = wl + LY::PBUF;
  float* obuf = wl + LY::OBUF;
  float* inb = wl + LY::INBUF;
  const uint32_t wl_lds = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)wl);

  f32x4 acc0[4][4];
  f32x4 accH[(L > 1) ? (L - 1) : 1][4][4];
#pragma unroll
  for (int mo = 0; mo < 4; ++mo)
#pragma unroll
    for (int mi = 0; mi < 4; ++mi) {
      acc0[mo][mi] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int l = 0; l < L - 1; ++l) accH[l][mo][mi] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
  float dbh[L], dwo[4], dbo[4], dwf[3];
#pragma unroll
  for (int l = 0; l < L; ++l) dbh[l] = 0.f;
#pragma unroll
  for (int c = 0; c < 4; ++c) { dwo[c] = 0.f; dbo[c] = 0.f; }
  dwf[0] = dwf[1] = dwf[2] = 0.f;

  const uint32_t beg = (uint32_t)chunk * (uint32_t)a.per_block, end = (uint32_t)min(a.P, (int64_t)beg + a.per_block);
  // Tile lists.  The SIMD's arbiter favours its older wave (waves 0-3 of the workgroup): with an even split those
  // finished ~2 of 16 tiles ahead of waves 4-7, which then ran their last tiles alone, without a partner to hide
  // LDS / DMA latency behind (wave timeline, -DNGM_BWD_TIMELINE).  So the older waves take NGM_OLDER_SHARE/32 = 18/32 of the tiles (17: 180.6 us, 18: 178.6, 19: 180.5):
  // tiles [0, nA) go round-robin to waves 0-3, tiles [nA, T) to waves 4-7.  Static, hence deterministic.
  const uint32_t T = (end - beg + 15u) >> 4;
  uint32_t nA = ((T * NGM_OLDER_SHARE + 31u) / 32u + 3u) & ~3u;
  if (nA > T) nA = T;
  const bool older = wave < 4;
  const uint32_t first = beg + 16u * (older ? (uint32_t)wave : nA + (uint32_t)(wave - 4));
  const uint32_t lend = older ? min(end, beg + 16u * nA) : end;       // end of this wave's list (samples)
  constexpr uint32_t TSTRIDE = 16 * (B16_WAVES / 2);                  // 4 waves share a list
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
  // Per-wave LDS tiles: R0 (dY of the last layer, later the encoding), act(1) (layer 1's input, later dY / dE
  // staging), act(2) (the last hidden activation, L = 2).  DMA schedule (L = 2), every wait a plain vmcnt(0) at a
  // point where only long-issued transfers are outstanding (counted waits proved fragile: a register spill the
  // compiler drops between two DMA batches shifts the count onto transfers that were only just issued):
  //   tile start      wait -> inputs(t), act(2)(t) are there;  issue act(1)(t) into its (free) tile
  //   before layer 1  wait -> act(1)(t) is there;               issue inputs(t+1), act(2)(t+1) (their tiles are free)
  // L = 1: inputs(t+1) after the inputs are read, act(1)(t+1) at the tile end (its tile doubles as E staging).
  float* R0 = wl + LY::XE;
  float* A1 = wl + LY::act(1);
  float* AL = wl + LY::act(L);
  if (first < lend) {
    issue_inputs(fs, a.S, first, end, lane, wl_lds + LY::INBUF * 4);
    issue_act(fs.act[L - 1], fs.gb, first, end, lane, wl_lds + LY::act(L) * 4);
  }
  // the weights are staged while the first tile's transfers are in flight (their LDS regions are disjoint)
  {
    FieldStage16<4, 4, L, S16_RS, true> stage;
    stage.issue(a.fc, a.pr, row);
    stage.commit(sm, a.fc);
  }
  __syncthreads();
  BTL(1);
  TICK_DECL;
  TICK(0);
  for (uint32_t base = first; base < lend; base += TSTRIDE) {
    const uint32_t n = base + j, nxt = base + TSTRIDE;
    const bool valid = n < end, more = nxt < lend;
    // ---- inputs and the last hidden activation tile (landed while the previous tile was differentiated)
    BTL(2);
    TICK(10);   // loop back-edge
    DMA_WAIT(0);
    TICK(3);    // wait for inputs + last hidden tile
    if (L == 2) issue_act(fs.act[0], fs.gb, base, end, lane, wl_lds + LY::act(1) * 4);
    TICK(5);    // issue act(1)
    {
      const float4* in4 = reinterpret_cast<const float4*>(inb);
      const float4 r0 = in4[j], r1 = in4[16 + j], dd = in4[32 + j], sp = in4[48 + j];
      const uint32_t nc = valid ? n : end - 1;
      const float t = ((nc + fs.par) & 1u) ? sp.z : sp.x;
      const float x = fmaf(t, r0.w, r0.x), y = fmaf(t, r1.x, r0.y), z = fmaf(t, r1.y, r0.z);
      const float4 dout = valid ? dd : make_float4(0.f, 0.f, 0.f, 0.f);
      WAVE_SYNC();
      if (q == 0) {
        *reinterpret_cast<float4*>(pbuf + 4 * j) = make_float4(x, y, z, 0.f);
        *reinterpret_cast<float4*>(obuf + 4 * j) = dout;
        dbo[0] += dout.x; dbo[1] += dout.y; dbo[2] += dout.z; dbo[3] += dout.w;
      }
      WAVE_SYNC();
      if (L == 1 && more) issue_inputs(fs, a.S, nxt, end, lane, wl_lds + LY::INBUF * 4);
    }
    TICK(1);
    // ---- output layer: d_out -> dY_L
    outer16b<4>(AL, obuf, lane, dwo);
    f32x4 dY[4];
    {
      f32x4 Hc[4];
      load16b(AL, lane, Hc);
      const float4 dout = *reinterpret_cast<const float4*>(obuf + 4 * j);
      const float4* w4 = reinterpret_cast<const float4*>(sm + LW::WOUT);
#pragma unroll
      for (int m = 0; m < 4; ++m) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float4 w = w4[16 * m + 4 * q + r];
          const float dh = fmaf(w.w, dout.w, fmaf(w.z, dout.z, fmaf(w.y, dout.y, w.x * dout.x)));
          dY[m][r] = (Hc[m][r] > 0.f) ? dh : 0.f;
        }
        __builtin_amdgcn_sched_barrier(0);   // 4 weight rows at a time: all 16 in flight would be 64 VGPRs at the pressure peak
      }
    }
    WAVE_SYNC();
    store16b(R0, lane, dY);
    WAVE_SYNC();
    dbh[L - 1] += colsum16b(R0, lane);
    TICK(4);
    float* Dlast = R0;      // tile holding dY of layer 0's output
    float* Etile = A1;      // tile that will receive the encoding
    if constexpr (L == 2) {
      TICK(4);
      DMA_WAIT(0);                                   // act(1) of this tile (issued at the tile start)
      TICK(0);    // wait for act(1)
      if (more) {                                    // INBUF and the last-hidden tile are free: next tile's copies
        issue_inputs(fs, a.S, nxt, end, lane, wl_lds + LY::INBUF * 4);
        issue_act(fs.act[1], fs.gb, nxt, end, lane, wl_lds + LY::act(2) * 4);
      }
      TICK(9);    // issue inputs + act(2) of the next tile
      wgrad16b(R0, A1, lane, accH[0]);
      TICK(6);
      f32x4 dX[4], Xl[4];
      dgrad16v<BLK>(sm + LW::w_off(1), lane, dY, dX);
      TICK(7);
      load16b(A1, lane, Xl);
#pragma unroll
      for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int r = 0; r < 4; ++r) dY[m][r] = (Xl[m][r] > 0.f) ? dX[m][r] : 0.f;
      WAVE_SYNC();
      store16b(A1, lane, dY);          // layer 1's input tile now carries dY of layer 0's output
      WAVE_SYNC();
      dbh[0] += colsum16b(A1, lane);
      Dlast = A1; Etile = R0;
      TICK(9);
    }
    // ---- encoding, as late as possible: E and d(enc)/d(arg) are only needed by layer 0's gradients
    f32x4 dEa[4];
    {
      const float4 pt = *reinterpret_cast<const float4*>(pbuf + 4 * j);
      f32x4 E[4];
      encode16<4, NEED_COS, ENC_GRAD, true>(sm + LW::ENCW, q, pt.x, pt.y, pt.z, E, dEa);
      WAVE_SYNC();
      store16b(Etile, lane, E);
      WAVE_SYNC();
    }
    TICK(2);
    wgrad16b(Dlast, Etile, lane, acc0);
    TICK(6);
    if (ENC_GRAD) {
      // dY is re-read from its staging tile: keeping it in registers across the encoding would cost 16 VGPRs
      // at the kernel's pressure peak (and any spill reload waits for the in-flight DMA as well)
      f32x4 dE[4], dYr[4];
      load16b(Dlast, lane, dYr);
      dgrad16v<BLK>(sm + LW::w_off(0), lane, dYr, dE);
#pragma unroll
      for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int r = 0; r < 4; ++r) dE[m][r] *= dEa[m][r];
      TICK(7);
      WAVE_SYNC();
      store16b(Dlast, lane, dE);
      WAVE_SYNC();
      outer16b<3>(Dlast, pbuf, lane, dwf);
      TICK(8);
    }
    // L = 1: the activation tile doubled as E staging and is free only now
    WAVE_SYNC();
    if (L == 1 && more) issue_act(fs.act[0], fs.gb, nxt, end, lane, wl_lds + LY::act(1) * 4);
  }
  DMA_WAIT(0);
  TICK(10);
  BTL(3);
  __syncthreads();
  BTL(4);
#ifdef NGM_ABLS_NOEPI   // timing ablation: keep the accumulators alive with one store, skip the reduction
  {
    float keep = dbh[0] + dwo[0] + dwo[1] + dwo[2] + dwo[3] + dwf[0] + dwf[1] + dwf[2] + dbo[0] + dbo[1] + dbo[2] + dbo[3];
#pragma unroll
    for (int mo = 0; mo < 4; ++mo)
#pragma unroll
      for (int mi = 0; mi < 4; ++mi)
#pragma unroll
        for (int r = 0; r < 4; ++r) keep += acc0[mo][mi][r] + accH[0][mo][mi][r];
    if (keep == 12345.f) a.partials[0] = 1.f;
  }
  return;
#endif
  // all of LDS is free now (the barrier above was the last use of weights and tiles): all C tiles in two rounds
  constexpr int EPI = (L == 2) ? 16 : 8;                     // 2 rounds for either depth
  static_assert(LW::WTOTAL + B16_WAVES * LY::WAVE_TOTAL >= B16_WAVES * EPI * 256, "epilogue staging does not fit");
  bwd16_epilogue<4, 4, L, ENC_GRAD, EPI>(a, sm, acc0, accH, dbh, dwo, dwf, dbo);
  BTL(5);
  TICK(11);
  TICK_REPORT
}

// ------------------------------------------------------------------------------------------------
template <int L>
static int launch_bwd16s(const FieldBwdArgs& a, int blocks, hipStream_t st) {
  const size_t lds = (size_t)Lds16s<L>::TOTAL * sizeof(float);
  if (lds > 160 * 1024) return NGM_E_UNSUPPORTED;
#define NGM_LB16S(NC, EG)                                                                                             \
  do {                                                                                                                \
    (void)hipFuncSetAttribute((const void*)k_field_bwd16s<L, NC, EG>, hipFuncAttributeMaxDynamicSharedMemorySize,     \
                              (int)lds);                                                                              \
    hipLaunchKernelGGL((k_field_bwd16s<L, NC, EG>), dim3(blocks), dim3(B16_THREADS), lds, st, a);                     \
  } while (0)
  if (a.fc.encoding == NGM_ENC_FOURIER) NGM_LB16S(false, true);
  else if (a.fc.encoding == NGM_ENC_NERF) NGM_LB16S(true, false);
  else NGM_LB16S(false, false);
#undef NGM_LB16S
  return 0;
}

// returns NGM_E_UNSUPPORTED when the stash variant does not apply (caller falls back to the recompute kernels)
int ngm_launch_field_bwd16s(const FieldBwdArgs& a, int blocks, hipStream_t st) {
  const int TI = (a.fc.dim_enc + 15) / 16, TH = (a.fc.dim_hidden + 15) / 16, L = a.fc.num_layers;
  if (!a.act || a.points || a.fc.skip_mode != NGM_SKIP_NO || a.fc.encoding == NGM_ENC_PERMUTO || a.fc.encoding == NGM_ENC_TRIPLANE || TI != 4 || TH != 4 || L < 1 || L > 2) return NGM_E_UNSUPPORTED;
  if ((a.P + 64) * 256 >= ((int64_t)1 << 32)) return NGM_E_UNSUPPORTED;   // 32-bit byte offsets inside a field
  NgmProfScope prof_(NGM_K_FIELD_BWD, st);
  if (L == 2) return launch_bwd16s<2>(a, blocks, st);
#ifndef NGM_FAST_BUILD
  if (L == 1) return launch_bwd16s<1>(a, blocks, st);
#endif
  return NGM_E_UNSUPPORTED;
}
