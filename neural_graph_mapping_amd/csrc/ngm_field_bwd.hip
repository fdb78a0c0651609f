// Backward of the per-field MLP + encoding on the matrix cores (gfx950).
//
// Input per sample: position (explicit points, or ray table + saved distance) and dL/d(raw outputs).
// Per 32-sample tile a wave (a) recomputes the forward activations (nothing but 24 B/sample was saved
// by the forward), (b) back-propagates through the layers with dgrad MFMAs whose B operands are the
// C-layout registers of the previous step, and (c) accumulates the weight gradients with wgrad MFMAs
// whose operands are the activations / pre-activation gradients transposed through a small per-wave
// LDS staging buffer (samples become the contraction index).  Weight-gradient accumulators stay in
// registers for the whole kernel; every workgroup writes ONE partial gradient vector, reduced (in a
// fixed order -> deterministic) by k_grad_reduce.
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include "ngm_field.h"
#include "ngm_launch.h"

#define WAVE_SYNC()                                        \
  do {                                                     \
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); \
    __builtin_amdgcn_wave_barrier();                       \
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); \
  } while (0)

template <int MI, int MH, int L, bool CAT = false>
struct BwdLds {
  using LY = FieldLds<MI, MH, L, CAT>;
  static constexpr int STR_E = 32 * MI + 4;
  static constexpr int STR_H = 32 * MH + 4;
  static constexpr int STR_D = (STR_E > STR_H) ? STR_E : STR_H;
  static constexpr int X0 = 0;                                    // E       [32][STR_E]
  static constexpr int x_off(int l) { return l == 0 ? 0 : 32 * STR_E + (l - 1) * 32 * STR_H; }  // X_l
  static constexpr int DBUF = 32 * STR_E + (L - 1) * 32 * STR_H;   // [32][STR_D]
  static constexpr int PBUF = DBUF + 32 * STR_D;                   // [32][4] positions
  static constexpr int OBUF = PBUF + 128;                          // [32][4] dL/dout
  static constexpr int WAVE_TOTAL = OBUF + 128;
  static constexpr int TOTAL = LY::TOTAL + NGM_WAVES_PER_BLOCK * WAVE_TOTAL;
};

// one bit per C-layout register: register r of tile m is positive -> bit 16 m + r
template <int M>
__device__ __forceinline__ uint32_t relu_bits(const f32x16 (&H)[M]) {
  static_assert(M * 16 <= 32, "one 32-bit mask per layer");
  uint32_t b = 0;
#pragma unroll
  for (int m = 0; m < M; ++m)
#pragma unroll
    for (int r = 0; r < 16; ++r) b |= (H[m][r] > 0.f ? 1u : 0u) << (16 * m + r);
  return b;
}

// C-layout registers -> staging buffer [sample][feature]; one 16-byte store per 4 registers
template <int M>
__device__ __forceinline__ void store_tile(float* buf, int stride, int lane, const f32x16 (&V)[M]) {
  const int j = lane & 31, hi = lane >> 5;
#pragma unroll
  for (int m = 0; m < M; ++m)
#pragma unroll
    for (int q = 0; q < 4; ++q)
      *reinterpret_cast<float4*>(buf + j * stride + 32 * m + 8 * q + 4 * hi) =
          make_float4(V[m][4 * q], V[m][4 * q + 1], V[m][4 * q + 2], V[m][4 * q + 3]);
}
template <int M>
__device__ __forceinline__ void load_tile(const float* buf, int stride, int lane, f32x16 (&V)[M]) {
  const int j = lane & 31, hi = lane >> 5;
#pragma unroll
  for (int m = 0; m < M; ++m)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const float4 v = *reinterpret_cast<const float4*>(buf + j * stride + 32 * m + 8 * q + 4 * hi);
      V[m][4 * q] = v.x; V[m][4 * q + 1] = v.y; V[m][4 * q + 2] = v.z; V[m][4 * q + 3] = v.w;
    }
}

// dX = W^T dY  (dgrad): M = input features (MIN tiles), K = output features (MOUT tiles).
template <int MIN, int MOUT>
__device__ __forceinline__ void layer_dgrad(const float* __restrict__ W, int lane, const f32x16 (&dY)[MOUT],
                                            f32x16 (&dX)[MIN]) {
  const int i = lane & 31, hi = lane >> 5;
  const int lane_off = (col_r(i) * 2 + col_hi(i)) * NGM_WGS + 4 * hi;
#pragma unroll
  for (int mi = 0; mi < MIN; ++mi)
#pragma unroll
    for (int r = 0; r < 16; ++r) dX[mi][r] = 0.f;
  constexpr int NG = MOUT * 4;
  float abuf[2][4][MIN];
  const float* Wl = W + lane_off;
#pragma unroll
  for (int rr = 0; rr < 4; ++rr)
#pragma unroll
    for (int mi = 0; mi < MIN; ++mi) abuf[0][rr][mi] = Wl[(0 * MIN + mi) * 16 * 2 * NGM_WGS + frow(rr, 0)];
#pragma unroll
  for (int g = 0; g < NG; ++g) {
    if (g + 1 < NG) {
      const int mo1 = (4 * (g + 1)) / 16, r1 = (4 * (g + 1)) % 16;
#pragma unroll
      for (int rr = 0; rr < 4; ++rr)
#pragma unroll
        for (int mi = 0; mi < MIN; ++mi)
          abuf[(g + 1) & 1][rr][mi] = Wl[(mo1 * MIN + mi) * 16 * 2 * NGM_WGS + frow(r1 + rr, 0)];
    }
    __builtin_amdgcn_sched_barrier(0);
    const int mo = (4 * g) / 16, r0 = (4 * g) % 16;
#pragma unroll
    for (int rr = 0; rr < 4; ++rr)
#pragma unroll
      for (int mi = 0; mi < MIN; ++mi) dX[mi] = mfma32(abuf[g & 1][rr][mi], dY[mo][r0 + rr], dX[mi]);
    __builtin_amdgcn_sched_barrier(0);
  }
}

// dW += dY^T (x) X^T  (wgrad): contraction over the 32 samples of the tile, operands from staging LDS.
template <int MOUT, int MIN>
__device__ __forceinline__ void layer_wgrad(const float* __restrict__ dbuf, int dstr, const float* __restrict__ xbuf,
                                            int xstr, int lane, f32x16 (&acc)[MOUT][MIN]) {
  const int i = lane & 31, hi = lane >> 5;
  // operands of k-step group g+1 (4 sample pairs) are fetched while group g multiplies
  float av[2][4][MOUT], bv[2][4][MIN];
  const float* dl = dbuf + (16 * hi) * dstr + i;
  const float* xl = xbuf + (16 * hi) * xstr + i;
#pragma unroll
  for (int tt = 0; tt < 4; ++tt) {
#pragma unroll
    for (int mo = 0; mo < MOUT; ++mo) av[0][tt][mo] = dl[tt * dstr + 32 * mo];
#pragma unroll
    for (int mi = 0; mi < MIN; ++mi) bv[0][tt][mi] = xl[tt * xstr + 32 * mi];
  }
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    if (g + 1 < 4) {
#pragma unroll
      for (int tt = 0; tt < 4; ++tt) {
#pragma unroll
        for (int mo = 0; mo < MOUT; ++mo) av[(g + 1) & 1][tt][mo] = dl[(4 * (g + 1) + tt) * dstr + 32 * mo];
#pragma unroll
        for (int mi = 0; mi < MIN; ++mi) bv[(g + 1) & 1][tt][mi] = xl[(4 * (g + 1) + tt) * xstr + 32 * mi];
      }
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int tt = 0; tt < 4; ++tt)
#pragma unroll
      for (int mo = 0; mo < MOUT; ++mo)
#pragma unroll
        for (int mi = 0; mi < MIN; ++mi) acc[mo][mi] = mfma32(av[g & 1][tt][mo], bv[g & 1][tt][mi], acc[mo][mi]);
    __builtin_amdgcn_sched_barrier(0);
  }
}

// column sums over the tile's samples with lane = feature (M*32 features; M==1 splits samples 2-way)
template <int M>
__device__ __forceinline__ float colsum(const float* buf, int stride, int lane) {
  constexpr int NF = 32 * M, PARTS = 64 / NF, PER = 32 / PARTS;
  const int fl = lane % NF, sp = lane / NF;
  // all LDS reads first (one latency), then a pairwise tree
  float v[PER];
#pragma unroll
  for (int k = 0; k < PER; ++k) v[k] = buf[(sp * PER + k) * stride + fl];
#pragma unroll
  for (int w = PER / 2; w >= 1; w >>= 1)
#pragma unroll
    for (int k = 0; k < w; ++k) v[k] += v[k + w];
  return v[0];
}

// acc[c] += sum_s col[s][fl] * row4[s][c]  (lane = feature): LDS reads batched 8 samples at a time
template <int M, int NC>
__device__ __forceinline__ void outer_accum(const float* colbuf, int stride, const float* row4, int lane, float (&acc)[NC]) {
  constexpr int NF = 32 * M, PARTS = 64 / NF, PER = 32 / PARTS;
  const int fl = lane % NF, sp = lane / NF;
#pragma unroll
  for (int k0 = 0; k0 < PER; k0 += 8) {
    float h[8];
    float4 d[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int s = sp * PER + k0 + k;
      h[k] = colbuf[s * stride + fl];
      d[k] = *reinterpret_cast<const float4*>(row4 + 4 * s);
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      acc[0] = fmaf(d[k].x, h[k], acc[0]);
      acc[1] = fmaf(d[k].y, h[k], acc[1]);
      acc[2] = fmaf(d[k].z, h[k], acc[2]);
      if (NC > 3) acc[3] = fmaf(d[k].w, h[k], acc[3]);
    }
  }
}

template <int MI, int MH, int L, bool NEED_COS, bool ENC_GRAD, int HASH, bool CAT = false>
__global__ __launch_bounds__(NGM_BLOCK) void k_field_bwd(FieldBwdArgs a) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  using LY = FieldLds<MI, MH, L, CAT>;
  using BL = BwdLds<MI, MH, L, CAT>;
  constexpr int MC = CAT ? MH + MI : MH;       // input tiles of the layers after the first (concat: hidden ++ encoding)
  const int f = blockIdx.x % a.F, chunk = blockIdx.x / a.F;
  const int64_t row = a.pr.field_index ? a.pr.field_index[f] : f;
  {
    FieldStage<MI, MH, L, CAT> fstage;       // every parameter load in flight at once, then the permuting LDS writes
    fstage.issue(a.fc, a.pr, row);
    fstage.commit(sm, a.fc);
  }
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int j = lane & 31, hi = lane >> 5;
  float* wl = sm + LY::TOTAL + wave * BL::WAVE_TOTAL;
  float* bufD = wl + BL::DBUF;
  float* pbuf = wl + BL::PBUF;
  float* obuf = wl + BL::OBUF;

  float div, off;
  scale_consts(a.fc.scale_mode, a.fc.field_radius, &div, &off);
  const bool ray_mode = a.points == nullptr;
  const bool posed = a.pos != nullptr;
  float px = 0, py = 0, pz = 0, qw = 1, qx = 0, qy = 0, qz = 0;
  if (!ray_mode && posed) {
    px = a.pos[3 * f]; py = a.pos[3 * f + 1]; pz = a.pos[3 * f + 2];
    qw = a.quat[4 * f]; qx = a.quat[4 * f + 1]; qy = a.quat[4 * f + 2]; qz = a.quat[4 * f + 3];
  }

  // ---- register-resident gradient accumulators
  f32x16 acc0[MH][MI];
  f32x16 accH[(L > 1) ? (L - 1) : 1][MH][MH];
#pragma unroll
  for (int mo = 0; mo < MH; ++mo) {
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc0[mo][mi][r] = 0.f;
#pragma unroll
    for (int l = 0; l < L - 1; ++l)
#pragma unroll
      for (int mi = 0; mi < MH; ++mi)
#pragma unroll
        for (int r = 0; r < 16; ++r) accH[l][mo][mi][r] = 0.f;
  }
  // concat: d(W_l[:, H:]) of the layers after the first (their encoding columns), and of the output layer
  f32x16 accE[(CAT && L > 1) ? (L - 1) : 1][MH][MI];
  float dwoE[4] = {0.f, 0.f, 0.f, 0.f};      // lane = encoding feature
  if constexpr (CAT) {
#pragma unroll
    for (int l = 0; l < L - 1; ++l)
#pragma unroll
      for (int mo = 0; mo < MH; ++mo)
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
          for (int r = 0; r < 16; ++r) accE[l][mo][mi][r] = 0.f;
  }
  float dbh[L];            // lane = hidden feature
  float dwo[4], dbo[4];    // output layer: dwo lane = hidden feature; dbo per sample lane
  float dwf[3];            // lane = encoding feature
#pragma unroll
  for (int l = 0; l < L; ++l) dbh[l] = 0.f;
#pragma unroll
  for (int c = 0; c < 4; ++c) { dwo[c] = 0.f; dbo[c] = 0.f; }
  dwf[0] = dwf[1] = dwf[2] = 0.f;

  const HashCtx hc = make_hash_ctx(a.fc, a.pr, row, HASH == 1 ? a.lattice_grad + (int64_t)f * a.lattice_grad_stride : nullptr);
  const TriCtx tc = make_tri_ctx(a.fc, a.pr, row, HASH == 2 ? a.tri_acc + (int64_t)f * a.tri_numel : nullptr);
  const bool add_enc = MI <= MH && a.fc.skip_mode == NGM_SKIP_ADD;
  const int64_t beg = (int64_t)chunk * a.per_block, end = min(a.P, beg + a.per_block);
  for (int64_t base = beg + wave * 32; base < end; base += 32 * NGM_WAVES_PER_BLOCK) {
    const int64_t n = base + j;
    const bool valid = n < end;
    // ---- sample position (scaled field-local) and incoming gradient
    float x = 0, y = 0, z = 0;
    float4 dout = make_float4(0.f, 0.f, 0.f, 0.f);
    if (valid) {
      const int64_t g = (int64_t)f * a.P + n;
      if (ray_mode) {
        const int64_t ray = g / a.S;
        const float4 r0 = reinterpret_cast<const float4*>(a.raytab)[2 * ray];
        const float4 r1 = reinterpret_cast<const float4*>(a.raytab)[2 * ray + 1];
        const float t = a.stashB[g].x;
        x = fmaf(t, r0.w, r0.x); y = fmaf(t, r1.x, r0.y); z = fmaf(t, r1.y, r0.z);
      } else {
        const float* p = a.points + g * 3;
        Vec3 v{p[0], p[1], p[2]};
        if (posed) { v = Vec3{v.x - px, v.y - py, v.z - pz}; v = quat_rotate_inv(qw, qx, qy, qz, v); }
        x = v.x / div + off; y = v.y / div + off; z = v.z / div + off;
      }
      dout = a.d_out[g];
    }
    // ---- forward recompute, staging every layer input
    f32x16 E[1][MI], dEa[MI];
    if constexpr (HASH == 1) encode_hash(sm + LY::ENCW, hc, hi, x, y, z, E[0][0]);
    else if constexpr (HASH == 2) encode_triplane<MI>(tc, hi, x, y, z, E[0]);
    else encode_sample<MI, NEED_COS, ENC_GRAD>(sm + LY::ENCW, hi, x, y, z, E[0], dEa);
    WAVE_SYNC();   // previous tile's readers of the staging buffers are done (in-order LDS) - compiler fence
    store_tile<MI>(wl + BL::x_off(0), BL::STR_E, lane, E[0]);
    if (hi == 0) {
      *reinterpret_cast<float4*>(pbuf + 4 * j) = make_float4(x, y, z, 0.f);
      *reinterpret_cast<float4*>(obuf + 4 * j) = dout;
    }
    f32x16 Hc[1][MH];
    layer_fwd<MI, MH, 1>(sm + LY::w_off(0), sm + LY::b_off(0), lane, E, Hc);
    // skip_mode "add": in_{l+1} = relu(z_l) + [enc; 0] (models.py:162-169).  The ReLU mask can then no longer be read
    // off the staged layer input, so it is kept as one bit per register (MH * 16 <= 32 bits per layer).
    uint32_t rmask[L];
    rmask[0] = relu_bits<MH>(Hc[0]);
    if (add_enc) skip_add<MI, MH, 1>(Hc, E);
#pragma unroll
    for (int l = 1; l < L; ++l) {
      store_tile<MH>(wl + BL::x_off(l), BL::STR_H, lane, Hc[0]);
      f32x16 Hn[1][MH];
      if constexpr (CAT) {
        f32x16 Xc[1][MC];
#pragma unroll
        for (int m = 0; m < MH; ++m) Xc[0][m] = Hc[0][m];
#pragma unroll
        for (int m = 0; m < MI; ++m) Xc[0][MH + m] = E[0][m];
        layer_fwd<MC, MH, 1>(sm + LY::w_off(l), sm + LY::b_off(l), lane, Xc, Hn);
      } else {
        layer_fwd<MH, MH, 1>(sm + LY::w_off(l), sm + LY::b_off(l), lane, Hc, Hn);
      }
#pragma unroll
      for (int m = 0; m < MH; ++m) Hc[0][m] = Hn[0][m];
      rmask[l] = relu_bits<MH>(Hc[0]);
      if (add_enc) skip_add<MI, MH, 1>(Hc, E);
    }
    f32x16 dEsum[MI];          // add mode: gradient reaching the encoding through the skip connections
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
      for (int r = 0; r < 16; ++r) dEsum[mi][r] = 0.f;
    // ---- output layer gradients (VALU, lane = hidden feature)
    store_tile<MH>(bufD, BL::STR_D, lane, Hc[0]);
    WAVE_SYNC();
    outer_accum<MH, 4>(bufD, BL::STR_D, obuf, lane, dwo);
    if constexpr (CAT) outer_accum<MI, 4>(wl + BL::x_off(0), BL::STR_E, obuf, lane, dwoE);   // encoding columns of W_out
    if (hi == 0) { dbo[0] += dout.x; dbo[1] += dout.y; dbo[2] += dout.z; dbo[3] += dout.w; }
    // dH_L = W_out^T dout, masked by the ReLU of the last hidden layer
    f32x16 dY[MH];
    {
      const float4* w4 = reinterpret_cast<const float4*>(sm + LY::WOUT);
#pragma unroll
      for (int mi = 0; mi < MH; ++mi)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float4 w = w4[32 * mi + frow(r, 0) + 4 * hi];
          const float dh = fmaf(w.w, dout.w, fmaf(w.z, dout.z, fmaf(w.y, dout.y, w.x * dout.x)));
          if (add_enc && mi < MI) dEsum[mi < MI ? mi : 0][r] += dh;
          dY[mi][r] = ((rmask[L - 1] >> (16 * mi + r)) & 1u) ? dh : 0.f;
        }
      if constexpr (CAT) {       // d(enc) through the output layer's encoding columns
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const float4 w = w4[32 * (MH + mi) + frow(r, 0) + 4 * hi];
            dEsum[mi][r] += fmaf(w.w, dout.w, fmaf(w.z, dout.z, fmaf(w.y, dout.y, w.x * dout.x)));
          }
      }
    }
    // ---- hidden layers, last to first
#pragma unroll
    for (int l = L - 1; l >= 0; --l) {
      WAVE_SYNC();
      store_tile<MH>(bufD, BL::STR_D, lane, dY);
      WAVE_SYNC();
      dbh[l] += colsum<MH>(bufD, BL::STR_D, lane);
      if (l == 0) {
        layer_wgrad<MH, MI>(bufD, BL::STR_D, wl + BL::x_off(0), BL::STR_E, lane, acc0);
        if constexpr (HASH == 2) {
          // triplane: d loss / d planes scattered straight from the lane's dE registers (fixed-point atomics)
          f32x16 dE[MI];
          layer_dgrad<MI, MH>(sm + LY::w_off(0), lane, dY, dE);
          if (add_enc || CAT) {          // what reached the encoding through the skip connections of the later layers
#pragma unroll
            for (int mi = 0; mi < MI; ++mi)
#pragma unroll
              for (int r = 0; r < 16; ++r) dE[mi][r] += dEsum[mi][r];
          }
          scatter_triplane_grad<MI>(tc, hi, x, y, z, dE, valid);
        }
        if constexpr (HASH == 1) {
          f32x16 dE[MI];
          layer_dgrad<MI, MH>(sm + LY::w_off(0), lane, dY, dE);
          if (add_enc || CAT) {          // what reached the encoding through the skip connections of the later layers
#pragma unroll
            for (int r = 0; r < 16; ++r) dE[0][r] += dEsum[0][r];
          }
          // hand dL/dE to k_hash_grad (level-major, coalesced); table scatter happens there in LDS
          if (valid) {
            const int64_t g = (int64_t)f * a.P + n, NP = (int64_t)a.F * a.P;
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
              for (int p = 0; p < 2; ++p) {
                const int level = 4 * q + 2 * hi + p;
                if (level < hc.nlev) a.hash_dE[level * NP + g] = make_float2(dE[0][4 * q + 2 * p], dE[0][4 * q + 2 * p + 1]);
              }
            if (hi == 0 && !a.hash_xyz_ready) a.hash_xyz[g] = make_float4(x, y, z, 0.f);
          }
        }
        if (ENC_GRAD) {
          f32x16 dE[MI];
          layer_dgrad<MI, MH>(sm + LY::w_off(0), lane, dY, dE);
#pragma unroll
          for (int mi = 0; mi < MI; ++mi)
#pragma unroll
            for (int r = 0; r < 16; ++r) dE[mi][r] = (dE[mi][r] + dEsum[mi][r]) * dEa[mi][r];
          WAVE_SYNC();
          store_tile<MI>(bufD, BL::STR_D, lane, dE);
          WAVE_SYNC();
          outer_accum<MI, 3>(bufD, BL::STR_D, pbuf, lane, dwf);
        }
      } else {
        layer_wgrad<MH, MH>(bufD, BL::STR_D, wl + BL::x_off(l), BL::STR_H, lane, accH[l - 1]);
        f32x16 dX[MH], Xl[MH];
        if constexpr (CAT) {
          // the layer's input is cat(hidden, encoding): weight gradient of the encoding columns from the staged
          // encoding tile, data gradient over all MH + MI input tiles, the encoding part joins dEsum
          layer_wgrad<MH, MI>(bufD, BL::STR_D, wl + BL::x_off(0), BL::STR_E, lane, accE[l - 1]);
          f32x16 dXc[MC];
          layer_dgrad<MC, MH>(sm + LY::w_off(l), lane, dY, dXc);
#pragma unroll
          for (int mi = 0; mi < MH; ++mi) dX[mi] = dXc[mi];
#pragma unroll
          for (int mi = 0; mi < MI; ++mi)
#pragma unroll
            for (int r = 0; r < 16; ++r) dEsum[mi][r] += dXc[MH + mi][r];
        } else {
          layer_dgrad<MH, MH>(sm + LY::w_off(l), lane, dY, dX);
        }
        if (add_enc) {
          // dX = gradient w.r.t. in_l = relu(z_{l-1}) + [enc; 0]
#pragma unroll
          for (int mi = 0; mi < MH; ++mi)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
              if (mi < MI) dEsum[mi < MI ? mi : 0][r] += dX[mi][r];
              dY[mi][r] = ((rmask[l - 1] >> (16 * mi + r)) & 1u) ? dX[mi][r] : 0.f;
            }
        } else {
          load_tile<MH>(wl + BL::x_off(l), BL::STR_H, lane, Xl);
#pragma unroll
          for (int mi = 0; mi < MH; ++mi)
#pragma unroll
            for (int r = 0; r < 16; ++r) dY[mi][r] = (Xl[mi][r] > 0.f) ? dX[mi][r] : 0.f;
        }
      }
    }
  }

  // ---- epilogue: reduce the 4 waves' accumulators in LDS (fixed order), write one partial vector
  __syncthreads();
  float* red = sm + LY::TOTAL;
  int64_t enc_off, w_off[NGM_MAX_LAYERS + 1], b_off[NGM_MAX_LAYERS + 1];
  const int64_t ptot = ngm_param_offsets(&a.fc, &enc_off, w_off, b_off);
  for (int64_t p = threadIdx.x; p < ptot; p += NGM_BLOCK) red[p] = 0.f;
  __syncthreads();
  const int D = a.fc.dim_enc, H = a.fc.dim_hidden;
  // wave-level finalisation of the lane=feature accumulators
  if (MH == 1) {
#pragma unroll
    for (int l = 0; l < L; ++l) dbh[l] += __shfl_down(dbh[l], 32, 64);
#pragma unroll
    for (int c = 0; c < 4; ++c) dwo[c] += __shfl_down(dwo[c], 32, 64);
  }
  if (MI == 1) {
#pragma unroll
    for (int c = 0; c < 3; ++c) dwf[c] += __shfl_down(dwf[c], 32, 64);
    if constexpr (CAT) {
#pragma unroll
      for (int c = 0; c < 4; ++c) dwoE[c] += __shfl_down(dwoE[c], 32, 64);
    }
  }
  const int DC = CAT ? H + D : H;          // row length of W_l (l >= 1) and of W_out
#pragma unroll
  for (int c = 0; c < 4; ++c) dbo[c] = wave_sum(dbo[c]);
  for (int w = 0; w < NGM_WAVES_PER_BLOCK; ++w) {
    if (wave == w) {
      // hidden weight matrices
#pragma unroll
      for (int mo = 0; mo < MH; ++mo) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int o = 32 * mo + frow(r, hi);
          if (o < H) {
#pragma unroll
            for (int mi = 0; mi < MI; ++mi) {
              const int i = 32 * mi + j;
              if (i < D) red[w_off[0] + (int64_t)o * D + i] += acc0[mo][mi][r];
            }
#pragma unroll
            for (int l = 1; l < L; ++l) {
#pragma unroll
              for (int mi = 0; mi < MH; ++mi) {
                const int i = 32 * mi + j;
                if (i < H) red[w_off[l] + (int64_t)o * DC + i] += accH[l - 1][mo][mi][r];
              }
              if constexpr (CAT) {
#pragma unroll
                for (int mi = 0; mi < MI; ++mi) {
                  const int i = 32 * mi + j;
                  if (i < D) red[w_off[l] + (int64_t)o * DC + H + i] += accE[l - 1][mo][mi][r];
                }
              }
            }
          }
        }
      }
      // lane = feature quantities
      if (lane < 32 * MH && lane < H) {
#pragma unroll
        for (int l = 0; l < L; ++l) red[b_off[l] + lane] += dbh[l];
#pragma unroll
        for (int c = 0; c < 4; ++c) red[w_off[L] + (int64_t)c * DC + lane] += dwo[c];
      }
      if (CAT && lane < 32 * MI && lane < D) {
#pragma unroll
        for (int c = 0; c < 4; ++c) red[w_off[L] + (int64_t)c * DC + H + lane] += dwoE[c];
      }
      if (lane < 4) red[b_off[L] + lane] += (lane == 0 ? dbo[0] : lane == 1 ? dbo[1] : lane == 2 ? dbo[2] : dbo[3]);
      if (ENC_GRAD && a.fc.encoding == NGM_ENC_FOURIER) {
        const int n_raw = a.fc.raw_coords ? 3 : 0;
        if (lane < 32 * MI && lane < D && lane >= n_raw) {
#pragma unroll
          for (int c = 0; c < 3; ++c) red[enc_off + (int64_t)(lane - n_raw) * 3 + c] += dwf[c];
        }
      }
    }
    __syncthreads();
  }
  float* dst = a.partials + (int64_t)blockIdx.x * a.p_pad;
  for (int64_t p = threadIdx.x; p < ptot; p += NGM_BLOCK) dst[p] = red[p];
}

// ------------------------------------------------------------------------------------------------
// deterministic reduction of the per-workgroup partial gradient vectors into the caller's tensors
// ------------------------------------------------------------------------------------------------
struct GradSeg { int64_t off, size; float* dst; int64_t stride; float* param; float* m; float* v; int64_t pstride; void* lp; int lp_dt; };
struct GradReduceK {
  int F, blocks_per_field, nseg;
  const float* partials;
  int64_t p_pad, ptot;
  GradSeg seg[2 * (NGM_MAX_LAYERS + 1) + 1];
  // fused sparse Adam (seg[k].param != NULL): torch.optim.Adam with L2-coupled weight decay, as k_adam_multi
  const int64_t* field_index; const int64_t* step_dev; int64_t step;
  float lr, beta1, beta2, eps, wd;
};

// 64 parameters x 4 interleaved quarter-sums of the per-workgroup partials per block (the quarters are
// combined in a fixed order -> deterministic); 4x the loads in flight of a one-thread-per-parameter loop.
__global__ void __launch_bounds__(256) k_grad_reduce(GradReduceK a) {
  __shared__ float part[4][64];
  const int f = blockIdx.y, q = threadIdx.x >> 6, l = threadIdx.x & 63;
  const int64_t p = (int64_t)blockIdx.x * 64 + l;
  // The kernel is a latency chain (a few KB per workgroup): everything that does not depend on the sums is issued FIRST --
  // the parameter / moment loads of the quarter that will apply the update, the bias corrections (two fp64 pow) -- and the
  // partial sums keep up to eight loads in flight per thread (they used to go four at a time, the Adam loads behind them).
  int seg = -1;
  int64_t so = 0;
  float pv = 0.f, m0 = 0.f, v0 = 0.f, lr_bc1 = 0.f, inv_sqrt_bc2 = 1.f;
  const bool adam = q == 0 && p < a.ptot;
  if (adam) {
    for (int k = 0; k < a.nseg; ++k)
      if (p >= a.seg[k].off && p < a.seg[k].off + a.seg[k].size) { seg = k; break; }
    if (seg >= 0 && a.seg[seg].param) {
      const int64_t row = a.field_index ? a.field_index[f] : f;
      so = row * a.seg[seg].pstride + (p - a.seg[seg].off);
      pv = a.seg[seg].param[so]; m0 = a.seg[seg].m[so]; v0 = a.seg[seg].v[so];
    }
  }
  // workgroup b of the backward kernel handled field b % F; quarter q sums blocks q, q + 4, ... in that order
  const float* src = a.partials + (int64_t)f * a.p_pad + p;
  const int64_t cs = (int64_t)a.F * a.p_pad;
  int c = q;
  // the first eight partials travel while the bias corrections are computed: a wave issues in order, and the two fp64 pow
  // (~2 us of instructions) in front of these loads used to be a second latency in the chain
  float v8[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  const bool pre = p < a.ptot && c + 28 < a.blocks_per_field;
  if (pre) {
#pragma unroll
    for (int j = 0; j < 8; ++j) v8[j] = src[(c + 4 * j) * cs];
  }
  if (adam && seg >= 0 && a.seg[seg].param) {
    const double step = (double)(a.step_dev ? *a.step_dev : a.step);
    lr_bc1 = (float)((double)a.lr / (1.0 - pow((double)a.beta1, step)));
    inv_sqrt_bc2 = (float)(1.0 / sqrt(1.0 - pow((double)a.beta2, step)));
  }
  float s = 0.f;
  if (pre) {
#pragma unroll
    for (int j = 0; j < 8; ++j) s += v8[j];
    c += 32;
  }
  if (p < a.ptot) {
#pragma unroll 1
    for (; c + 28 < a.blocks_per_field; c += 32) {
      float v[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = src[(c + 4 * j) * cs];
#pragma unroll
      for (int j = 0; j < 8; ++j) s += v[j];
    }
#pragma unroll 1
    for (; c + 12 < a.blocks_per_field; c += 16) {
      const float v0_ = src[c * cs], v1 = src[(c + 4) * cs], v2 = src[(c + 8) * cs], v3 = src[(c + 12) * cs];
      s += v0_; s += v1; s += v2; s += v3;
    }
    for (; c < a.blocks_per_field; c += 4) s += src[c * cs];
  }
  part[q][l] = s;
  __syncthreads();
  if (q != 0 || p >= a.ptot || seg < 0) return;
  s = ((part[0][l] + part[1][l]) + part[2][l]) + part[3][l];
  const int64_t i = p - a.seg[seg].off;
  if (a.seg[seg].dst) a.seg[seg].dst[(int64_t)f * a.seg[seg].stride + i] = s;
  if (a.seg[seg].param) {
    // the update of rm.py:1183-1221 on row field_index[f], straight from the reduced gradient (no second launch, no
    // gradient round trip); same arithmetic as k_adam_multi
    const float g = s + a.wd * pv;
    const float mn = a.beta1 * m0 + (1.0f - a.beta1) * g;
    const float vn = a.beta2 * v0 + (1.0f - a.beta2) * g * g;
    a.seg[seg].m[so] = mn; a.seg[seg].v[so] = vn;
    const float pn = pv - lr_bc1 * (mn / (sqrtf(vn) * inv_sqrt_bc2 + a.eps));
    a.seg[seg].param[so] = pn;
    if (a.seg[seg].lp) ngm_stp(a.seg[seg].lp, so, pn, a.seg[seg].lp_dt);      // the reduced-precision copy the kernels read
  }
}

// Few partials per field (<= 8: 32 and more active fields on the chip): the four-quarter split above leaves each quarter
// two loads and launches four times the workgroups the work needs (32 fields x 137 = 4 384 of them for the 64 + 2 x 64
// network: 18.7 us, more than twice the 8-field launch).  Here a thread owns ONE parameter: its <= 8 partials in flight together,
// summed in exactly the order of k_grad_reduce -- quarter i = p_i + p_(i+4), then ((q0 + q1) + q2) + q3 -- so the bits are the same.
// One parameter p of field f: its partials summed in exactly k_grad_reduce's order -- quarter i = partials i, i + 4, i + 8, ... in
// that order, then ((q0 + q1) + q2) + q3 -- and the sparse Adam update.  <= 8 partials: all in flight together.
__device__ __forceinline__ void grad_reduce_one(const GradReduceK& a, int f, int64_t p) {
  if (p >= a.ptot) return;
  int seg = -1;
  for (int k = 0; k < a.nseg; ++k)
    if (p >= a.seg[k].off && p < a.seg[k].off + a.seg[k].size) { seg = k; break; }
  int64_t so = 0;
  float pv = 0.f, m0 = 0.f, v0 = 0.f;
  const bool adam = seg >= 0 && a.seg[seg].param;
  if (adam) {
    const int64_t row = a.field_index ? a.field_index[f] : f;
    so = row * a.seg[seg].pstride + (p - a.seg[seg].off);
    pv = a.seg[seg].param[so]; m0 = a.seg[seg].m[so]; v0 = a.seg[seg].v[so];
  }
  const float* src = a.partials + (int64_t)f * a.p_pad + p;
  const int64_t cs = (int64_t)a.F * a.p_pad;
  float q[4] = {0.f, 0.f, 0.f, 0.f};
  if (a.blocks_per_field <= 8) {
    float v[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) v[c] = (c < a.blocks_per_field) ? src[c * cs] : 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      float s = 0.f;
      if (i < a.blocks_per_field) s += v[i];
      if (i + 4 < a.blocks_per_field) s += v[i + 4];
      q[i] = s;
    }
  } else {
    for (int c0 = 0; c0 < a.blocks_per_field; c0 += 16) {          // 16 loads in flight; every quarter keeps its own order
      float v[16];
#pragma unroll
      for (int u = 0; u < 16; ++u) v[u] = (c0 + u < a.blocks_per_field) ? src[(int64_t)(c0 + u) * cs] : 0.f;
#pragma unroll
      for (int u = 0; u < 16; ++u)
        if (c0 + u < a.blocks_per_field) q[u & 3] += v[u];
    }
  }
  float lr_bc1 = 0.f, inv_sqrt_bc2 = 1.f;
  if (adam) {
    const double step = (double)(a.step_dev ? *a.step_dev : a.step);
    lr_bc1 = (float)((double)a.lr / (1.0 - pow((double)a.beta1, step)));
    inv_sqrt_bc2 = (float)(1.0 / sqrt(1.0 - pow((double)a.beta2, step)));
  }
  const float s = ((q[0] + q[1]) + q[2]) + q[3];
  if (seg < 0) return;
  const int64_t i = p - a.seg[seg].off;
  if (a.seg[seg].dst) a.seg[seg].dst[(int64_t)f * a.seg[seg].stride + i] = s;
  if (adam) {
    const float g = s + a.wd * pv;
    const float mn = a.beta1 * m0 + (1.0f - a.beta1) * g;
    const float vn = a.beta2 * v0 + (1.0f - a.beta2) * g * g;
    a.seg[seg].m[so] = mn; a.seg[seg].v[so] = vn;
    const float pn = pv - lr_bc1 * (mn / (sqrtf(vn) * inv_sqrt_bc2 + a.eps));
    a.seg[seg].param[so] = pn;
    if (a.seg[seg].lp) ngm_stp(a.seg[seg].lp, so, pn, a.seg[seg].lp_dt);
  }
}
__global__ void __launch_bounds__(256) k_grad_reduce_flat(GradReduceK a) {
  grad_reduce_one(a, blockIdx.y, (int64_t)blockIdx.x * 256 + threadIdx.x);
}

static int build_grad_reduce(const GradReduceArgs& g, GradReduceK& k) {
  k.F = g.F; k.blocks_per_field = g.blocks_per_field; k.partials = g.partials; k.p_pad = g.p_pad;
  int64_t enc_off, w_off[NGM_MAX_LAYERS + 1], b_off[NGM_MAX_LAYERS + 1];
  k.ptot = ngm_param_offsets(&g.fc, &enc_off, w_off, b_off);
  int n = 0;
  if (g.fc.encoding == NGM_ENC_FOURIER) {
    const int64_t sz = (int64_t)(g.fc.raw_coords ? g.fc.dim_enc - 3 : g.fc.dim_enc) * 3;
    k.seg[n++] = GradSeg{enc_off, sz, g.gr.enc_w, g.gr.enc_w_stride, nullptr, nullptr, nullptr, 0, nullptr, 0};
  }
  for (int l = 0; l <= g.fc.num_layers; ++l) {
    const int din = (l == 0) ? g.fc.dim_enc : g.fc.dim_hidden + (g.fc.skip_mode == NGM_SKIP_CONCAT ? g.fc.dim_enc : 0);
    const int dout = (l == g.fc.num_layers) ? g.fc.dim_out : g.fc.dim_hidden;
    k.seg[n++] = GradSeg{w_off[l], (int64_t)din * dout, g.gr.w[l], g.gr.w_stride[l], nullptr, nullptr, nullptr, 0, nullptr, 0};
    k.seg[n++] = GradSeg{b_off[l], (int64_t)dout, g.gr.b[l], g.gr.b_stride[l], nullptr, nullptr, nullptr, 0, nullptr, 0};
  }
  k.nseg = n;
  k.field_index = nullptr; k.step_dev = nullptr; k.step = 1; k.lr = k.beta1 = k.beta2 = k.eps = k.wd = 0.f;
  if (g.adam.tensors) {
    if (g.adam.num != n) return NGM_E_INVALID;
    for (int i = 0; i < n; ++i) {
      const ngm_adam_tensor& t = g.adam.tensors[i];
      if (t.numel != k.seg[i].size || !t.param || !t.exp_avg || !t.exp_avg_sq) return NGM_E_INVALID;
      k.seg[i].param = t.param; k.seg[i].m = t.exp_avg; k.seg[i].v = t.exp_avg_sq; k.seg[i].pstride = t.stride;
      k.seg[i].lp = t.param_lp; k.seg[i].lp_dt = t.lp_dtype;
    }
    k.field_index = g.adam.field_index; k.step_dev = g.adam.step_dev; k.step = g.adam.step;
    k.lr = g.adam.lr; k.beta1 = g.adam.beta1; k.beta2 = g.adam.beta2; k.eps = g.adam.eps; k.wd = g.adam.wd;
  }
  return 0;
}
int ngm_launch_grad_reduce(const GradReduceArgs& g, hipStream_t st) {
  NgmProfScope prof_(NGM_K_GRAD_REDUCE, st);
  GradReduceK k;
  const int rc = build_grad_reduce(g, k);
  if (rc) return rc;
  if (k.blocks_per_field <= 8) {
    dim3 grid((unsigned)((k.ptot + 255) / 256), (unsigned)g.F);
    hipLaunchKernelGGL(k_grad_reduce_flat, grid, dim3(256), 0, st, k);
    return 0;
  }
  dim3 grid((unsigned)((k.ptot + 63) / 64), (unsigned)g.F);
  hipLaunchKernelGGL(k_grad_reduce, grid, dim3(256), 0, st, k);
  return 0;
}

// ------------------------------------------------------------------------------------------------
template <int MI, int MH, int L>
static int launch_bwd(const FieldBwdArgs& a, int blocks, hipStream_t st) {
#define NGM_LB(NC, EG, HS, CT)                                                                                          \
  do {                                                                                                                  \
    const size_t lds = BwdLds<MI, MH, L, CT>::TOTAL * sizeof(float);                                                    \
    if (lds > 160 * 1024) return NGM_E_UNSUPPORTED;                                                                      \
    (void)hipFuncSetAttribute((const void*)k_field_bwd<MI, MH, L, NC, EG, HS, CT>, hipFuncAttributeMaxDynamicSharedMemorySize, \
                              (int)lds);                                                                                \
    hipLaunchKernelGGL((k_field_bwd<MI, MH, L, NC, EG, HS, CT>), dim3(blocks), dim3(NGM_BLOCK), lds, st, a);            \
  } while (0)
  const bool cat = a.fc.skip_mode == NGM_SKIP_CONCAT;      // every encoding (models.py:159-161 is encoding-agnostic)
  if (a.fc.encoding == NGM_ENC_PERMUTO) {
    if constexpr (MI == 1) { if (cat) NGM_LB(false, false, 1, true); else NGM_LB(false, false, 1, false); }
    else return NGM_E_UNSUPPORTED;
  } else if (a.fc.encoding == NGM_ENC_TRIPLANE) {
    if (!a.tri_acc) return NGM_E_UNSUPPORTED;
    if (cat) NGM_LB(false, false, 2, true); else NGM_LB(false, false, 2, false);
  } else if (a.fc.encoding == NGM_ENC_FOURIER) { if (cat) NGM_LB(false, true, 0, true); else NGM_LB(false, true, 0, false); }
  else if (a.fc.encoding == NGM_ENC_NERF) { if (cat) NGM_LB(true, false, 0, true); else NGM_LB(true, false, 0, false); }
  else { if (cat) NGM_LB(false, false, 0, true); else NGM_LB(false, false, 0, false); }
#undef NGM_LB
  return 0;
}

// triplane: Q23.40 accumulators -> the caller's gradient tensor (fully overwritten)
__global__ void k_tri_finish(const long long* acc, int64_t numel, float* grad, int64_t gstride) {
  const int f = blockIdx.y;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < numel; i += (int64_t)gridDim.x * blockDim.x)
    grad[(int64_t)f * gstride + i] = (float)((double)acc[(int64_t)f * numel + i] * (1.0 / 1099511627776.0));
}
int ngm_launch_tri_finish(const FieldBwdArgs& a, hipStream_t st) {
  if (!a.tri_acc || !a.planes_grad) return NGM_E_INVALID;
  const int bx = (int)std::min<int64_t>((a.tri_numel + 255) / 256, 1024);
  hipLaunchKernelGGL(k_tri_finish, dim3(bx, a.F), dim3(256), 0, st, a.tri_acc, a.tri_numel, a.planes_grad, a.planes_grad_stride);
  return 0;
}

int ngm_launch_field_bwd(const FieldBwdArgs& a, int blocks, hipStream_t st) {
  NgmProfScope prof_(NGM_K_FIELD_BWD, st);
  const FieldShape s = field_shape(&a.fc);
  if (s.MI == 2 && s.MH == 2 && s.L == 2) return launch_bwd<2, 2, 2>(a, blocks, st);
#ifndef NGM_FAST_BUILD
  if (s.MI == 2 && s.MH == 2 && s.L == 1) return launch_bwd<2, 2, 1>(a, blocks, st);
  if (s.MI == 1 && s.MH == 1 && s.L == 1) return launch_bwd<1, 1, 1>(a, blocks, st);
  if (s.MI == 1 && s.MH == 1 && s.L == 2) return launch_bwd<1, 1, 2>(a, blocks, st);
#endif
  return NGM_E_UNSUPPORTED;
}

// ------------------------------------------------------------------------------------------------
// Permutohedral table gradient: one workgroup per (field, level, sample chunk).  The level's table
// gradient (T x 2 floats) lives in LDS; every sample adds its 4 vertices with LDS atomics; the table
// is then flushed once with global float atomics (chunks <= 8 adds per entry instead of one global
// atomic per sample-vertex).
// ------------------------------------------------------------------------------------------------
// float -> Q23.40 two's complement with fp32 / integer instructions only (the double-precision route was 10 f64
// instructions per value): |v| * 2^9 = hi + rem, hi = floor, rem in [0,1), both exact for a non-negative operand;
// magnitude = hi * 2^31 + trunc(rem * 2^31), then the sign.  |v| < 2^22; error < 2^-40 (truncation toward zero).
__device__ __forceinline__ unsigned long long to_fix(float v) {
  // signed throughout (truncation toward zero in both stages = truncation of the magnitude: the same bits as the
  // sign-magnitude form this replaces, without its abs / 64-bit negate): v 2^9 = hi + rem, hi = trunc (exact), |rem| < 1
  // with the sign of v; value = hi 2^31 + trunc(rem 2^31) as a 64-bit integer.
  const float s = v * 512.0f;
  const float tr = truncf(s);
  const int hi = (int)tr;                                               // |hi| < 2^31
  const int lo = (int)((s - tr) * 2147483648.0f);                       // |lo| < 2^31, sign of v
  const long long r = (long long)hi * 2147483648ll + (long long)lo;
  return (unsigned long long)r;
}

struct HashGradArgs {
  ngm_field_cfg fc;
  ngm_params pr;
  int F; int64_t P; int chunks; int64_t per_chunk;
  const float2* dE; const float4* xyz;
  float* gtab; int64_t gstride;
  float* part;     // [F][L][chunks][2T] per-workgroup partial tables (plain stores, reduced in fixed order)
  // optional fused sparse Adam on the tables (ad_param != NULL), as k_adam_multi
  float* ad_param; float* ad_m; float* ad_v; int64_t ad_stride; void* ad_lp; int ad_lp_dt;
  const int64_t* ad_field_index; const int64_t* ad_step_dev; int64_t ad_step;
  float ad_lr, ad_beta1, ad_beta2, ad_eps, ad_wd;
  // Round 6: the MLP's gradient reduction + Adam (k_grad_reduce's work, a few KB per field) rides along as extra workgroups
  // behind the hash_blocks table workgroups: it depends on the MLP backward only, like this kernel, and as a launch of its own
  // it was 8 us of latency chain + boundary in a 195 us iteration.  mlp_bx = 0: none.
  int hash_blocks, mlp_bx;
  GradReduceK mlp;
};

// FLT = false (NGM_HASH_ATOMICS_EXACT, the default): Q23.40 fixed point in LDS -- integer LDS atomics run at full bank rate
// and make the table gradient order-independent -> bitwise reproducible.  FLT = true (NGM_HASH_ATOMICS_FLOAT, opt-in): plain
// fp32 LDS atomics (ds_add_f32), what the reference's CUDA package does with global float atomics: no fixed-point conversion
// (88 of ~300 vector instructions per sample-level), half the LDS per workgroup (32 KB per level: four workgroups per CU
// instead of two), the sums in whatever order the adds land -- not reproducible run to run.
template <bool FLT> struct HashAcc { using T = unsigned long long; };
template <> struct HashAcc<true> { using T = float; };
template <bool FLT>
__device__ __forceinline__ float hash_acc_value(typename HashAcc<FLT>::T v) {
  if constexpr (FLT) return v;
  else return (float)((double)(long long)v * (1.0 / 1099511627776.0));
}
// NL = levels per workgroup.  NL = 2 (round 6 experiment, NGM_HASH_PAIR=1; NOT the default: 2 us slower): a workgroup of 1024
// threads owns the level PAIR (l, L - 1 - l) of its (field, chunk) with both tables in LDS (2 x 64 KB): the position of a
// sample is read once for two levels (384 -> 256 B per sample over the 16 levels) and a thread carries two independent
// simplex searches.  Measured equal-to-slower than one level per workgroup: the loop is not bound by its loads.
template <bool FLT, int NL, int NT = 512 * NL>
__global__ __launch_bounds__(NT) void k_hash_grad(HashGradArgs a) {
  using acc_t = typename HashAcc<FLT>::T;
  extern __shared__ __attribute__((aligned(16))) unsigned char tab_raw[];
  acc_t* tab = reinterpret_cast<acc_t*>(tab_raw);
  if ((int)blockIdx.x >= a.hash_blocks) {          // the MLP reduction riding along: one thread per parameter (grad_reduce_one)
    const int id = (int)blockIdx.x - a.hash_blocks;
    grad_reduce_one(a.mlp, id / a.mlp_bx, (int64_t)(id % a.mlp_bx) * blockDim.x + threadIdx.x);
    return;
  }
  int chunk, f, level[NL];
  {
    const int nlev = a.fc.nr_levels, total = a.hash_blocks, id = blockIdx.x;
    if constexpr (NL == 2) {
      chunk = id % a.chunks;
      const int r = id / a.chunks, lh = r % (nlev / 2);
      f = r / (nlev / 2);
      level[0] = lh; level[1] = nlev - 1 - lh;
    } else if ((nlev & 1) == 0) {
      // 1-D grid, decoded so that blocks id and id + total / 2 -- which share a CU when total / 2 = the number of CUs, two
      // blocks being resident per CU -- work on levels l and L - 1 - l: a coarse level (long runs: the segmented scans run)
      // costs ~1.5x a fine one, and two coarse blocks on one CU set the kernel's time
      const int half = id >= total / 2, j = id - half * (total / 2);
      chunk = j % a.chunks;
      const int r = j / a.chunks, lh = r % (nlev / 2);
      f = r / (nlev / 2);
      level[0] = half ? nlev - 1 - lh : lh;
    } else {
      chunk = id % a.chunks;
      level[0] = (id / a.chunks) % nlev;
      f = id / (a.chunks * nlev);
    }
  }
  const int T = 1 << a.fc.log2_hashmap_size;
  const int64_t row = a.pr.field_index ? a.pr.field_index[f] : f;
  for (int i = threadIdx.x; i < NL * 2 * T; i += blockDim.x) tab[i] = acc_t(0);
  // Adam's bias corrections (two double-precision pow) once per workgroup, up front, by one thread -- not by every thread in
  // the epilogue; read back after the barrier that ends the sample loop
  __shared__ float adam_c[2];
  if (a.chunks == 1 && a.ad_param && threadIdx.x == 0) {
    const double step = (double)(a.ad_step_dev ? *a.ad_step_dev : a.ad_step);
    adam_c[0] = (float)((double)a.ad_lr / (1.0 - pow((double)a.ad_beta1, step)));
    adam_c[1] = (float)(1.0 / sqrt(1.0 - pow((double)a.ad_beta2, step)));
  }
  float lp[NL][8];
#pragma unroll
  for (int q = 0; q < NL; ++q) {
    const float* hs = a.pr.shift + row * a.pr.shift_stride + 3 * level[q];
    lp[q][0] = a.fc.level_scale[3 * level[q]]; lp[q][1] = a.fc.level_scale[3 * level[q] + 1]; lp[q][2] = a.fc.level_scale[3 * level[q] + 2];
    lp[q][3] = 0.f; lp[q][4] = hs[0]; lp[q][5] = hs[1]; lp[q][6] = hs[2]; lp[q][7] = 0.f;
  }
  __syncthreads();
  const int64_t NP = (int64_t)a.F * a.P;
  const int64_t beg = (int64_t)chunk * a.per_chunk, end = min(a.P, beg + a.per_chunk);
  const uint32_t mask = (uint32_t)T - 1u;
  const int lane = threadIdx.x & 63;
  // the next iteration's position and gradients travel while this one is worked on (few waves per SIMD: a load issued
  // where it is used exposes the whole HBM latency once per iteration)
  float4 p_nx = make_float4(0.f, 0.f, 0.f, 0.f);
  float2 d_nx[NL];
#pragma unroll
  for (int q = 0; q < NL; ++q) d_nx[q] = make_float2(0.f, 0.f);
  if (beg + threadIdx.x < end) {
    const int64_t g = (int64_t)f * a.P + beg + threadIdx.x;
    p_nx = a.xyz[g];
#pragma unroll
    for (int q = 0; q < NL; ++q) d_nx[q] = a.dE[level[q] * NP + g];
  }
  for (int64_t s0 = beg; s0 < end; s0 += blockDim.x) {      // trip count uniform across the wave (shuffles inside)
    const int64_t s = s0 + threadIdx.x;
    const bool valid = s < end;
    const float4 p = p_nx;
    float2 dq[NL];
#pragma unroll
    for (int q = 0; q < NL; ++q) dq[q] = valid ? d_nx[q] : make_float2(0.f, 0.f);
    if (s + blockDim.x < end) {
      const int64_t g = (int64_t)f * a.P + s + blockDim.x;
      p_nx = a.xyz[g];
#pragma unroll
      for (int q = 0; q < NL; ++q) d_nx[q] = a.dE[level[q] * NP + g];
    }
    uint32_t idxq[NL][4]; float bwq[NL][4];
#pragma unroll
    for (int q = 0; q < NL; ++q) {
#pragma unroll
      for (int r = 0; r < 4; ++r) { idxq[q][r] = 0u; bwq[q][r] = 0.f; }
      if (valid) {
#ifdef NGM_ABLH_NOSIMPLEX   // timing ablations of k_hash_grad (results meaningless when defined)
        idxq[q][0] = (uint32_t)(p.x * 1000.f) & mask; idxq[q][1] = (idxq[q][0] + 1) & mask; idxq[q][2] = (idxq[q][0] + 2) & mask; idxq[q][3] = (idxq[q][0] + 3) & mask;
        bwq[q][0] = p.x; bwq[q][1] = p.y; bwq[q][2] = p.z; bwq[q][3] = 1.f - p.x;
#else
        permuto_simplex(p.x, p.y, p.z, lp[q], mask, idxq[q], bwq[q]);
#endif
      }
    }
#pragma unroll
    for (int q = 0; q < NL; ++q) {
      acc_t* tq = tab + (size_t)q * 2 * T;
      const uint32_t (&idx)[4] = idxq[q];
      const float (&bw)[4] = bwq[q];
      const float2 d = dq[q];
#ifdef NGM_ABLH_NOSCATTER
      if (valid) { tq[threadIdx.x] += acc_t(idx[0] + idx[1] + idx[2] + idx[3]) + acc_t(to_fix(d.x * bw[0] + d.y * bw[1] + bw[2] + bw[3])); }
      continue;
#endif
      // consecutive lanes = consecutive samples of a ray: at coarse levels they sit in the same simplex in long
      // runs, which would serialise the LDS atomic unit (measured 163 LDS cycles per ds_add_f32).  Reduce each run
      // inside the wave first (segmented scans; a run = lanes whose four vertices all repeat the previous lane's, so
      // one run structure serves the 8 sums) and let its last lane issue the atomics.
      bool same = valid && lane > 0;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const uint32_t key = valid ? idx[r] : 0xffffffffu;
        const uint32_t prev = (uint32_t)__builtin_amdgcn_update_dpp((int)0xfffffffe, (int)key, NGM_DPP_WAVE_SHR1, 0xf, 0xf, false);
        same = same && (key == prev);
      }
      const unsigned long long hm = __ballot(!same);
      float v[8];
#pragma unroll
      for (int r = 0; r < 4; ++r) { v[2 * r] = d.x * bw[r]; v[2 * r + 1] = d.y * bw[r]; }
      bool issue = valid;
#ifdef NGM_ABLH_NOMERGE      // timing ablation: every lane issues its own atomics
      if (false) {
#else
      if (__popcll(hm) <= 40) {                                   // wave-uniform
#endif
        const unsigned long long below = hm & ((2ull << lane) - 1ull);
        const int k = lane - (63 - __clzll(below));
        seg_scan_add_n<8>(v, k, lane);
        issue = valid && ((lane == 63) || ((hm >> (lane + 1)) & 1ull));
      }
      if (issue) {
        if constexpr (FLT) {
          // no-return fp32 LDS atomics: ds_add_f32 (checked in the disassembly: no CAS loop)
#pragma unroll
          for (int r = 0; r < 4; ++r) { atomicAdd(&tq[idx[r]], v[2 * r]); atomicAdd(&tq[T + idx[r]], v[2 * r + 1]); }
        } else {
#pragma unroll
          for (int r = 0; r < 4; ++r) { atomicAdd(&tq[idx[r]], to_fix(v[2 * r])); atomicAdd(&tq[T + idx[r]], to_fix(v[2 * r + 1])); }
        }
      }
    }
  }
  __syncthreads();
  // LDS holds one plane per feature (8-byte stride: the 64 lanes of an atomic spread over 32 bank pairs; interleaved, 16-byte
  // entries reach only 16); tables are written in the parameter layout [entry][feature]
#pragma unroll
  for (int q = 0; q < NL; ++q) {
    const acc_t* tq = tab + (size_t)q * 2 * T;
    const int lv = level[q];
    if (a.chunks == 1) {
      // This workgroup saw EVERY sample of its (field, level): its LDS table is the gradient.  It is written where
      // k_hash_reduce would have put it, and the sparse Adam of that table follows right here -- no partial table through HBM
      // (2 x 32 KB per level and field), no k_hash_reduce launch.  (The reference's default iteration: 32 fields x 16 levels =
      // one chunk per level; the M1 batch has 4 chunks per level and keeps the reduction kernel.)
      float* gdst = a.gtab + (int64_t)f * a.gstride + (int64_t)lv * T * 2;
      float lr_bc1 = 0.f, inv_sqrt_bc2 = 1.f;
      int64_t prow = 0;
      if (a.ad_param) {
        lr_bc1 = adam_c[0]; inv_sqrt_bc2 = adam_c[1];
        prow = (a.ad_field_index ? a.ad_field_index[f] : f) * a.ad_stride + (int64_t)lv * T * 2;
      }
      for (int i4 = threadIdx.x; i4 < T / 2; i4 += blockDim.x) {            // float4 = two entries x two features
        const int e0 = 2 * i4;
        float4 s4;
        s4.x = hash_acc_value<FLT>(tq[e0]);
        s4.y = hash_acc_value<FLT>(tq[T + e0]);
        s4.z = hash_acc_value<FLT>(tq[e0 + 1]);
        s4.w = hash_acc_value<FLT>(tq[T + e0 + 1]);
        reinterpret_cast<float4*>(gdst)[i4] = s4;
        if (a.ad_param) {                                                     // same arithmetic as k_hash_reduce / k_adam_multi
          const int64_t o4 = prow / 4 + i4;                                   // tables are 16-byte aligned rows (checked by the launcher)
          float4 p = reinterpret_cast<float4*>(a.ad_param)[o4], m = reinterpret_cast<float4*>(a.ad_m)[o4],
                 v = reinterpret_cast<float4*>(a.ad_v)[o4];
          float* pp = &p.x; float* pm = &m.x; float* pv = &v.x; const float* pg = &s4.x;
#pragma unroll
          for (int c = 0; c < 4; ++c) {
            const float g = pg[c] + a.ad_wd * pp[c];
            const float mn = a.ad_beta1 * pm[c] + (1.0f - a.ad_beta1) * g;
            const float vn = a.ad_beta2 * pv[c] + (1.0f - a.ad_beta2) * g * g;
            pm[c] = mn; pv[c] = vn;
            pp[c] = pp[c] - lr_bc1 * (mn / (sqrtf(vn) * inv_sqrt_bc2 + a.ad_eps));
          }
          reinterpret_cast<float4*>(a.ad_m)[o4] = m; reinterpret_cast<float4*>(a.ad_v)[o4] = v;
          reinterpret_cast<float4*>(a.ad_param)[o4] = p;      // (written through: no measurable difference, round 6)
          if (a.ad_lp) {
#pragma unroll
            for (int c = 0; c < 4; ++c) ngm_stp(a.ad_lp, 4 * o4 + c, pp[c], a.ad_lp_dt);
          }
        }
      }
    } else {
      float* dst = a.part + (((int64_t)f * a.fc.nr_levels + lv) * a.chunks + chunk) * 2 * T;
      for (int i = threadIdx.x; i < 2 * T; i += blockDim.x) dst[i] = hash_acc_value<FLT>(tq[(i & 1) * T + (i >> 1)]);
    }
  }
}

// gtab[f][level][i] = sum over chunks (fixed order): overwrites -> no zero-fill of the gradient needed
__global__ void k_hash_reduce(HashGradArgs a) {
  const int T = 1 << a.fc.log2_hashmap_size;
  const int level = blockIdx.y, f = blockIdx.z;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;          // float4 index
  if (i >= T / 2) return;
  const float4* src = reinterpret_cast<const float4*>(a.part + ((int64_t)f * a.fc.nr_levels + level) * a.chunks * 2 * T);
  float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int c = 0; c < a.chunks; ++c) {
    const float4 v = src[(int64_t)c * (T / 2) + i];
    s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
  }
  reinterpret_cast<float4*>(a.gtab + (int64_t)f * a.gstride + (int64_t)level * T * 2)[i] = s;
  if (a.ad_param) {
    const double step = (double)(a.ad_step_dev ? *a.ad_step_dev : a.ad_step);
    const float lr_bc1 = (float)((double)a.ad_lr / (1.0 - pow((double)a.ad_beta1, step)));
    const float inv_sqrt_bc2 = (float)(1.0 / sqrt(1.0 - pow((double)a.ad_beta2, step)));
    const int64_t row = a.ad_field_index ? a.ad_field_index[f] : f;
    const int64_t o4 = (row * a.ad_stride + (int64_t)level * T * 2) / 4 + i;       // tables are 16-byte aligned rows
    float4 p = reinterpret_cast<float4*>(a.ad_param)[o4], m = reinterpret_cast<float4*>(a.ad_m)[o4],
           v = reinterpret_cast<float4*>(a.ad_v)[o4];
    float* pp = &p.x; float* pm = &m.x; float* pv = &v.x; const float* pg = &s.x;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const float g = pg[c] + a.ad_wd * pp[c];
      const float mn = a.ad_beta1 * pm[c] + (1.0f - a.ad_beta1) * g;
      const float vn = a.ad_beta2 * pv[c] + (1.0f - a.ad_beta2) * g * g;
      pm[c] = mn; pv[c] = vn;
      pp[c] = pp[c] - lr_bc1 * (mn / (sqrtf(vn) * inv_sqrt_bc2 + a.ad_eps));
    }
    reinterpret_cast<float4*>(a.ad_m)[o4] = m; reinterpret_cast<float4*>(a.ad_v)[o4] = v;
    reinterpret_cast<float4*>(a.ad_param)[o4] = p;
    if (a.ad_lp) {
#pragma unroll
      for (int c = 0; c < 4; ++c) ngm_stp(a.ad_lp, 4 * o4 + c, pp[c], a.ad_lp_dt);
    }
  }
}

int ngm_launch_hash_grad(const FieldBwdArgs& fb, hipStream_t st, bool* adam_applied, const GradReduceArgs* mlp_reduce,
                         bool* mlp_reduced) {
  HashGradArgs a;
  a.hash_blocks = 0; a.mlp_bx = 0;
  memset(&a.mlp, 0, sizeof(a.mlp));
  if (mlp_reduced) *mlp_reduced = false;
  a.fc = fb.fc; a.pr = fb.pr; a.F = fb.F; a.P = fb.P; a.dE = fb.hash_dE; a.xyz = fb.hash_xyz;
  a.gtab = fb.lattice_grad; a.gstride = fb.lattice_grad_stride;
  const int T = 1 << fb.fc.log2_hashmap_size;
  const bool flt = fb.fc.hash_grad_atomics == NGM_HASH_ATOMICS_FLOAT;
  const size_t lds1 = (size_t)2 * T * (flt ? sizeof(float) : sizeof(unsigned long long));
  if (lds1 > 150 * 1024) return NGM_E_UNSUPPORTED;
  // level pairs (k_hash_grad<., 2>): both tables of a pair in one workgroup's LDS.  NGM_HASH_PAIR=0: one level per workgroup
  // (the round-5 kernel; A/B inside one library)
  static const bool pair_on = getenv("NGM_HASH_PAIR") != nullptr && atoi(getenv("NGM_HASH_PAIR")) == 1;
  const bool pair = pair_on && (fb.fc.nr_levels & 1) == 0 && 2 * lds1 <= 150 * 1024;
  const size_t lds = pair ? 2 * lds1 : lds1;
  const int units = (int)fb.F * (pair ? fb.fc.nr_levels / 2 : fb.fc.nr_levels);      // (field, level) or (field, level pair)
  // One level per workgroup: 2 blocks per CU = exactly what is resident at a time (64 KB of LDS each), one round: with the
  // coarse / fine level pairing of the block decoding the CUs finish together (M1 hash batch, kernel time: 4 chunks 99 us,
  // 6: 100, 8: 102, 5: 115; before the pairing 4 chunks cost 118 -- two coarse levels on one CU).  Level pairs: one
  // 1024-thread workgroup per CU.
  int ncu = 256;
  {
    int dev = 0;
    hipDeviceProp_t prop;
    static int cached = 0;
    if (!cached && hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0)
      cached = prop.multiProcessorCount;
    if (cached) ncu = cached;
  }
  const int slots = pair ? ncu : 2 * ncu;
  int chunks = (int)((slots + (int64_t)units - 1) / (int64_t)units);
  const int64_t max_chunks = (fb.P + 4095) / 4096;
  if (chunks > max_chunks) chunks = (int)max_chunks;
  if (chunks < 1) chunks = 1;
  if (chunks > 8) chunks = 8;
  static const char* env_chunks = getenv("NGM_HASH_CHUNKS");      // experiment knob (1..8)
  if (env_chunks && atoi(env_chunks) >= 1 && atoi(env_chunks) <= 8 && atoi(env_chunks) <= max_chunks) chunks = atoi(env_chunks);
  a.chunks = chunks; a.per_chunk = (fb.P + chunks - 1) / chunks;
  a.part = fb.hash_part;
  a.ad_param = nullptr; a.ad_lp = nullptr; a.ad_lp_dt = 0;
  if (fb.lattice_adam.tensors) {
    const ngm_adam_tensor& t = fb.lattice_adam.tensors[0];
    const int64_t per = (int64_t)fb.fc.nr_levels * T * 2;
    // vector path only: rows and pointers 16-byte aligned (else the caller's separate Adam launch does it)
    if (t.numel == per && (t.stride & 3) == 0 &&
        ((reinterpret_cast<uintptr_t>(t.param) | reinterpret_cast<uintptr_t>(t.exp_avg) | reinterpret_cast<uintptr_t>(t.exp_avg_sq)) & 15) == 0) {
      a.ad_param = t.param; a.ad_m = t.exp_avg; a.ad_v = t.exp_avg_sq; a.ad_stride = t.stride;
      a.ad_lp = t.param_lp; a.ad_lp_dt = t.lp_dtype;
      a.ad_field_index = fb.lattice_adam.field_index; a.ad_step_dev = fb.lattice_adam.step_dev; a.ad_step = fb.lattice_adam.step;
      a.ad_lr = fb.lattice_adam.lr; a.ad_beta1 = fb.lattice_adam.beta1; a.ad_beta2 = fb.lattice_adam.beta2;
      a.ad_eps = fb.lattice_adam.eps; a.ad_wd = fb.lattice_adam.wd;
    }
  }
  a.hash_blocks = chunks * units;
  static const bool no_ride = getenv("NGM_NO_REDUCE_RIDE") != nullptr;          // developer A/B switch
  if (mlp_reduce && !no_ride) {
    if (build_grad_reduce(*mlp_reduce, a.mlp)) return NGM_E_INVALID;
    const int nt = (pair || (getenv("NGM_HASH_THREADS") != nullptr && atoi(getenv("NGM_HASH_THREADS")) == 1024)) ? 1024 : 512;
    a.mlp_bx = (int)((a.mlp.ptot + nt - 1) / nt);
    if (mlp_reduced) *mlp_reduced = true;
  }
  const int grid_x = a.hash_blocks + a.mlp_bx * (int)fb.F;
  {
    NgmProfScope prof_(NGM_K_HASH_GRAD, st);
#define NGM_HG(FLT_, NL_, NT_)                                                                                                  \
    do {                                                                                                                        \
      (void)hipFuncSetAttribute((const void*)k_hash_grad<FLT_, NL_, NT_>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
      hipLaunchKernelGGL((k_hash_grad<FLT_, NL_, NT_>), dim3(grid_x), dim3(NT_), lds, st, a);                                    \
    } while (0)
    // Round 6 A/B on one box (profiles/r06_hash_ablation.txt), M1 hash batch at 2397 MHz: one level per workgroup, 512 threads
    // (two resident workgroups = 4 waves per SIMD) 76.7 us; 1024 threads (8 waves per SIMD) 84.9 us; level pairs 78.8 us.  More
    // waves do not help (the LDS atomic unit is shared by them), sharing the position load does not either (the loop is bound
    // by its own instructions, not by the loads).  NGM_HASH_THREADS=1024 / NGM_HASH_PAIR=1 reproduce the comparison.
    static const bool wide = getenv("NGM_HASH_THREADS") != nullptr && atoi(getenv("NGM_HASH_THREADS")) == 1024;
    if (pair) { if (flt) NGM_HG(true, 2, 1024); else NGM_HG(false, 2, 1024); }
    else if (wide) { if (flt) NGM_HG(true, 1, 1024); else NGM_HG(false, 1, 1024); }
    else { if (flt) NGM_HG(true, 1, 512); else NGM_HG(false, 1, 512); }
#undef NGM_HG
  }
  if (chunks > 1) {        // one chunk: k_hash_grad wrote the gradient table and applied the update itself
    NgmProfScope prof_(NGM_K_HASH_REDUCE, st);
    hipLaunchKernelGGL(k_hash_reduce, dim3((T / 2 + 255) / 256, fb.fc.nr_levels, fb.F), dim3(256), 0, st, a);
  }
  if (adam_applied) *adam_applied = a.ad_param != nullptr;
  return 0;
}
