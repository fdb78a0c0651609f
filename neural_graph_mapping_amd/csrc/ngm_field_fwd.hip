// Forward kernels of the per-field NeRF hot path (gfx950).
//   k_field_points_fwd : NeuralFieldSet.forward(use_vmap=True)  (models.py:329-345)
//   k_render_fwd       : NeuralGraphMap._render_ijs(use_vmap=True) fused in ONE pass
//                        (rm.py:439-666): ray setup -> stratified + depth-guided samples (rank
//                        merge, no sort) -> world->field-local -> encoding -> MLP on MFMA ->
//                        alpha compositing (wave segmented prefix-product scan) -> loss partial sums.
// One workgroup = 4 waves = one field (weights resident in LDS); each wave pushes 64 samples per
// step through the MLP as two 32-column MFMA tiles and owns one sample per lane for all per-sample
// scalar work.
#include "ngm_field.h"
#include "ngm_launch.h"

#include <algorithm>
#include <cstring>

#define WAVE_SYNC()                                        \
  do {                                                     \
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); \
    __builtin_amdgcn_wave_barrier();                       \
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); \
  } while (0)

// ------------------------------------------------------------------------------------------------
// NT: threads per workgroup; the bf16-split variant (83 KB of weights: one workgroup per CU) runs eight waves on them
template <int MI, int MH, int L, bool NEED_COS, int HASH, int SKIP, bool B3 = false, int NT = NGM_BLOCK>
__global__ __launch_bounds__(NT) void k_field_points_fwd(PointsFwdArgs a) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int f = blockIdx.x % a.F, chunk = blockIdx.x / a.F;
  const int64_t row = a.pr.field_index ? a.pr.field_index[f] : f;
  {
    FieldStage<MI, MH, L, SKIP == 2> fstage;       // every parameter load in flight at once, then the permuting LDS writes
    fstage.issue(a.fc, a.pr, row);
    fstage.commit(sm, a.fc);
  }
  __syncthreads();
  ngm_u32x4* const b3w = reinterpret_cast<ngm_u32x4*>(sm + FieldLds<MI, MH, L, SKIP == 2>::TOTAL);
  if constexpr (B3) {          // ngm_matmul_mode BF16X3: bf16 weight planes behind the fp32 fragments (ngm_field.h)
    b3_build_planes<MI, MH, L>(sm, b3w);
    __syncthreads();
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float div, off;
  scale_consts(a.fc.scale_mode, a.fc.field_radius, &div, &off);
  const bool posed = a.pos != nullptr;
  float px = 0, py = 0, pz = 0, qw = 1, qx = 0, qy = 0, qz = 0;
  if (posed) {
    px = a.pos[3 * f]; py = a.pos[3 * f + 1]; pz = a.pos[3 * f + 2];
    qw = a.quat[4 * f]; qx = a.quat[4 * f + 1]; qy = a.quat[4 * f + 2]; qz = a.quat[4 * f + 3];
  }
  const HashCtx hc = make_hash_ctx(a.fc, a.pr, row, nullptr);
  const TriCtx tc = make_tri_ctx(a.fc, a.pr, row, nullptr);
  const int64_t beg = (int64_t)chunk * a.per_block, end = min(a.P, beg + a.per_block);
  for (int64_t base = beg + wave * 64; base < end; base += NT) {
    const int64_t idx = base + lane;
    const bool valid = idx < end;
    float x = 0, y = 0, z = 0;
    if (valid) {
      const float* p = a.points + ((int64_t)f * a.P + idx) * 3;
      const Vec3 v = scaled_local_point(Vec3{p[0], p[1], p[2]}, posed, px, py, pz, qw, qx, qy, qz, div, off);
      x = v.x; y = v.y; z = v.z;
    }
    ActStash ast;
    ast.base = (HASH == 0 && MH == 2 && SKIP == 0) ? a.act : nullptr; ast.layer_stride = a.act_layer_stride;
    ast.g0 = (int64_t)f * a.P + base; ast.nvalid = (int)min((int64_t)64, end - base); ast.nlayers = NGM_MAX_LAYERS;
    const float4 o = eval_64<MI, MH, L, NEED_COS, HASH, SKIP, B3, false>(sm, lane, x, y, z, &hc, ast.base ? &ast : nullptr, nullptr, b3w, &tc);
    if (valid) reinterpret_cast<float4*>(a.out)[(int64_t)f * a.P + idx] = o;
  }
}

// ------------------------------------------------------------------------------------------------
// k_encode_points: the positional encoding ALONE (standalone stage entry point for roofline accounting;
// positional_encodings.py:19-66 hash / :164-276 Fourier, NeRF octaves): one lane per point, features to HBM as
// (F, P, dim_enc).  The fused kernels never materialise this tensor (the features go straight into MFMA operand registers);
// the arithmetic is theirs -- permuto_simplex + four 8-byte gathers per level, hardware sine of the argument in revolutions.
// ------------------------------------------------------------------------------------------------
struct EncodeArgs {
  ngm_field_cfg fc; ngm_params pr; int F; int64_t P;
  const float* points; const float* pos; const float* quat; float* out;
};
__global__ __launch_bounds__(256) void k_encode_points(EncodeArgs a) {
  const int f = blockIdx.y;
  const int64_t row = a.pr.field_index ? a.pr.field_index[f] : f;
  const int D = a.fc.dim_enc;
  float div, off;
  scale_consts(a.fc.scale_mode, a.fc.field_radius, &div, &off);
  const bool posed = a.pos != nullptr;
  float px = 0, py = 0, pz = 0, qw = 1, qx = 0, qy = 0, qz = 0;
  if (posed) {
    px = a.pos[3 * f]; py = a.pos[3 * f + 1]; pz = a.pos[3 * f + 2];
    qw = a.quat[4 * f]; qx = a.quat[4 * f + 1]; qy = a.quat[4 * f + 2]; qz = a.quat[4 * f + 3];
  }
  const HashCtx hc = make_hash_ctx(a.fc, a.pr, row, nullptr);
  for (int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; p < a.P; p += (int64_t)gridDim.x * blockDim.x) {
    const float* pt = a.points + ((int64_t)f * a.P + p) * 3;
    Vec3 v{pt[0], pt[1], pt[2]};
    if (posed) { v = Vec3{v.x - px, v.y - py, v.z - pz}; v = quat_rotate_inv(qw, qx, qy, qz, v); }
    const float x = v.x / div + off, y = v.y / div + off, z = v.z / div + off;
    float* o = a.out + ((int64_t)f * a.P + p) * D;
    if (a.fc.encoding == NGM_ENC_PERMUTO) {
      for (int l0 = 0; l0 < a.fc.nr_levels; l0 += 2) {       // two levels = one 16-byte store
        float ft[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          const int level = l0 + u;
          if (level >= a.fc.nr_levels) break;
          const float* hs = a.pr.shift + row * a.pr.shift_stride + 3 * level;
          const float lp[8] = {a.fc.level_scale[3 * level], a.fc.level_scale[3 * level + 1], a.fc.level_scale[3 * level + 2], 0.f,
                               hs[0], hs[1], hs[2], 0.f};
          uint32_t idx[4]; float bw[4]; float2 tv[4];
          permuto_simplex(x, y, z, lp, hc.mask, idx, bw);
          ngm_ldp2x4(hc.tab, (size_t)level * hc.T, idx, hc.dt, tv);
#pragma unroll
          for (int r = 0; r < 4; ++r) { ft[2 * u] = fmaf(tv[r].x, bw[r], ft[2 * u]); ft[2 * u + 1] = fmaf(tv[r].y, bw[r], ft[2 * u + 1]); }
        }
        if (l0 + 1 < a.fc.nr_levels && (D & 3) == 0) *reinterpret_cast<float4*>(o + 2 * l0) = make_float4(ft[0], ft[1], ft[2], ft[3]);
        else { o[2 * l0] = ft[0]; o[2 * l0 + 1] = ft[1]; if (l0 + 1 < a.fc.nr_levels) { o[2 * l0 + 2] = ft[2]; o[2 * l0 + 3] = ft[3]; } }
      }
    } else {
      const float inv2pi = 0.15915494309189535f;
      const int n_raw = (a.fc.encoding == NGM_ENC_FOURIER) ? (a.fc.raw_coords ? 3 : 0) : (a.fc.encoding == NGM_ENC_NERF ? 0 : 3);
      for (int ft = 0; ft < D; ++ft) {
        float val;
        if (ft < n_raw) val = (ft == 0) ? x : (ft == 1) ? y : z;
        else if (a.fc.encoding == NGM_ENC_FOURIER) {
          const int64_t e0 = row * a.pr.enc_w_stride + (int64_t)(ft - n_raw) * 3;
          const float arg = fmaf(ngm_ldp(a.pr.enc_w, e0 + 2, a.pr.dtype), z,
                                 fmaf(ngm_ldp(a.pr.enc_w, e0 + 1, a.pr.dtype), y, ngm_ldp(a.pr.enc_w, e0, a.pr.dtype) * x));
          val = __builtin_amdgcn_sinf(__builtin_amdgcn_fractf(arg * inv2pi));
        } else if (a.fc.encoding == NGM_ENC_NERF) {
          const int half = 3 * a.fc.num_octaves;
          const int g = (ft < half) ? ft : ft - half;
          const int d = g / a.fc.num_octaves, oc = g % a.fc.num_octaves;
          const float m = exp2f((float)(a.fc.start_octave + oc)) * 3.14159265358979323846f;
          const float rev = __builtin_amdgcn_fractf(m * ((d == 0) ? x : (d == 1) ? y : z) * inv2pi);
          val = (ft < half) ? __builtin_amdgcn_sinf(rev) : __builtin_amdgcn_cosf(rev);
        } else val = 0.f;
        o[ft] = val;
      }
    }
  }
}
int ngm_launch_encode_points(const ngm_field_cfg& fc, const ngm_params& pr, int F, int64_t P, const float* points, const float* pos,
                             const float* quat, float* out, hipStream_t st) {
  if (fc.encoding == NGM_ENC_TRIPLANE) return NGM_E_UNSUPPORTED;
  EncodeArgs a;
  a.fc = fc; a.pr = pr; a.F = F; a.P = P; a.points = points; a.pos = pos; a.quat = quat; a.out = out;
  const int bx = (int)std::min<int64_t>((P + 255) / 256, 4096);
  hipLaunchKernelGGL(k_encode_points, dim3(std::max(bx, 1), F), dim3(256), 0, st, a);
  return 0;
}

// ------------------------------------------------------------------------------------------------
// Backward of the encoding ALONE (ngm_encode_bwd): d_enc (F, P, dim_enc) -> the encoding's parameter
// gradients.  Fourier: dW[j][c] = sum_p d_enc[p][n_raw + j] cos(W_j . x_p) x_p[c] (positional_encodings.py:197-212), a wave
// per subset of the points with lane = feature, per-wave partials summed in a fixed order (deterministic).  Hash: the
// gradients are transposed to the level-major stream k_hash_grad reads (what k_hash_mlp_bwd hands it in the training step)
// together with the scaled local positions; the table gradient is that kernel's.  NeRF octaves / no encoding: no parameters.
// ------------------------------------------------------------------------------------------------
struct EncodeBwdArgs {
  ngm_field_cfg fc; ngm_params pr; int F; int64_t P;
  const float* points; const float* pos; const float* quat; const float* d_enc;
  float2* dE; float4* xyz;                   // hash: outputs for k_hash_grad
  float* part; int nblk; int64_t per_blk;    // fourier: [F][nblk][4][64][3] partial sums
  float* grad; int64_t grad_stride;          // fourier: (F, dim_enc - n_raw, 3)
};
__device__ __forceinline__ Vec3 encode_local_point(const EncodeBwdArgs& a, int f, int64_t p, float div, float off) {
  const float* pt = a.points + ((int64_t)f * a.P + p) * 3;
  Vec3 v{pt[0], pt[1], pt[2]};
  if (a.pos) {
    const float px = a.pos[3 * f], py = a.pos[3 * f + 1], pz = a.pos[3 * f + 2];
    v = Vec3{v.x - px, v.y - py, v.z - pz};
    v = quat_rotate_inv(a.quat[4 * f], a.quat[4 * f + 1], a.quat[4 * f + 2], a.quat[4 * f + 3], v);
  }
  return Vec3{v.x / div + off, v.y / div + off, v.z / div + off};      // the arithmetic of k_encode_points
}
__global__ __launch_bounds__(256) void k_encode_bwd_prep_hash(EncodeBwdArgs a) {
  const int f = blockIdx.y, D = a.fc.dim_enc, nlev = a.fc.nr_levels;
  float div, off;
  scale_consts(a.fc.scale_mode, a.fc.field_radius, &div, &off);
  const int64_t NP = (int64_t)a.F * a.P;
  for (int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; p < a.P; p += (int64_t)gridDim.x * blockDim.x) {
    const int64_t g = (int64_t)f * a.P + p;
    const Vec3 v = encode_local_point(a, f, p, div, off);
    a.xyz[g] = make_float4(v.x, v.y, v.z, 0.f);
    const float* d = a.d_enc + g * D;
    if ((D & 3) == 0 && nlev * 2 == D) {          // 16-byte row reads (two levels each), 8-byte level-major writes
      for (int l0 = 0; l0 < nlev; l0 += 2) {
        const float4 q = *reinterpret_cast<const float4*>(d + 2 * l0);
        a.dE[l0 * NP + g] = make_float2(q.x, q.y);
        a.dE[(l0 + 1) * NP + g] = make_float2(q.z, q.w);
      }
    } else {
      for (int level = 0; level < nlev; ++level) a.dE[level * NP + g] = make_float2(d[2 * level], d[2 * level + 1]);
    }
  }
}
__global__ __launch_bounds__(256) void k_encode_bwd_fourier(EncodeBwdArgs a) {
  const int f = blockIdx.y, blk = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t row = a.pr.field_index ? a.pr.field_index[f] : f;
  const int D = a.fc.dim_enc, n_raw = a.fc.raw_coords ? 3 : 0, nfeat = D - n_raw;
  float div, off;
  scale_consts(a.fc.scale_mode, a.fc.field_radius, &div, &off);
  const bool act = lane < nfeat;
  float w0 = 0.f, w1 = 0.f, w2 = 0.f;
  if (act) {
    const int64_t e0 = row * a.pr.enc_w_stride + (int64_t)lane * 3;
    w0 = ngm_ldp(a.pr.enc_w, e0, a.pr.dtype); w1 = ngm_ldp(a.pr.enc_w, e0 + 1, a.pr.dtype); w2 = ngm_ldp(a.pr.enc_w, e0 + 2, a.pr.dtype);
  }
  float s0 = 0.f, s1 = 0.f, s2 = 0.f;
  const int64_t beg = (int64_t)blk * a.per_blk, end = min(a.P, beg + a.per_blk);
  // tiles of 256 points: every thread transforms ONE point into the field's scaled local frame (three IEEE divisions, the
  // quaternion) and parks it in LDS; then each wave walks its quarter of the tile with lane = feature (the point is an LDS
  // broadcast, the gradients one coalesced 244-byte row)
  __shared__ float4 xl[256];
  for (int64_t t0 = beg; t0 < end; t0 += 256) {
    const int64_t pt = t0 + threadIdx.x;
    if (pt < end) { const Vec3 v = encode_local_point(a, f, pt, div, off); xl[threadIdx.x] = make_float4(v.x, v.y, v.z, 0.f); }
    __syncthreads();
    const int n = (int)min<int64_t>(256, end - t0);
#pragma unroll 4
    for (int t = wave; t < n; t += 4) {
      const float4 v = xl[t];
      const float arg = fmaf(w2, v.z, fmaf(w1, v.y, w0 * v.x));                   // the argument k_encode_points forms
      const float c = __builtin_amdgcn_cosf(__builtin_amdgcn_fractf(arg * 0.15915494309189535f));
      const float d = act ? a.d_enc[((int64_t)f * a.P + t0 + t) * D + n_raw + lane] : 0.f;
      const float dc = d * c;
      s0 = fmaf(dc, v.x, s0); s1 = fmaf(dc, v.y, s1); s2 = fmaf(dc, v.z, s2);
    }
    __syncthreads();
  }
  float* dst = a.part + ((((int64_t)f * a.nblk + blk) * 4 + wave) * 64 + lane) * 3;
  dst[0] = s0; dst[1] = s1; dst[2] = s2;
}
// Sum of the per-wave partials, fixed order -> deterministic.  Round 6: this was one thread per (feature, coordinate) walking
// all nblk x 4 partials alone -- 256 dependent strided loads: 236 us next to 202 us for the kernel that reads the gigabyte (the
// stage stood at 0.31 of the HBM peak because of its REDUCTION).  Now 16 groups of 64 threads per (field, coordinate) sum a
// sixteenth each with eight loads in flight, and one wave adds the 16 group sums in group order.
__global__ __launch_bounds__(1024) void k_encode_bwd_fourier_reduce(EncodeBwdArgs a) {
  __shared__ float gs[16][64];
  const int f = blockIdx.x, c = blockIdx.y, j = threadIdx.x & 63, g = threadIdx.x >> 6;
  const int nfeat = a.fc.dim_enc - (a.fc.raw_coords ? 3 : 0);
  const int Q = a.nblk * 4, per = (Q + 15) / 16, q0 = g * per, q1 = min(Q, q0 + per);
  const float* src = a.part + ((int64_t)f * Q * 64 + j) * 3 + c;
  float s = 0.f;
  if (j < nfeat) {
    int q = q0;
    for (; q + 8 <= q1; q += 8) {
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = src[(int64_t)(q + u) * 192];
#pragma unroll
      for (int u = 0; u < 8; ++u) s += v[u];
    }
    for (; q < q1; ++q) s += src[(int64_t)q * 192];
  }
  gs[g][j] = s;
  __syncthreads();
  if (g == 0 && j < nfeat) {
    float t = 0.f;
#pragma unroll
    for (int u = 0; u < 16; ++u) t += gs[u][j];
    a.grad[(int64_t)f * a.grad_stride + (int64_t)j * 3 + c] = t;
  }
}
int64_t ngm_encode_bwd_fourier_scratch(int F, int64_t P) {
  const int nblk = (int)std::max<int64_t>(1, std::min<int64_t>((P + 1023) / 1024, 256));
  return (int64_t)F * nblk * 4 * 64 * 3 * 4;
}
int ngm_launch_encode_bwd_fourier(const ngm_field_cfg& fc, const ngm_params& pr, int F, int64_t P, const float* points,
                                  const float* pos, const float* quat, const float* d_enc, float* grad, int64_t grad_stride,
                                  float* scratch, hipStream_t st) {
  EncodeBwdArgs a;
  memset(&a, 0, sizeof(a));
  a.fc = fc; a.pr = pr; a.F = F; a.P = P; a.points = points; a.pos = pos; a.quat = quat; a.d_enc = d_enc;
  a.nblk = (int)std::max<int64_t>(1, std::min<int64_t>((P + 1023) / 1024, 256));
  a.per_blk = (P + a.nblk - 1) / a.nblk;
  a.part = scratch; a.grad = grad; a.grad_stride = grad_stride;
  hipLaunchKernelGGL(k_encode_bwd_fourier, dim3(a.nblk, F), dim3(256), 0, st, a);
  hipLaunchKernelGGL(k_encode_bwd_fourier_reduce, dim3(F, 3), dim3(1024), 0, st, a);
  return 0;
}
int ngm_launch_encode_bwd_prep_hash(const ngm_field_cfg& fc, const ngm_params& pr, int F, int64_t P, const float* points,
                                    const float* pos, const float* quat, const float* d_enc, float2* dE, float4* xyz,
                                    hipStream_t st) {
  EncodeBwdArgs a;
  memset(&a, 0, sizeof(a));
  a.fc = fc; a.pr = pr; a.F = F; a.P = P; a.points = points; a.pos = pos; a.quat = quat; a.d_enc = d_enc; a.dE = dE; a.xyz = xyz;
  const int bx = (int)std::min<int64_t>((P + 255) / 256, 4096);
  hipLaunchKernelGGL(k_encode_bwd_prep_hash, dim3(std::max(bx, 1), F), dim3(256), 0, st, a);
  return 0;
}

// ------------------------------------------------------------------------------------------------
// fused render forward
// ------------------------------------------------------------------------------------------------
#define RF_BRMAX 32     // rays per wave batch
#define RF_MAXS_CAP 1024  // samples per wave batch (upper bound; actual size a.maxs is chosen by the host)
#define RF_RT 16        // floats per ray-table row
#define RF_RA 12        // floats per ray-accumulator row

// per-wave LDS carve (floats): ray table, ray accumulators, then 5 sample planes of `maxs`
struct RenderWaveLds {
  float (*rt)[RF_RT];     // ox,oy,oz, dx,dy,dz (field-local, scaled), dzcam, near,far,gnear,gfar, gt
  float (*ra)[RF_RA];     // C(3), D, W, Cv(3), Dv
  float* tbuf; float* wbuf; float* cbuf[3];
  __device__ __forceinline__ RenderWaveLds(float* base, int maxs) {
    rt = reinterpret_cast<float (*)[RF_RT]>(base);
    ra = reinterpret_cast<float (*)[RF_RA]>(base + RF_BRMAX * RF_RT);
    float* p = base + RF_BRMAX * (RF_RT + RF_RA);
    tbuf = p; wbuf = p + maxs; cbuf[0] = p + 2 * maxs; cbuf[1] = p + 3 * maxs; cbuf[2] = p + 4 * maxs;
  }
  static __host__ __device__ int floats(int maxs) { return RF_BRMAX * (RF_RT + RF_RA) + 5 * maxs; }
};

__device__ __forceinline__ int fdiv_idx(int idx, float inv_s, int S) {
  int q = (int)(((float)idx + 0.5f) * inv_s);
  // guard the reciprocal rounding at segment borders
  if (q * S > idx) --q;
  if ((q + 1) * S <= idx) ++q;
  return q;
}

// HALF (round 6): the instance for batches of at most 32 samples per wave (a rank of an 8-GPU run: one 24-sample ray per wave):
// the wave step evaluates ONE 32-sample tile (eval_32) where the batch allows it.  A separate instance, chosen by the launcher,
// so that the default instance's code and register allocation stay what they were (in-line it cost the M1 forward 0.9 us).
template <int MI, int MH, int L, bool NEED_COS, int HASH, int SKIP, bool B3 = false, bool NEUS = false, bool HALF = false>
__global__ __launch_bounds__(512) void k_render_fwd(RenderFwdArgs a) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  using LY = FieldLds<MI, MH, L, SKIP == 2>;
  const int F = a.rays.F, R = a.rays.R;
  const int f = blockIdx.x % F, chunk = blockIdx.x / F;
  const int64_t row = a.pr.field_index ? a.pr.field_index[f] : f;
#ifdef NGM_PHASE_TIMING
  // timeline of every wave of the middle block: 16 summary slots, then 64 log entries per wave
  PhaseClock pclk;
  pclk.begin((a.debug_cycles && blockIdx.x == gridDim.x / 2) ? a.debug_cycles + 16 + 64 * (threadIdx.x >> 6) : nullptr);
  PhaseClock* const pc = a.debug_cycles ? &pclk : nullptr;
#endif
  // the field parameters travel while the first batch of rays is set up (both are latency chains, and every
  // wave of the chip is in this phase at the same time, so there is no matrix work to hide them under)
  FieldStage<MI, MH, L, SKIP == 2> fstage;
  fstage.issue(a.fc, a.pr, row);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int nwaves = blockDim.x >> 6;
  RenderWaveLds wl(sm + LY::TOTAL + wave * RenderWaveLds::floats(a.maxs), a.maxs);

  const int S = a.S, S_c = a.rc.num_samples_coarse, S_g = a.S - S_c;
  const float inv_s = 1.0f / (float)S;
  const bool guided = S_g > 0;
  const float tau = a.rc.truncation_distance;
  const float gamma = a.rc.geometry_factor, cf = a.rc.color_factor;
  const int mode = a.rc.geometry_mode;

  float div, off;
  scale_consts(a.fc.scale_mode, a.fc.field_radius, &div, &off);
  const int64_t pf = a.rays.pose_index ? a.rays.pose_index[f] : f;
  const float px = a.rays.field_pos[3 * pf], py = a.rays.field_pos[3 * pf + 1], pz = a.rays.field_pos[3 * pf + 2];
  const float qw = a.rays.field_quat[4 * pf], qx = a.rays.field_quat[4 * pf + 1], qy = a.rays.field_quat[4 * pf + 2],
              qz = a.rays.field_quat[4 * pf + 3];

  // this wave's rays [r_beg, r_end) inside field f
  const int blk_beg = chunk * a.rays_per_block, blk_end = min(R, blk_beg + a.rays_per_block);
  const int per_wave = (blk_end - blk_beg + nwaves - 1) / nwaves;
  const int r_beg = min(blk_end, blk_beg + wave * per_wave), r_end = min(blk_end, r_beg + per_wave);
  const int BR = max(1, min(RF_BRMAX, a.maxs / S));

  const uint64_t poff = philox_launch_offset(a.rays);
  // loss partial sums of this lane: five fp32 sums (indexed by their NGM_LS_*_SUM slot) and the five counts packed in two
  // words (cnt_s: free-space | TSDF << 16, per sample; cnt_r: photometric | depth << 10 | termination << 20, per ray) --
  // three registers less in a kernel that sits at the 256-register limit, where a spilled accumulator is reloaded behind
  // the stash stores (one vmcnt for loads and stores: the reload waits for all of them)
  float ls[11];
#pragma unroll
  for (int i = 0; i < 11; ++i) ls[i] = 0.f;
  uint32_t cnt_s = 0, cnt_r = 0;
  const HashCtx hc = make_hash_ctx(a.fc, a.pr, row, nullptr);
  const TriCtx tc = make_tri_ctx(a.fc, a.pr, row, nullptr);
  float neus_isd = 0.f;                                             // rm.py:641-644: 1 / |_neus_sd| of this field
  if constexpr (NEUS) neus_isd = 1.0f / fabsf(a.neus_sd[row * a.neus_sd_stride]);

  // phases (1)+(2) of a batch; run for batch b+1 at the end of batch b's pass (and for the first batch while the
  // weights are still in flight)
  auto prepare = [&](int rb) __attribute__((always_inline)) {
    const int nb = min(BR, r_end - rb);
    const int nsamp = nb * S;
    // ---- (1) ray setup: one lane per ray
    if (lane < nb) {
      const int64_t ray = (int64_t)f * R + rb + lane;
      // pose rows first: three 16-byte loads that travel with the ray's scalars
      const float* Tp = a.rays.c2w_per_ray ? a.rays.c2ws + ray * 16 : a.rays.c2ws;
      float4 T0, T1, T2;
      if ((reinterpret_cast<uintptr_t>(a.rays.c2ws) & 15) == 0) {
        const float4* T4 = reinterpret_cast<const float4*>(Tp);
        T0 = T4[0]; T1 = T4[1]; T2 = T4[2];
      } else {
        T0 = make_float4(Tp[0], Tp[1], Tp[2], Tp[3]); T1 = make_float4(Tp[4], Tp[5], Tp[6], Tp[7]);
        T2 = make_float4(Tp[8], Tp[9], Tp[10], Tp[11]);
      }
      // the ray's targets too, before anything is stored: behind the ray-table stores below the compiler may not move these
      // loads up (it cannot prove the tables apart), and they were one more memory round trip at the end of the set-up
      float4 tgt4 = make_float4(0.f, 0.f, 0.f, 0.f);
      unsigned char tgt_dm = 0, tgt_tm = 0;
      float tgt_tp = 0.f;
      if (a.has_targets) {
        tgt4 = reinterpret_cast<const float4*>(a.tg.rgbds)[ray];
        tgt_dm = a.tg.depth_mask[ray];
        const unsigned char* tmp = a.tg.term_mask ? a.tg.term_mask + ray : a.tg.depth_mask + ray;
        const float* tpp = (a.tg.term_mask && a.tg.term_probs) ? a.tg.term_probs + ray : a.tg.rgbds;
        tgt_tm = *tmp;
        tgt_tp = *tpp;
      }
      const RayGeom g = ray_geom(a.rc, a.rays, ray, guided);
      Vec3 dw{T0.x * g.dx + T0.y * g.dy + T0.z * g.dz, T1.x * g.dx + T1.y * g.dy + T1.z * g.dz,
              T2.x * g.dx + T2.y * g.dy + T2.z * g.dz};
      Vec3 ow{T0.w - px, T1.w - py, T2.w - pz};
      dw = quat_rotate_inv(qw, qx, qy, qz, dw);
      ow = quat_rotate_inv(qw, qx, qy, qz, ow);
      float* rt = wl.rt[lane];
      rt[0] = ow.x / div + off; rt[1] = ow.y / div + off; rt[2] = ow.z / div + off;
      rt[3] = dw.x / div; rt[4] = dw.y / div; rt[5] = dw.z / div;
      rt[6] = g.dz; rt[7] = g.near; rt[8] = g.far; rt[9] = g.gnear; rt[10] = g.gfar; rt[11] = g.gt;
      if (a.raytab) {
        float4* o = reinterpret_cast<float4*>(a.raytab + ray * 8);
        o[0] = make_float4(rt[0], rt[1], rt[2], rt[3]);
        o[1] = make_float4(rt[4], rt[5], rt[6], g.gt);
      }
#pragma unroll
      for (int c = 0; c < RF_RA; ++c) wl.ra[lane][c] = 0.f;
      if (a.has_targets) {
        // the ray's targets ride along in the free table slots: the output phase at the end of the batch then
        // has no global load (and no memory latency) left in it
        const float4 tg = tgt4;
        const unsigned char dm = tgt_dm, tm = tgt_tm;
        const float tp = tgt_tp;
        rt[12] = tg.x; rt[13] = tg.y; rt[14] = tg.z; rt[15] = tg.w;
        wl.ra[lane][9] = dm ? 1.f : 0.f;
        wl.ra[lane][10] = (a.tg.term_mask && tm) ? 1.f : 0.f;
        wl.ra[lane][11] = tp;
      }
    }
    WAVE_SYNC();
    PTICK(pc, 1);
    // ---- (2) sorted sample distances: closed-form rank of every source element (no sort).  The draws are made once
    // per element into the (still unused) weight plane; the ranking reads its neighbours' draws from there.
#ifndef NGM_ABLF_NOSAMPLER
    // (one Philox block per FOUR elements: the batch's rays are consecutive, so are their elements in each stratum's stream)
    jitter_fill(a.rays, poff, 0, (int64_t)f * R + rb, nb, S_c, wl.wbuf, S, lane);
    if (S_g > 0) jitter_fill(a.rays, poff, 1, (int64_t)f * R + rb, nb, S_g, wl.wbuf + S_c, S, lane);
    WAVE_SYNC();
#endif
    for (int idx = lane; idx < nsamp; idx += 64) {
      const int rl = fdiv_idx(idx, inv_s, S), e = idx - rl * S;
      const int64_t ray = (int64_t)f * R + rb + rl;
      RayGeom g;
      g.near = wl.rt[rl][7]; g.far = wl.rt[rl][8]; g.gnear = wl.rt[rl][9]; g.gfar = wl.rt[rl][10];
      float t; int rank;
#ifdef NGM_ABLF_NOSAMPLER   // timing ablations of the fused forward (results meaningless when defined)
      t = g.near + (g.far - g.near) * (float)e / (float)S; rank = e;
#else
      sample_rank(a.rc, a.rays, poff, g, ray, e, S_c, S_g, &t, &rank, wl.wbuf + rl * S);
#endif
      wl.tbuf[rl * S + rank] = t;
    }
    WAVE_SYNC();
    PTICK(pc, 2);
  };
  if (r_beg < r_end) prepare(r_beg);
  fstage.commit(sm, a.fc);
  __syncthreads();
  // opt-in bf16 three-way split of the hidden layers: weight planes behind the per-wave regions
  ngm_u32x4* const b3w = reinterpret_cast<ngm_u32x4*>(sm + LY::TOTAL + nwaves * RenderWaveLds::floats(a.maxs));
  if constexpr (B3) {
    b3_build_planes<MI, MH, L>(sm, b3w);
    __syncthreads();
  }
  PTICK(pc, 0);
  // compositing of one 64-sample step (transmittance scan carried across steps, weights, per-ray sums, stash)
  int rb_cur = 0, nsamp_cur = 0;
  auto composite = [&](int idx, bool valid, int ci, int rl, int k, float t, float c0, float c1, float c2, float depth,
                       float geom, float occ, float& carry) __attribute__((always_inline)) {
    const int rb = rb_cur, nsamp = nsamp_cur;
      // transmittance: segmented inclusive product of (1-occ), carried across steps
      float q = seg_scan_mul(1.0f - occ, k, lane);
      if (k > lane) q *= carry;
      const float up = lane_prev(q, carry);
      const float T_excl = (k == 0) ? 1.0f : (lane == 0 ? carry : up);
      carry = lane_value(q, 63);
      const float w = valid ? occ * T_excl : 0.f;
      if (valid) {
        wl.wbuf[idx] = w; wl.cbuf[0][idx] = c0; wl.cbuf[1][idx] = c1; wl.cbuf[2][idx] = c2;
        if (a.stashA) {
          const int64_t gs = ((int64_t)f * R + rb) * S + idx;
          typedef float v4f __attribute__((ext_vector_type(4)));
          typedef float v2f __attribute__((ext_vector_type(2)));
          const v4f sa = {c0, c1, c2, geom};
          const v2f sb = {t, T_excl};
#ifdef NGM_STASH_NT      // rounds 1-5: non-temporal
          __builtin_nontemporal_store(sa, reinterpret_cast<v4f*>(a.stashA + gs));   // read once, by the next kernel
          __builtin_nontemporal_store(sb, reinterpret_cast<v2f*>(a.stashB + gs));
#else                    // round 6: written through (sc1), like the activation stash (ngm_field.h act_store): the backward reads these
                         // rows from another CU, most often another XCD -- M1 backward 141.4 -> 137.7 us, k_hash_mlp_bwd 37.3 -> 33.8 us
          asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(reinterpret_cast<v4f*>(a.stashA + gs)), "v"(sa));
          asm volatile("global_store_dwordx2 %0, %1, off sc1" ::"v"(reinterpret_cast<v2f*>(a.stashB + gs)), "v"(sb));
#endif
        }
      }
      float sc[5] = {w * c0, w * c1, w * c2, w * depth, w};      // five segmented sums over the same rays: one fused scan
      seg_scan_add_n<5>(sc, k, lane);
      const float s0 = sc[0], s1 = sc[1], s2 = sc[2], s3 = sc[3], s4 = sc[4];
      const bool tail = valid && (k == S - 1 || lane == 63 || idx == nsamp - 1);
      if (tail) {
        float* ra = wl.ra[rl];
        ra[0] += s0; ra[1] += s1; ra[2] += s2; ra[3] += s3; ra[4] += s4;
      }
  };
#ifdef NGM_FWD_STAGGER   // experiment: offset the second wave of every SIMD by a fraction of a step so that its matrix
  // phases meet the first wave's VALU phases instead of its matrix phases (NGM_FWD_STAGGER x 64 clocks)
  if (wave >= nwaves / 2) __builtin_amdgcn_s_sleep(NGM_FWD_STAGGER);
#endif
  for (int rb = r_beg; rb < r_end; rb += BR) {
    const int nb = min(BR, r_end - rb);
    const int nsamp = nb * S;
    rb_cur = rb; nsamp_cur = nsamp;
    // ---- (3) MLP + compositing pass, 64 consecutive flat samples per step
    float carry = 1.0f;
    for (int base = 0; base < nsamp; base += 64) {
      const int idx = base + lane;
      const bool valid = idx < nsamp;
      // lanes past the end work on a clamped index (selects, not branches: every LDS read below is issued at once)
      const int ci = min(idx, nsamp - 1);
      const int rl = fdiv_idx(ci, inv_s, S);
      const int k = valid ? ci - rl * S : 0;
      const float* rt = wl.rt[rl];
      const float t_raw = wl.tbuf[ci];
      const float t = valid ? t_raw : 0.f;
      const float x = valid ? fmaf(t, rt[3], rt[0]) : 0.f, y = valid ? fmaf(t, rt[4], rt[1]) : 0.f,
                  z = valid ? fmaf(t, rt[5], rt[2]) : 0.f;
      ActStash ast;
#ifdef NGM_ABLF_NOACT
      ast.base = nullptr; ast.layer_stride = 0;
#else
      ast.base = (HASH == 1 ? MI == 1 : (HASH == 0 && MH == 2)) ? a.act : nullptr; ast.layer_stride = a.act_layer_stride;
#endif
      ast.g0 = ((int64_t)f * R + rb) * S + base; ast.nvalid = nsamp - base;
      ast.nlayers = a.act_layers > 0 ? a.act_layers : NGM_MAX_LAYERS;
#ifdef NGM_ABLF_NOMLP
      const float4 o = make_float4(x, y, z, x * y);
#else
      PTICK(pc, 3);
#ifdef NGM_PHASE_TIMING
      const float4 o = eval_64<MI, MH, L, NEED_COS, HASH, SKIP, B3>(sm, lane, x, y, z, &hc, &ast, pc, b3w, &tc);
#else
      float4 o;
      if constexpr (HALF && ((B3 && HASH == 0) || (!B3 && HASH == 1)) && !NEED_COS && SKIP == 0 && !NEUS) {
        // (wave-uniform) at most 32 samples left in the batch: one tile instead of two (ngm_field.h eval_32)
        if (nsamp - base <= 32) o = eval_32<MI, MH, L, B3, HASH>(sm, lane, x, y, z, &ast, b3w, &hc);
        else o = eval_64<MI, MH, L, NEED_COS, HASH, SKIP, B3>(sm, lane, x, y, z, &hc, &ast, nullptr, b3w, &tc);
      } else {
        o = eval_64<MI, MH, L, NEED_COS, HASH, SKIP, B3>(sm, lane, x, y, z, &hc, &ast, nullptr, b3w, &tc);
      }
#endif
#endif
      const float c0 = cf * o.x, c1 = cf * o.y, c2 = cf * o.z;
      const float depth = -(rt[6] * t);
      // samples behind the camera (z_cam = dz * t > 0; only possible with near < 0): constant geometry, rm.py:614-622
      const float geom = (a.rc.overwrite_behind_camera && rt[6] * t > 0.f) ? behind_camera_geometry(mode) : o.w;
      // free-space / TSDF loss terms read the geometry itself (rm.py:624-639): same for every mode
      if (a.has_targets && valid) {
        const float gt = rt[11];
        const float thr = (gt - tau) * (gt != 0.0f ? 1.0f : 0.0f);       // rm.py:625-627
        if (t < thr) { const float e = geom * tau - tau; ls[NGM_LS_FS_SUM] += e * e; cnt_s += 1u; }
        const float dl = gt - t;
        if (fabsf(dl) < tau && gt != 0.0f) {                              // rm.py:633-637
          const float e = geom * tau - dl; ls[NGM_LS_TSDF_SUM] += e * e; cnt_s += 0x10000u;
        }
      }
      if constexpr (NEUS) {
        // neus (rm.py:753-758): occ_k needs the geometry of sample k + 1, i.e. another lane's -- or the next step's --
        // MLP output.  Colours and geometry are parked in the wave's LDS planes (the weight plane holds the geometry
        // until the compositing pass below overwrites it with the weights); compositing runs once the batch is complete.
        if (valid) { wl.wbuf[idx] = geom; wl.cbuf[0][idx] = c0; wl.cbuf[1][idx] = c1; wl.cbuf[2][idx] = c2; }
      } else {
      float occ;
      if (mode == NGM_GEO_DENSITY) {                                       // rm.py:746-749, last sample dropped
        const float o_d = occ_density(geom, wl.tbuf[min(ci + 1, nsamp - 1)] - t, nullptr);
        occ = (k < S - 1) ? o_d : 0.f;
      } else occ = occ_pointwise_fwd(mode, gamma, geom);
      occ = valid ? occ : 0.f;
        composite(idx, valid, ci, rl, k, t, c0, c1, c2, depth, geom, occ, carry);
      }
      WAVE_SYNC();
      PTICK(pc, 8);
    }
    if constexpr (NEUS) {
      // ---- (3b) neus compositing over the parked planes: tno = sigmoid(isd gamma g), occ_k = max((tno_k - tno_{k+1}) /
      // (tno_k + 1e-5), 0), last sample dropped (rm.py:753-758, last_index = -1)
      const float isd_g = neus_isd * gamma;
      carry = 1.0f;
      for (int base = 0; base < nsamp; base += 64) {
        const int idx = base + lane;
        const bool valid = idx < nsamp;
        const int ci = min(idx, nsamp - 1);
        const int rl = fdiv_idx(ci, inv_s, S);
        const int k = valid ? ci - rl * S : 0;
        const float* rt = wl.rt[rl];
        const float t = valid ? wl.tbuf[ci] : 0.f;
        const float geom = wl.wbuf[ci], gnext = wl.wbuf[min(ci + 1, nsamp - 1)];
        const float c0 = wl.cbuf[0][ci], c1 = wl.cbuf[1][ci], c2 = wl.cbuf[2][ci];
        const float depth = -(rt[6] * t);
        const float tno = ngm_sigmoid(isd_g * geom), tnx = ngm_sigmoid(isd_g * gnext);
        float occ = fmaxf((tno - tnx) / (tno + 1e-5f), 0.f);
        occ = (valid && k < S - 1) ? occ : 0.f;
        WAVE_SYNC();     // every lane has read its neighbour's geometry before the weights overwrite the plane
        composite(idx, valid, ci, rl, k, t, c0, c1, c2, depth, geom, occ, carry);
        WAVE_SYNC();
      }
    }
    // ---- (4) variance pass around the finished means (rm.py:781-790)
#ifdef NGM_ABLF_NOVAR
    for (int base = 0; base < 0; base += 64) {
#else
    for (int base = 0; base < nsamp; base += 64) {
#endif
      const int idx = base + lane;
      const bool valid = idx < nsamp;
      const int ci = min(idx, nsamp - 1);
      const int rl = fdiv_idx(ci, inv_s, S);
      const int k = valid ? ci - rl * S : 0;
      const float* ra = wl.ra[rl];
      const float w_raw = wl.wbuf[ci], t_raw = wl.tbuf[ci];
      const float q0 = wl.cbuf[0][ci], q1 = wl.cbuf[1][ci], q2 = wl.cbuf[2][ci];
      const float w = valid ? w_raw : 0.f;
      const float t = valid ? t_raw : 0.f;
      const float depth = -(wl.rt[rl][6] * t);
      float e0 = ra[0] - (valid ? q0 : 0.f), e1 = ra[1] - (valid ? q1 : 0.f),
            e2 = ra[2] - (valid ? q2 : 0.f), e3 = ra[3] - depth;
      float vv[4] = {w * (e0 * e0), w * (e1 * e1), w * (e2 * e2), w * (e3 * e3)};
      seg_scan_add_n<4>(vv, k, lane);
      const float v0 = vv[0], v1 = vv[1], v2 = vv[2], v3 = vv[3];
      const bool tail = valid && (k == S - 1 || lane == 63 || idx == nsamp - 1);
      WAVE_SYNC();   // all lanes have read ra[] means before tails update the variance slots
      if (tail) {
        float* rw = wl.ra[rl];
        rw[5] += v0; rw[6] += v1; rw[7] += v2; rw[8] += v3;
      }
      WAVE_SYNC();
    }
    PTICK(pc, 9);
    // ---- (5) per-ray outputs + loss partial sums
    if (lane < nb) {
      const int64_t ray = (int64_t)f * R + rb + lane;
      const float* ra = wl.ra[lane];
      const float term = 1.0f - (1.0f - ra[4]);                           // rm.py:774,796
      if (a.pred.rgbds) reinterpret_cast<float4*>(a.pred.rgbds)[ray] = make_float4(ra[0], ra[1], ra[2], ra[3]);
      if (a.pred.color_vars) { float* cv = a.pred.color_vars + ray * 3; cv[0] = ra[5]; cv[1] = ra[6]; cv[2] = ra[7]; }
      if (a.pred.depth_vars) a.pred.depth_vars[ray] = ra[8];
      if (a.pred.term_probs) a.pred.term_probs[ray] = term;
      if (a.has_targets) {
        const float* rt = wl.rt[lane];
        const float4 tg = make_float4(rt[12], rt[13], rt[14], rt[15]);
        const bool m = (ra[9] != 0.f) && (term > a.rc.term_threshold);  // rm.py:1787
        if (m) {
          const float p0 = tg.x - ra[0], p1 = tg.y - ra[1], p2 = tg.z - ra[2];
          const float l1 = fabsf(p0) + fabsf(p1) + fabsf(p2);
          if (a.rc.photometric_mode == NGM_PHOTO_GAUSSIAN_NLL) {        // losses.py:30-36: 0.5 e^2 / var + log sqrt(var), no epsilon
            ls[NGM_LS_PHOTO_SUM] += (0.5f * p0 * p0 / ra[5] + logf(sqrtf(ra[5]))) + (0.5f * p1 * p1 / ra[6] + logf(sqrtf(ra[6]))) +
                                    (0.5f * p2 * p2 / ra[7] + logf(sqrtf(ra[7])));
            ls[NGM_LS_PHOTO_L1_SUM] += l1;                               // its L1 branch (mean NLL > 2), decided on the global sums
          } else
            ls[NGM_LS_PHOTO_SUM] += (a.rc.photometric_mode == NGM_PHOTO_L2) ? fmaf(p2, p2, fmaf(p1, p1, p0 * p0))   // losses.py:28-29
                                                                             : l1;                                  // losses.py:26-27
          cnt_r += 1u;
          const float e = ra[3] - tg.w, ae = fabsf(e), dlt = a.rc.huber_delta;
          if (a.rc.depth_mode == NGM_DEPTH_GAUSSIAN_NLL) {               // losses.py:64-69
            const float dv = ra[8] + 1e-15f;
            ls[NGM_LS_DEPTH_SUM] += 0.5f * e * e / dv + logf(sqrtf(dv));
          } else if (a.rc.depth_mode == NGM_DEPTH_LAPLACIAN_NLL)         // losses.py:70-75
            ls[NGM_LS_DEPTH_SUM] += ae / sqrtf(0.5f * ra[8] + 1e-6f) + 0.5f * logf(2.0f * ra[8] + 1e-6f);
          else
            ls[NGM_LS_DEPTH_SUM] += (ae < dlt) ? 0.5f * e * e : dlt * (ae - 0.5f * dlt);
          cnt_r += 1u << 10;
        }
        if (ra[10] != 0.f) {
          const float e = term - ra[11];
          ls[NGM_LS_TERM_SUM] += e * e; cnt_r += 1u << 20;
        }
        if (a.rayseed) {
          // what k_stash_bwd derives per ray from prediction and target, minus the global normalisers: e = prediction -
          // target; L1: sign(e), L2: e (the backward multiplies by 2 k); Huber'(depth error); termination error
          const float e0 = ra[0] - tg.x, e1 = ra[1] - tg.y, e2 = ra[2] - tg.z, ed = ra[3] - tg.w, dlt = a.rc.huber_delta;
          const bool l2 = a.rc.photometric_mode == NGM_PHOTO_L2;
          float4 s0 = make_float4(0.f, 0.f, 0.f, 0.f);
          if (m) {
            s0.x = l2 ? e0 : (float)((e0 > 0.f) - (e0 < 0.f)); s0.y = l2 ? e1 : (float)((e1 > 0.f) - (e1 < 0.f));
            s0.z = l2 ? e2 : (float)((e2 > 0.f) - (e2 < 0.f));
            s0.w = (fabsf(ed) < dlt) ? ed : dlt * (float)((ed > 0.f) - (ed < 0.f));
          }
          float4* o = reinterpret_cast<float4*>(a.rayseed + ray * 8);
          o[0] = s0;
          o[1] = make_float4((ra[10] != 0.f) ? term - ra[11] : 0.f, 0.f, 0.f, 0.f);
        }
      }
    }
    WAVE_SYNC();
    PTICK(pc, 10);
    if (rb + BR < r_end) prepare(rb + BR);
  }
  // ---- block reduction of the loss partial sums (deterministic order)
  if (a.has_targets && a.loss_partials) {
    __syncthreads();
    float* red = sm + LY::TOTAL;  // reuse wave 0's scratch (>= 8*16 floats)
    ls[NGM_LS_FS_CNT] = (float)(cnt_s & 0xffffu); ls[NGM_LS_TSDF_CNT] = (float)(cnt_s >> 16);
    ls[NGM_LS_PHOTO_CNT] = (float)(cnt_r & 1023u); ls[NGM_LS_DEPTH_CNT] = (float)((cnt_r >> 10) & 1023u);
    ls[NGM_LS_TERM_CNT] = (float)(cnt_r >> 20);
#pragma unroll
    for (int i = 0; i < 11; ++i) {
      const float v = wave_sum(ls[i]);
      if (lane == 0) red[wave * 16 + i] = v;
    }
    __syncthreads();
    if (threadIdx.x < NGM_NUM_LOSS_SUMS) {
      float v = 0.f;
      if (threadIdx.x < 11)
        for (int w = 0; w < nwaves; ++w) v += red[w * 16 + threadIdx.x];
      a.loss_partials[(int64_t)blockIdx.x * NGM_NUM_LOSS_SUMS + threadIdx.x] = v;
    }
  }
#ifdef NGM_PHASE_TIMING
  PTICK(pc, 11);
  if (pc && blockIdx.x == gridDim.x / 2 && threadIdx.x == 0) {
    for (int k = 0; k < 14; ++k) a.debug_cycles[k] = pclk.acc[k];
    a.debug_cycles[14] = __builtin_readcyclecounter() - pclk.start;
    a.debug_cycles[15] = __builtin_amdgcn_s_memrealtime() - pclk.rstart;   // 100 MHz ticks: gives the shader clock
  }
#endif
}

// ------------------------------------------------------------------------------------------------
// host launchers
// ------------------------------------------------------------------------------------------------
// every encoding x skip_mode {no, add, concat} (models.py:159-169 is encoding-agnostic)
#define NGM_LAUNCH_ONE(KERNEL, NC, HS, SK, GRID, BLK, LDSW, LDSX)                                                       \
  do {                                                                                                                 \
    const size_t lds_ = (FieldLds<MI, MH, L, (SK) == 2>::TOTAL + (LDSX)) * sizeof(float);                              \
    (void)(LDSW);                                                                                                      \
    (void)hipFuncSetAttribute((const void*)KERNEL<MI, MH, L, NC, HS, SK>, hipFuncAttributeMaxDynamicSharedMemorySize,   \
                              (int)lds_);                                                                              \
    hipLaunchKernelGGL((KERNEL<MI, MH, L, NC, HS, SK>), dim3(GRID), BLK, lds_, st, a);                                 \
  } while (0)
#define NGM_LAUNCH_VARIANT(KERNEL, NC, HS, GRID, BLK, LDSW, LDSX)                               \
  do {                                                                                          \
    if (a.fc.skip_mode == NGM_SKIP_ADD) NGM_LAUNCH_ONE(KERNEL, NC, HS, 1, GRID, BLK, LDSW, LDSX);   \
    else if (a.fc.skip_mode == NGM_SKIP_CONCAT) NGM_LAUNCH_ONE(KERNEL, NC, HS, 2, GRID, BLK, LDSW, LDSX); \
    else NGM_LAUNCH_ONE(KERNEL, NC, HS, 0, GRID, BLK, LDSW, LDSX);                              \
  } while (0)

// the bf16 split path (ngm_matmul_mode) is compiled for 49..64-wide layers, <= 2 hidden layers, Fourier / no encoding, skip no
template <int MI, int MH, int L>
static constexpr bool b3_shape() { return MI == 2 && MH == 2 && L <= 2; }
static bool b3_wanted(const ngm_field_cfg& fc) {
  return (fc.matmul_mode == NGM_MATMUL_BF16X3 || fc.matmul_mode == NGM_MATMUL_AUTO) && fc.skip_mode == NGM_SKIP_NO &&
         (fc.encoding == NGM_ENC_FOURIER || fc.encoding == NGM_ENC_NONE);
}

template <int MI, int MH, int L>
static int launch_points(const PointsFwdArgs& a, int blocks, hipStream_t st) {
  const dim3 blk(NGM_BLOCK);
  if constexpr (b3_shape<MI, MH, L>()) {
    if (b3_wanted(a.fc)) {       // standalone evaluation: the mode is a preference here (fp32 MFMA where not compiled)
      g_ngm_last_matmul[1] = NGM_MATMUL_BF16X3;
      const size_t lds = FieldLds<MI, MH, L>::TOTAL * sizeof(float) + (size_t)B3Lds<MI, MH, L>::TOTAL * 16;
      (void)hipFuncSetAttribute((const void*)k_field_points_fwd<MI, MH, L, false, 0, 0, true, 512>,
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      hipLaunchKernelGGL((k_field_points_fwd<MI, MH, L, false, 0, 0, true, 512>), dim3(blocks), dim3(512), lds, st, a);
      return 0;
    }
  }
  g_ngm_last_matmul[1] = NGM_MATMUL_F32;
  if (a.fc.encoding == NGM_ENC_PERMUTO) {
    if constexpr (MI == 1) NGM_LAUNCH_VARIANT(k_field_points_fwd, false, 1, blocks, blk, 0, 0);
    else return NGM_E_UNSUPPORTED;
  } else if (a.fc.encoding == NGM_ENC_TRIPLANE) NGM_LAUNCH_VARIANT(k_field_points_fwd, false, 2, blocks, blk, 0, 0); else if (a.fc.encoding == NGM_ENC_NERF) NGM_LAUNCH_VARIANT(k_field_points_fwd, true, 0, blocks, blk, 0, 0);
  else NGM_LAUNCH_VARIANT(k_field_points_fwd, false, 0, blocks, blk, 0, 0);
  return 0;
}
template <int MI, int MH, int L>
static int launch_render(const RenderFwdArgs& a, int blocks, hipStream_t st) {
  const size_t wave_lds = (size_t)a.waves_per_block * RenderWaveLds::floats(a.maxs);
  const dim3 blk(64 * a.waves_per_block);
  g_ngm_last_matmul[0] = NGM_MATMUL_F32;
  g_ngm_last_fwd_one_tile = 0;
  if (a.rc.geometry_mode == NGM_GEO_NEUS) {
    // neus in the fused kernel (two-pass compositing over the wave's LDS planes): Fourier / no encoding, skip no, fp32 MFMA
    if (a.fc.skip_mode != NGM_SKIP_NO || (a.fc.encoding != NGM_ENC_FOURIER && a.fc.encoding != NGM_ENC_NONE) || !a.neus_sd)
      return NGM_E_UNSUPPORTED;
    const size_t lds = (FieldLds<MI, MH, L>::TOTAL + wave_lds) * sizeof(float);
    (void)hipFuncSetAttribute((const void*)k_render_fwd<MI, MH, L, false, 0, 0, false, true>,
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL((k_render_fwd<MI, MH, L, false, 0, 0, false, true>), dim3(blocks), blk, lds, st, a);
    return 0;
  }
  if (a.fc.matmul_mode == NGM_MATMUL_BF16X3) {
    // opt-in: hidden layers as a three-way bf16 split on v_mfma_f32_32x32x16_bf16 (ngm_field.h, layer_fwd_b3)
    if constexpr (MI == 2 && MH == 2 && L <= 2) {
      if (a.fc.skip_mode == NGM_SKIP_NO && (a.fc.encoding == NGM_ENC_FOURIER || a.fc.encoding == NGM_ENC_NONE)) {
        const size_t lds = (FieldLds<MI, MH, L>::TOTAL + wave_lds) * sizeof(float) + (size_t)B3Lds<MI, MH, L>::TOTAL * 16;
        if (lds > 160 * 1024) return NGM_E_UNSUPPORTED;
        g_ngm_last_matmul[0] = NGM_MATMUL_BF16X3;
        // every wave's batches hold at most 32 samples (rays per wave x samples per ray): the one-tile instance
        const int per_wave = (a.rays_per_block + a.waves_per_block - 1) / a.waves_per_block;
        static const bool no_half = getenv("NGM_NO_HALF_STEP") != nullptr;          // developer A/B switch
        g_ngm_last_fwd_one_tile = (per_wave * a.S <= 32 && !no_half) ? 1 : 0;
        if (per_wave * a.S <= 32 && !no_half) {
          (void)hipFuncSetAttribute((const void*)k_render_fwd<MI, MH, L, false, 0, 0, true, false, true>,
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
          hipLaunchKernelGGL((k_render_fwd<MI, MH, L, false, 0, 0, true, false, true>), dim3(blocks), blk, lds, st, a);
          return 0;
        }
        (void)hipFuncSetAttribute((const void*)k_render_fwd<MI, MH, L, false, 0, 0, true>,
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL((k_render_fwd<MI, MH, L, false, 0, 0, true>), dim3(blocks), blk, lds, st, a);
        return 0;
      }
    }
    return NGM_E_UNSUPPORTED;
  }
  if (a.fc.encoding == NGM_ENC_PERMUTO) {
    if constexpr (MI == 1) {
      // the reference's default network on small per-rank batches: the one-tile instance (see the split path above)
      const int per_wave = (a.rays_per_block + a.waves_per_block - 1) / a.waves_per_block;
      static const bool no_half = getenv("NGM_NO_HALF_STEP") != nullptr;
      if (per_wave * a.S <= 32 && !no_half && a.fc.skip_mode == NGM_SKIP_NO) {
        const size_t lds_ = (FieldLds<MI, MH, L>::TOTAL + wave_lds) * sizeof(float);
        (void)hipFuncSetAttribute((const void*)k_render_fwd<MI, MH, L, false, 1, 0, false, false, true>,
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_);
        hipLaunchKernelGGL((k_render_fwd<MI, MH, L, false, 1, 0, false, false, true>), dim3(blocks), blk, lds_, st, a);
        g_ngm_last_fwd_one_tile = 1;
        return 0;
      }
      NGM_LAUNCH_VARIANT(k_render_fwd, false, 1, blocks, blk, 0, wave_lds);
    }
    else return NGM_E_UNSUPPORTED;
  } else if (a.fc.encoding == NGM_ENC_TRIPLANE) NGM_LAUNCH_VARIANT(k_render_fwd, false, 2, blocks, blk, 0, wave_lds); else if (a.fc.encoding == NGM_ENC_NERF) NGM_LAUNCH_VARIANT(k_render_fwd, true, 0, blocks, blk, 0, wave_lds);
  else NGM_LAUNCH_VARIANT(k_render_fwd, false, 0, blocks, blk, 0, wave_lds);
  return 0;
}

#ifdef NGM_FAST_BUILD
#define NGM_SHAPE_DISPATCH(FN, ...)                                             \
  do {                                                                          \
    const FieldShape s_ = field_shape(&a.fc);                                   \
    if (s_.MI == 2 && s_.MH == 2 && s_.L == 2) return FN<2, 2, 2>(__VA_ARGS__); \
    return NGM_E_UNSUPPORTED;                                                   \
  } while (0)
#else
#define NGM_SHAPE_DISPATCH(FN, ...)                                             \
  do {                                                                          \
    const FieldShape s_ = field_shape(&a.fc);                                   \
    if (s_.MI == 2 && s_.MH == 2 && s_.L == 2) return FN<2, 2, 2>(__VA_ARGS__); \
    if (s_.MI == 2 && s_.MH == 2 && s_.L == 1) return FN<2, 2, 1>(__VA_ARGS__); \
    if (s_.MI == 1 && s_.MH == 1 && s_.L == 1) return FN<1, 1, 1>(__VA_ARGS__); \
    if (s_.MI == 1 && s_.MH == 1 && s_.L == 2) return FN<1, 1, 2>(__VA_ARGS__); \
    if (s_.MI == 2 && s_.MH == 2 && s_.L == 3) return FN<2, 2, 3>(__VA_ARGS__); \
    return NGM_E_UNSUPPORTED;                                                   \
  } while (0)
#endif

int ngm_launch_points_fwd(const PointsFwdArgs& a, int blocks, hipStream_t st) {
  NgmProfScope prof_(NGM_K_POINTS_FWD, st);
  NGM_SHAPE_DISPATCH(launch_points, a, blocks, st);
}
int ngm_launch_render_fwd(const RenderFwdArgs& a, int blocks, hipStream_t st) {
  NgmProfScope prof_(NGM_K_RENDER_FWD, st);
  NGM_SHAPE_DISPATCH(launch_render, a, blocks, st);
}
