// Device-side building blocks shared by all gfx950 kernels of the render/train hot path.
// wave = 64 lanes everywhere (CDNA4); MFMA = v_mfma_f32_32x32x2_f32 (exact fp32, fmaf-chain numerics).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/ngm_hip.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

#ifndef NGM_BLOCK
#define NGM_WAVE 64
#define NGM_BLOCK 256
#define NGM_WAVES_PER_BLOCK 4
#endif

// ------------------------------------------------------------------------------------------------
// MFMA 32x32x2 f32 fragment maps (cdna_hip_programming.md section 3):
//   A: lane l holds A[i = l&31][k = l>>5]       B: lane l holds B[k = l>>5][j = l&31]
//   C/D: lane l, reg r holds C[row = (r&3) + 8*(r>>2) + 4*(l>>5)][col = l&31]
// We always put FEATURES on M (rows) and SAMPLES on N (cols = lane&31).  A layer's output
// registers (C layout) are then directly the B operands of the next layer: k-step r of input tile
// mi consumes register r, i.e. feature 32*mi + frow(r, hi) -- the weights (A operand) are stored in
// LDS in exactly that permuted k order, so activations never move between lanes.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ constexpr int frow(int r, int hi) { return (r & 3) + 8 * (r >> 2) + 4 * hi; }

__device__ __forceinline__ f32x16 mfma32(float a, float b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}

// weight fragment storage in LDS: element W[32*mo + io][32*mi + frow(r,hi)] lives at
//   ((((mo*MIN + mi)*16 + r)*2 + hi) * WGS + io);  WGS = 33 keeps both the forward read
// (lanes vary io) and the transposed dgrad read (lanes vary (r,hi)) bank-conflict free.
#define NGM_WGS 33
__device__ __forceinline__ constexpr int wfrag_size(int mout, int min_) { return mout * min_ * 16 * 2 * NGM_WGS; }
// (r', hi') such that frow(r', hi') == c for c in [0,32)
__device__ __forceinline__ constexpr int col_r(int c) { return (c & 3) + 4 * (c >> 3); }
__device__ __forceinline__ constexpr int col_hi(int c) { return (c >> 2) & 1; }

// ------------------------------------------------------------------------------------------------
// fast accurate sin/cos: Cody-Waite 3-term reduction by pi/2 + cephes kernels on [-pi/4,pi/4].
// Branch-free (no large-argument fallback: a data-dependent branch per feature splits the MFMA loops
// into hundreds of basic blocks).  Max abs error 9.3e-8 for |x| < 1e5 (checked against fp64 on the
// host, 2e7 samples); beyond that the accuracy degrades smoothly (|x| * 2^-48), never NaN for finite x.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void ngm_sincosf(float x, float* s_out, float* c_out) {
  const float n = rintf(x * 0.63661977236758134f);
  float r = fmaf(n, -1.57079601287841796875f, x);       // pi/2 hi
  r = fmaf(n, -3.1391647326017846353e-7f, r);           // pi/2 mid
  r = fmaf(n, -5.3903025299577648e-15f, r);             // pi/2 lo
  const float z = r * r;
  // cephes single-precision kernels on [-pi/4, pi/4]
  float ps = -1.9515295891e-4f;
  ps = fmaf(ps, z, 8.3321608736e-3f);
  ps = fmaf(ps, z, -1.6666654611e-1f);
  const float sr = fmaf(r * z, ps, r);
  float pc = 2.443315711809948e-5f;
  pc = fmaf(pc, z, -1.388731625493765e-3f);
  pc = fmaf(pc, z, 4.166664568298827e-2f);
  const float cr = fmaf(z * z, pc, fmaf(-0.5f, z, 1.0f));
  const int q = (int)n;
  const float s1 = (q & 1) ? cr : sr;
  const float c1 = (q & 1) ? sr : cr;
  *s_out = (q & 2) ? -s1 : s1;
  *c_out = ((q + 1) & 2) ? -c1 : c1;
}

// sin alone (the forward of the Fourier encoding needs no cosine): reduction by pi to [-pi/2, pi/2] with the
// round-to-nearest "magic number" trick (the parity of k is the low mantissa bit of n), one odd polynomial of
// degree 11 (minimax in r^2, fitted offline: approximation error 2e-11), sign from the parity bit.  15 VALU
// instead of the 28 of ngm_sincosf; max abs error 1.2e-7 for |x| < 400 (fp64 reference, 4e6 samples), 1.9e-7
// for |x| < 4000.  Valid for |x| < 2^22 * pi; branch-free.
__device__ __forceinline__ float ngm_sinf(float x) {
  const float magic = 12582912.0f;                       // 1.5 * 2^23
  const float n = fmaf(x, 0.3183098861837907f, magic);
  const float k = n - magic;
  float r = fmaf(k, -3.140625f, x);                      // pi hi (8 bits)
  r = fmaf(k, -9.670257568359375e-4f, r);                // pi mid
  r = fmaf(k, -6.2771141529083251953e-7f, r);            // pi lo
  const float z = r * r;
  float p = -2.3846690373585197e-08f;
  p = fmaf(p, z, 2.7522618610619714e-06f);
  p = fmaf(p, z, -1.9840804033395678e-04f);
  p = fmaf(p, z, 8.33333049561397e-03f);
  p = fmaf(p, z, -1.6666666606465025e-01f);
  const float s = fmaf(r * z, p, r);
  return __uint_as_float(__float_as_uint(s) ^ (__float_as_uint(n) << 31));
}

// ------------------------------------------------------------------------------------------------
// reduced-precision parameter STORAGE (ngm_params.dtype): element i of a tensor whose nominal float* base addresses
// fp32, bf16 or fp16 elements; always widened to fp32 (exact), all arithmetic stays fp32
// ------------------------------------------------------------------------------------------------
// Both conversions and a select: as an if / else this put every 16-bit load into a branch diamond, and the compiler waits
// for a load before leaving the block that issued it -- the weight staging of the 16-bit storage types was a chain of
// serial memory round trips (section 3.14 of DESIGN.md).
__device__ __forceinline__ float ngm_widen(uint32_t h16, int dt) {
  const float as_bf16 = __uint_as_float(h16 << 16);
  const float as_f16 = (float)__builtin_bit_cast(_Float16, (unsigned short)h16);
  return (dt == NGM_DT_BF16) ? as_bf16 : as_f16;
}
__device__ __forceinline__ float ngm_ldp(const float* base, int64_t i, int dt) {
  if (dt == NGM_DT_F32) return base[i];
  return ngm_widen(reinterpret_cast<const unsigned short*>(base)[i], dt);
}
// N elements at arbitrary (valid) offsets behind ONE storage-type branch: all loads are issued before any is used (through
// ngm_ldp every element is a branch diamond of its own and is waited for where it was loaded)
template <int N>
__device__ __forceinline__ void ngm_ldp_gather(const float* base0, int64_t first, const int (&off)[N], int dt, float (&x)[N]) {
  if (dt == NGM_DT_F32) {
    const float* base = base0 + first;
#pragma unroll
    for (int e = 0; e < N; ++e) x[e] = base[off[e]];
  } else {
    const unsigned short* b = reinterpret_cast<const unsigned short*>(base0) + first;
    uint32_t raw[N];
#pragma unroll
    for (int e = 0; e < N; ++e) raw[e] = b[off[e]];
#pragma unroll
    for (int e = 0; e < N; ++e) x[e] = ngm_widen(raw[e], dt);
  }
}
// elements i..i+3; `vec`: one 16-byte (fp32) / 8-byte (16-bit) load is allowed (caller checked alignment and bounds)
__device__ __forceinline__ float4 ngm_ldp4(const float* base, int64_t i, int dt, bool vec, int n_valid) {
  float4 x = make_float4(0.f, 0.f, 0.f, 0.f);
  if (vec) {
    if (dt == NGM_DT_F32) return *reinterpret_cast<const float4*>(base + i);
    const uint2 raw = *reinterpret_cast<const uint2*>(reinterpret_cast<const unsigned short*>(base) + i);
    return make_float4(ngm_widen(raw.x & 0xffffu, dt), ngm_widen(raw.x >> 16, dt), ngm_widen(raw.y & 0xffffu, dt),
                       ngm_widen(raw.y >> 16, dt));
  }
  x.x = ngm_ldp(base, i, dt);
  if (n_valid > 1) x.y = ngm_ldp(base, i + 1, dt);
  if (n_valid > 2) x.z = ngm_ldp(base, i + 2, dt);
  if (n_valid > 3) x.w = ngm_ldp(base, i + 3, dt);
  return x;
}
// hash table entry (2 features): 8 bytes fp32, 4 bytes 16-bit
__device__ __forceinline__ float2 ngm_ldp2(const void* tab, size_t entry, int dt) {
  if (dt == NGM_DT_F32) return reinterpret_cast<const float2*>(tab)[entry];
  const uint32_t raw = reinterpret_cast<const uint32_t*>(tab)[entry];
  return make_float2(ngm_widen(raw & 0xffffu, dt), ngm_widen(raw >> 16, dt));
}
// Four table entries at once: ONE storage-type branch around the four loads.  Through ngm_ldp2 every gather sat in a branch
// diamond of its own and the compiler waited for it (s_waitcnt vmcnt(0)) before issuing the next: the 64 gathers of a
// 64-sample step of the hash forward were 64 serial L2 round trips.
__device__ __forceinline__ void ngm_ldp2x4(const void* tab, size_t base, const uint32_t (&idx)[4], int dt, float2 (&v)[4]) {
  if (dt == NGM_DT_F32) {
    const float2* t = reinterpret_cast<const float2*>(tab) + base;
#pragma unroll
    for (int r = 0; r < 4; ++r) v[r] = t[idx[r]];
  } else {
    const uint32_t* t = reinterpret_cast<const uint32_t*>(tab) + base;
    uint32_t raw[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) raw[r] = t[idx[r]];
#pragma unroll
    for (int r = 0; r < 4; ++r) v[r] = make_float2(ngm_widen(raw[r] & 0xffffu, dt), ngm_widen(raw[r] >> 16, dt));
  }
}
// fp32 -> storage type, round to nearest even (the copy an Adam update leaves for the kernels)
__device__ __forceinline__ void ngm_stp(void* base, int64_t i, float v, int dt) {
  unsigned short h;
  if (dt == NGM_DT_BF16) {
    const uint32_t u = __float_as_uint(v);
    h = (v != v) ? (unsigned short)0x7fc0 : (unsigned short)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16);
  } else {
    h = __builtin_bit_cast(unsigned short, (_Float16)v);
  }
  reinterpret_cast<unsigned short*>(base)[i] = h;
}

__device__ __forceinline__ float ngm_sigmoid(float x) { return 1.0f / (1.0f + expf(-x)); }

// relu on a matrix-core result in ONE instruction.  fmaxf(y, 0) on an MFMA output compiles to two v_max_f32 (the first
// quiets a possible signalling NaN); v_med3_f32(y, 0, +inf) needs no canonicalisation and agrees with fmaxf for every
// input, NaN included (both return 0).
// one v_max_f32 (the builtin max / med3 forms get a second, canonicalising v_max from the compiler: MFMA results
// could be signalling NaNs as far as it knows)
__device__ __forceinline__ float ngm_relu(float y) {
  float r;
  asm("v_max_f32 %0, 0, %1" : "=v"(r) : "v"(y));
  return r;
}

// Two sines per instruction stream: the same routine as ngm_sinf on a float2, so that the 13 arithmetic steps
// compile to packed-fp32 instructions (v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32: two lanes' worth of fp32 per
// issue slot on gfx950); only the sign flip stays per element.  Element-wise identical to ngm_sinf.
typedef float ngm_v2f __attribute__((ext_vector_type(2)));
__device__ __forceinline__ ngm_v2f ngm_splat2(float v) { return ngm_v2f{v, v}; }
__device__ __forceinline__ ngm_v2f ngm_fma2(ngm_v2f a, ngm_v2f b, ngm_v2f c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ ngm_v2f ngm_sinf2(ngm_v2f x) {
  const ngm_v2f magic = ngm_splat2(12582912.0f);
  const ngm_v2f n = ngm_fma2(x, ngm_splat2(0.3183098861837907f), magic);
  const ngm_v2f k = n - magic;
  ngm_v2f r = ngm_fma2(k, ngm_splat2(-3.140625f), x);
  r = ngm_fma2(k, ngm_splat2(-9.670257568359375e-4f), r);
  r = ngm_fma2(k, ngm_splat2(-6.2771141529083251953e-7f), r);
  const ngm_v2f z = r * r;
  ngm_v2f p = ngm_splat2(-2.3846690373585197e-08f);
  p = ngm_fma2(p, z, ngm_splat2(2.7522618610619714e-06f));
  p = ngm_fma2(p, z, ngm_splat2(-1.9840804033395678e-04f));
  p = ngm_fma2(p, z, ngm_splat2(8.33333049561397e-03f));
  p = ngm_fma2(p, z, ngm_splat2(-1.6666666606465025e-01f));
  const ngm_v2f s = ngm_fma2(r * z, p, r);
  ngm_v2f o;
  o.x = __uint_as_float(__float_as_uint(s.x) ^ (__float_as_uint(n.x) << 31));
  o.y = __uint_as_float(__float_as_uint(s.y) ^ (__float_as_uint(n.y) << 31));
  return o;
}

// ------------------------------------------------------------------------------------------------
// Write-through (sc1) stores for data one kernel hands to the NEXT launch (round 6).  A consumer workgroup lands on any XCD;
// written through, the data is at the memory side when the producer ends (nothing dirty left in its L2 for the boundary to
// flush) and the consumer's loads are served from there.  16- and 8-byte forms only: narrower write-through stores are one
// fabric write each (MI355X_MICROARCH.md, stores of each flavour).
// ------------------------------------------------------------------------------------------------
typedef float ngm_v4f_ __attribute__((ext_vector_type(4)));
typedef float ngm_v2f_ __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void ngm_store_wt(float4* p, const float4& v) {
  const ngm_v4f_ x = {v.x, v.y, v.z, v.w};
  asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(p), "v"(x));
}
__device__ __forceinline__ void ngm_store_wt(ngm_v4f_* p, ngm_v4f_ x) { asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(p), "v"(x)); }
__device__ __forceinline__ void ngm_store_wt(ngm_v2f_* p, ngm_v2f_ x) { asm volatile("global_store_dwordx2 %0, %1, off sc1" ::"v"(p), "v"(x)); }

// ------------------------------------------------------------------------------------------------
// Philox4x32-10 counter RNG (used when the caller passes no explicit torch.rand draws)
// ------------------------------------------------------------------------------------------------
// One block = counter (ctr low 32, ctr high 32, stream id, offset low 32) under key (seed low 32, seed high 32) -> four words.
__device__ __forceinline__ void philox_block(uint64_t seed, uint64_t offset, uint64_t ctr, uint32_t stream_id, uint32_t (&w)[4]) {
  uint32_t c0 = (uint32_t)ctr, c1 = (uint32_t)(ctr >> 32), c2 = stream_id, c3 = (uint32_t)offset;
  uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
#pragma unroll
  for (int i = 0; i < 10; ++i) {
    // one 32 x 32 -> 64 product per multiplier (v_mad_u64_u32) instead of a mul_hi / mul_lo pair: the integer multiplies
    // are quarter-rate and all there is to a round
    const uint64_t pr0 = (uint64_t)0xD2511F53u * c0, pr1 = (uint64_t)0xCD9E8D57u * c2;
    const uint32_t hi0 = (uint32_t)(pr0 >> 32), lo0 = (uint32_t)pr0;
    const uint32_t hi1 = (uint32_t)(pr1 >> 32), lo1 = (uint32_t)pr1;
    const uint32_t n0 = hi1 ^ c1 ^ k0, n2 = hi0 ^ c3 ^ k1;
    c0 = n0; c1 = lo1; c2 = n2; c3 = lo0;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  w[0] = c0; w[1] = c1; w[2] = c2; w[3] = c3;
}
__device__ __forceinline__ float philox_word_uniform(uint32_t w) { return (float)(w >> 8) * (1.0f / 16777216.0f); }   // [0,1)
// Element `idx` of stream `stream_id`: word idx & 3 of block idx >> 2 (round 6: all four words of a block are used -- rounds
// 1-5 spent a whole block per element and threw three words away; tests/_philox_host.py restates this mapping on the host and
// the in-kernel draws are compared with it bit for bit).  This form still computes a block per call: the samplers that draw
// for whole rays use jitter_fill below, which shares a block between four elements.
__device__ __forceinline__ float philox_uniform(uint64_t seed, uint64_t offset, uint64_t idx, uint32_t stream_id) {
  uint32_t w[4];
  philox_block(seed, offset, idx >> 2, stream_id, w);
  const uint32_t k = (uint32_t)idx & 3u;
  return philox_word_uniform((k & 2u) ? ((k & 1u) ? w[3] : w[2]) : ((k & 1u) ? w[1] : w[0]));
}
// k_sample_rays_weighted: block = element (NOT idx >> 2), its first TWO words are the bin draw and the offset draw
__device__ __forceinline__ void philox_uniform2(uint64_t seed, uint64_t offset, uint64_t idx, uint32_t stream_id, float* u0, float* u1) {
  uint32_t c0 = (uint32_t)idx, c1 = (uint32_t)(idx >> 32), c2 = stream_id, c3 = (uint32_t)offset;
  uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
#pragma unroll
  for (int i = 0; i < 10; ++i) {
    const uint64_t pr0 = (uint64_t)0xD2511F53u * c0, pr1 = (uint64_t)0xCD9E8D57u * c2;
    const uint32_t hi0 = (uint32_t)(pr0 >> 32), lo0 = (uint32_t)pr0;
    const uint32_t hi1 = (uint32_t)(pr1 >> 32), lo1 = (uint32_t)pr1;
    const uint32_t n0 = hi1 ^ c1 ^ k0, n2 = hi0 ^ c3 ^ k1;
    c0 = n0; c1 = lo1; c2 = n2; c3 = lo0;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  *u0 = (float)(c0 >> 8) * (1.0f / 16777216.0f);
  *u1 = (float)(c1 >> 8) * (1.0f / 16777216.0f);
}

// ------------------------------------------------------------------------------------------------
// quaternion / pose helpers (models.py:329-339 with pytorch3d's Hamilton convention, real first)
// ------------------------------------------------------------------------------------------------
struct Vec3 { float x, y, z; };
__device__ __forceinline__ Vec3 cross3(Vec3 a, Vec3 b) {
  return Vec3{a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}
// pytorch3d's quaternion_apply(quaternion_invert(q), v) for q = (w, u), models.py:338-339: the RAW Hamilton products
// q' (0, v) q'* with q' = (w, a), a = -u, i.e. (w^2 - a.a) v + 2 (a.v) a + 2 w (a x v) = |q|^2 v + w t + a x t, t = 2 (a x v):
// the rotation for a unit quaternion, and -- like the reference, which never normalises -- |q|^2 times it otherwise
__device__ __forceinline__ Vec3 quat_rotate_inv(float w, float ux, float uy, float uz, Vec3 v) {
  const Vec3 u{-ux, -uy, -uz};
  const float n2 = w * w + ux * ux + uy * uy + uz * uz;
  Vec3 t = cross3(u, v);
  t.x *= 2.0f; t.y *= 2.0f; t.z *= 2.0f;
  const Vec3 c = cross3(u, t);
  return Vec3{n2 * v.x + w * t.x + c.x, n2 * v.y + w * t.y + c.y, n2 * v.z + w * t.z + c.z};
}
// world point -> field-local -> scaled coordinates (models.py:329-339, 278-285) for the STANDALONE evaluation kernels (vmap-style
// k_field_points_fwd and the kNN path's k_knn_eval), every operation rounded separately: the two then give the same bits for
// the same point whatever the optimiser fuses around them (an ulp of difference here is 2^7 pi ulps in a NeRF octave argument)
__device__ __forceinline__ Vec3 scaled_local_point(Vec3 p, bool posed, float px, float py, float pz, float qw, float qx, float qy,
                                                   float qz, float div, float off) {
#pragma clang fp contract(off)
  Vec3 v = p;
  if (posed) {
    v = Vec3{p.x - px, p.y - py, p.z - pz};
    const Vec3 u{-qx, -qy, -qz};
    const float n2 = qw * qw + qx * qx + qy * qy + qz * qz;      // quat_rotate_inv's arithmetic, rounded op by op
    Vec3 t{u.y * v.z - u.z * v.y, u.z * v.x - u.x * v.z, u.x * v.y - u.y * v.x};
    t.x *= 2.0f; t.y *= 2.0f; t.z *= 2.0f;
    const Vec3 c{u.y * t.z - u.z * t.y, u.z * t.x - u.x * t.z, u.x * t.y - u.y * t.x};
    v = Vec3{n2 * v.x + qw * t.x + c.x, n2 * v.y + qw * t.y + c.y, n2 * v.z + qw * t.z + c.z};
  }
  return Vec3{v.x / div + off, v.y / div + off, v.z / div + off};
}
__device__ __forceinline__ void scale_consts(int scale_mode, float radius, float* div, float* off) {
  if (scale_mode == NGM_SCALE_UNIT_CUBE) { *div = 2.0f * radius; *off = 0.5f; }
  else if (scale_mode == NGM_SCALE_UNIT_BALL) { *div = radius; *off = 0.0f; }
  else { *div = 1.0f; *off = 0.0f; }
}

// ------------------------------------------------------------------------------------------------
// ray sampler arithmetic (camera.py:186-203, 269-276; rm.py:521-545).  Separate roundings (no FMA
// contraction) so that distances reproduce the reference's op sequence bit for bit.
// ------------------------------------------------------------------------------------------------
struct RayGeom {
  float dx, dy, dz;        // unit direction, camera frame (OpenGL: -z forward)
  float near, far;         // coarse stratum
  float gnear, gfar;       // depth-guided stratum (falls back to near/far, rm.py:522-530)
  float gt;
};

// NOTE: plain operators, not __fmul_rn/__fadd_rn: those header wrappers are compiled with the `contract`
// fast-math flag and still fuse into FMAs after inlining; the pragma only governs operators written here.
__device__ __forceinline__ float strat_lin(const float* lin_tab, int n, int i) {
#pragma clang fp contract(off)
  if (lin_tab) return lin_tab[i];
  // torch.linspace(0,1,n+1) scalar formula (RangeFactories): symmetric about the middle
  const float step = 1.0f / (float)n;
  const int steps = n + 1, half = steps / 2;
  const float a = step * (float)i;
  const float b = step * (float)(steps - i - 1);
  return (i < half) ? a : (1.0f - b);
}
// t_i = (delta*u + lin_i*(far-near)) + near, every operation rounded separately (camera.py:269-276)
__device__ __forceinline__ float strat_t(float near, float far, int n, int i, float u, const float* lin_tab) {
#pragma clang fp contract(off)
  const float span = far - near;
  const float delta = span / (float)n;
  const float b = strat_lin(lin_tab, n, i) * span;
  const float du = delta * u;
  const float s = du + b;
  return s + near;
}

__device__ __forceinline__ RayGeom ray_geom(const ngm_render_cfg& cfg, const ngm_rays& rays, int64_t ray, bool guided) {
#pragma clang fp contract(off)
  RayGeom g;
  // optional per-ray arrays: every load is issued unconditionally (absent arrays read a valid dummy address) so
  // that they travel together; behind null checks each one would be its own basic block and its own memory latency
  const float* dummy = reinterpret_cast<const float*>(rays.ijs);
  const float near_v = *(rays.near ? rays.near + ray : dummy);
  const float far_v = *(rays.far ? rays.far + ray : dummy);
  const float gt_v = *(rays.gt ? rays.gt + ray : dummy);
  const int64_t i = rays.ijs[2 * ray], j = rays.ijs[2 * ray + 1];
  const float vx = ((float)j - cfg.cx) / cfg.fx;
  const float vy = -(((float)i - cfg.cy) / cfg.fy);
  const float vz = -1.0f;
  const float xx = vx * vx, yy = vy * vy, zz = vz * vz;
  const float sxy = xx + yy;
  const float nrm = sqrtf(sxy + zz);
  const float den = fmaxf(nrm, 1e-12f);
  g.dx = vx / den; g.dy = vy / den; g.dz = vz / den;
  g.near = rays.near ? near_v : rays.near_const;
  g.far = rays.far ? far_v : rays.far_const;
  g.gt = rays.gt ? gt_v : 0.0f;
  g.gnear = g.near; g.gfar = g.far;
  if (guided) {
    const bool invalid = (g.gt == 0.0f) || (g.near > g.gt) || (g.far < g.gt);
    if (!invalid) { g.gnear = g.gt - cfg.range_depth_guided; g.gfar = g.gt + cfg.range_depth_guided; }
  }
  return g;
}

// utils.transform_points (utils.py:276-286 via rm.py:547) of the camera-frame sample d t of `ray`: p_w = R p_c + t, the
// products summed left to right, every operation rounded separately
__device__ __forceinline__ void sample_world_point(const ngm_rays& rays, const RayGeom& rg, int64_t ray, float t, float* wx,
                                                   float* wy, float* wz) {
#pragma clang fp contract(off)
  const float* T = rays.c2w_per_ray ? rays.c2ws + ray * 16 : rays.c2ws;
  const float cx = rg.dx * t, cy = rg.dy * t, cz = rg.dz * t;
  *wx = (T[0] * cx + T[1] * cy + T[2] * cz) + T[3];
  *wy = (T[4] * cx + T[5] * cy + T[6] * cz) + T[7];
  *wz = (T[8] * cx + T[9] * cy + T[10] * cz) + T[11];
}

// Philox stream offset of this launch: read ONCE per kernel (the device-side part is what a hipGraph replay advances);
// re-reading it per draw put a global-load latency in front of every sample of the sampler.
__device__ __forceinline__ uint64_t philox_launch_offset(const ngm_rays& rays) {
  return rays.philox_offset + (rays.philox_offset_dev ? *rays.philox_offset_dev : 0ull);
}
__device__ __forceinline__ float jitter(const ngm_rays& rays, uint64_t poff, int which, int64_t ray, int n, int i) {
  const float* u = which ? rays.u_guided : rays.u_coarse;
  if (u) return u[ray * n + i];
  return philox_uniform(rays.philox_seed, poff, (uint64_t)(ray * n + i), (uint32_t)which);
}
// idx / S for 0 <= idx < 2^24 by reciprocal multiplication, corrected at the segment borders
__device__ __forceinline__ int fdiv_small(int idx, float inv_s, int S) {
  int q = (int)(((float)idx + 0.5f) * inv_s);
  if (q * S > idx) --q;
  if ((q + 1) * S <= idx) ++q;
  return q;
}
// The draws of stratum `which` for the `nrays` consecutive rays from `ray0` on, by one wave: element e of ray ray0 + rl goes to
// out[rl * out_stride + e].  The elements of consecutive rays are consecutive indices of the stream (ray * n + e), so one
// Philox block (~100 vector instructions) serves four of them; the block-per-element form was ~100 of the sampler's ~255
// instructions per sample (profiles/r05_pmc_stages_sq.json).  Same values as jitter() element by element.
__device__ __forceinline__ void jitter_fill(const ngm_rays& rays, uint64_t poff, int which, int64_t ray0, int nrays, int n,
                                            float* out, int out_stride, int lane) {
  const float* u = which ? rays.u_guided : rays.u_coarse;
  const int total = nrays * n;
  const float inv_n = 1.0f / (float)n;
  if (u) {
    const float* src = u + ray0 * n;
    for (int j = lane; j < total; j += 64) {
      const int rl = fdiv_small(j, inv_n, n);
      out[rl * out_stride + (j - rl * n)] = src[j];
    }
    return;
  }
  const uint64_t i0 = (uint64_t)ray0 * (uint64_t)n, i1 = i0 + (uint64_t)total;
  for (uint64_t blk = (i0 >> 2) + (uint64_t)lane; (blk << 2) < i1; blk += 64) {
    uint32_t w[4];
    philox_block(rays.philox_seed, poff, blk, (uint32_t)which, w);
    const int j0 = (int)((int64_t)(blk << 2) - (int64_t)i0);            // may be -3..-1 for the first block
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int j = j0 + k;
      if (j >= 0 && j < total) {
        const int rl = fdiv_small(j, inv_n, n);
        out[rl * out_stride + (j - rl * n)] = philox_word_uniform(w[k]);
      }
    }
  }
}

// number of elements of stratum (near,far,n) that are < x (strict=1) or <= x (strict=0).
// Closed form from the stratified structure (element j lies in [near+j*d, near+(j+1)*d)), with the
// three candidates around the boundary compared explicitly so fp32 rounding cannot mis-rank.
// ucache (optional): this ray's draws of stratum `which`, already made (LDS) -- each Philox draw is ~100 VALU
// instructions and every element is looked at by up to three elements of the other stratum.
__device__ __forceinline__ int strat_count_below(const ngm_rays& rays, uint64_t poff, int which, int64_t ray, float near, float far,
                                                 int n, const float* lin_tab, float x, bool strict,
                                                 const float* ucache = nullptr) {
  const float span = far - near;
  int q;
  if (!(span > 0.0f)) q = 0;
  else {
    float qf = floorf((x - near) / (span / (float)n));
    qf = fminf(fmaxf(qf, -2.0f), (float)n + 2.0f);
    q = (int)qf;
  }
  int cnt = min(max(q - 1, 0), n);
  if (!(span > 0.0f)) cnt = 0;
  const int lo = (span > 0.0f) ? max(q - 1, 0) : 0;
  const int hi = (span > 0.0f) ? min(q + 1, n - 1) : n - 1;
  for (int j = lo; j <= hi; ++j) {
    const float tj = strat_t(near, far, n, j, ucache ? ucache[j] : jitter(rays, poff, which, ray, n, j), lin_tab);
    cnt += strict ? (tj < x) : (tj <= x);
  }
  return min(cnt, n);
}

// sorted distance + rank of source element e (0..S_c-1 coarse, S_c.. guided) of a ray
// ucache (optional): the S_c coarse then S_g guided draws of this ray, in source order
__device__ __forceinline__ void sample_rank(const ngm_render_cfg& cfg, const ngm_rays& rays, uint64_t poff, const RayGeom& g,
                                            int64_t ray, int e, int S_c, int S_g, float* t_out, int* rank_out,
                                            const float* ucache = nullptr) {
  if (e < S_c) {
    const float u = ucache ? ucache[e] : jitter(rays, poff, 0, ray, S_c, e);
    const float t = strat_t(g.near, g.far, S_c, e, u, rays.lin_coarse);
    int rank = e;
    if (S_g > 0) rank += strat_count_below(rays, poff, 1, ray, g.gnear, g.gfar, S_g, rays.lin_guided, t, true,
                                           ucache ? ucache + S_c : nullptr);
    *t_out = t; *rank_out = rank;
  } else {
    const int j = e - S_c;
    const float u = ucache ? ucache[e] : jitter(rays, poff, 1, ray, S_g, j);
    const float t = strat_t(g.gnear, g.gfar, S_g, j, u, rays.lin_guided);
    const int rank = j + strat_count_below(rays, poff, 0, ray, g.near, g.far, S_c, rays.lin_coarse, t, false, ucache);
    *t_out = t; *rank_out = rank;
  }
}

// phase timing of the fused forward (debug builds, -DNGM_PHASE_TIMING; compiled out otherwise): PTICK(pc, k) adds
// the shader-clock cycles since the previous tick to slot k
struct PhaseClock {
  unsigned long long acc[14];
  unsigned long long last, start, rstart;
  unsigned long long* log;   // optional per-wave event log: (slot << 48) | cycles since begin(), 64 entries
  int nlog;
  __device__ __forceinline__ void begin(unsigned long long* wave_log = nullptr) {
#pragma unroll
    for (int i = 0; i < 14; ++i) acc[i] = 0;
    log = wave_log; nlog = 0;
    rstart = __builtin_amdgcn_s_memrealtime();
    start = last = __builtin_readcyclecounter();
  }
  __device__ __forceinline__ void tick(int k) {
    const unsigned long long now = __builtin_readcyclecounter();
    acc[k] += now - last;
    if (log && nlog < 64 && (threadIdx.x & 63) == 0) log[nlog] = ((unsigned long long)k << 48) | (now - start);
    ++nlog;
    last = __builtin_readcyclecounter();
  }
};
#ifdef NGM_PHASE_TIMING
#define PTICK(pc, k) do { if (pc) (pc)->tick(k); } while (0)
#else
#define PTICK(pc, k)
#endif

// ------------------------------------------------------------------------------------------------
// wave-level segmented scans over 64 lanes = 64 consecutive flat samples.  A segment = one ray;
// k = sample index inside the ray, so lane-d belongs to the same segment iff k >= d (&& lane >= d).
// ------------------------------------------------------------------------------------------------
// The lane exchange is DPP (row_shr inside the 16-lane rows, then row_bcast:15 / row_bcast:31 to carry the row
// totals forward): 6 VALU-only steps, no LDS-pipe ds_bpermute and no wait.  Lanes the DPP source does not
// reach keep `ident`, so the in-row bound check comes for free; `k >= d` is the segment condition.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_take(float ident, float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(ident), __float_as_int(v), CTRL, ROW_MASK, 0xf, false));
}
#define NGM_DPP_ROW_SHR(d) (0x110 + (d))
#define NGM_DPP_ROW_BCAST15 0x142
#define NGM_DPP_ROW_BCAST31 0x143
#define NGM_DPP_WAVE_SHR1 0x138
__device__ __forceinline__ float seg_scan_add(float v, int k, int lane) {
  const int pr = lane & 15;
  float o;
  o = dpp_take<NGM_DPP_ROW_SHR(1), 0xf>(0.f, v); v += (k >= 1) ? o : 0.f;
  o = dpp_take<NGM_DPP_ROW_SHR(2), 0xf>(0.f, v); v += (k >= 2) ? o : 0.f;
  o = dpp_take<NGM_DPP_ROW_SHR(4), 0xf>(0.f, v); v += (k >= 4) ? o : 0.f;
  o = dpp_take<NGM_DPP_ROW_SHR(8), 0xf>(0.f, v); v += (k >= 8) ? o : 0.f;
  o = dpp_take<NGM_DPP_ROW_BCAST15, 0xa>(0.f, v); v += (k > pr) ? o : 0.f;          // rows 1,3 += total of row 0,2
  o = dpp_take<NGM_DPP_ROW_BCAST31, 0xc>(0.f, v); v += (k > lane - 32) ? o : 0.f;   // rows 2,3 += total of lanes 0..31
  return v;
}
// N segmented sum scans over the same segmentation at once.  A step of seg_scan_add is v_mov_dpp + v_cndmask + v_add per
// value; here it is ONE v_fmac_f32_dpp: v += dpp(v) * m with the step's condition as a 1.0 / 0.0 factor shared by the N
// values (same single rounding as the add; values must be finite).  The compiler does not fold a DPP move into an fmac,
// hence the asm; its hazard recognizer does not look into asm either, so the two wait states a DPP read needs after a VALU
// write of the same register are supplied by hand: an s_nop in front of every step, and value-minor order inside a step.
template <int N>
__device__ __forceinline__ void seg_scan_add_n(float (&v)[N], int k, int lane) {
  static_assert(N >= 3, "value-minor order must put two instructions between a register's write and its next DPP read");
  const int pr = lane & 15;
  const float m[6] = {(k >= 1) ? 1.f : 0.f, (k >= 2) ? 1.f : 0.f, (k >= 4) ? 1.f : 0.f, (k >= 8) ? 1.f : 0.f,
                      (k > pr) ? 1.f : 0.f, (k > lane - 32) ? 1.f : 0.f};
  // volatile asm statements keep their program order: one s_nop per step (covers the compiler's own VALU write in front
  // of the first step), then the N values in turn
#define NGM_SSA_STEP(CTRL, M)                                                                       \
  do {                                                                                              \
    asm volatile("s_nop 1");                                                                        \
    _Pragma("unroll") for (int c_ = 0; c_ < N; ++c_)                                                \
      asm volatile("v_fmac_f32_dpp %0, %0, %1 " CTRL : "+v"(v[c_]) : "v"(M));                       \
  } while (0)
  NGM_SSA_STEP("row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1", m[0]);
  NGM_SSA_STEP("row_shr:2 row_mask:0xf bank_mask:0xf bound_ctrl:1", m[1]);
  NGM_SSA_STEP("row_shr:4 row_mask:0xf bank_mask:0xf bound_ctrl:1", m[2]);
  NGM_SSA_STEP("row_shr:8 row_mask:0xf bank_mask:0xf bound_ctrl:1", m[3]);
  NGM_SSA_STEP("row_bcast:15 row_mask:0xa bank_mask:0xf", m[4]);             // rows 1, 3 += total of rows 0, 2
  NGM_SSA_STEP("row_bcast:31 row_mask:0xc bank_mask:0xf", m[5]);             // rows 2, 3 += total of lanes 0..31
#undef NGM_SSA_STEP
}
__device__ __forceinline__ float seg_scan_mul(float v, int k, int lane) {
  const int pr = lane & 15;
  float o;
  o = dpp_take<NGM_DPP_ROW_SHR(1), 0xf>(1.f, v); v *= (k >= 1) ? o : 1.f;
  o = dpp_take<NGM_DPP_ROW_SHR(2), 0xf>(1.f, v); v *= (k >= 2) ? o : 1.f;
  o = dpp_take<NGM_DPP_ROW_SHR(4), 0xf>(1.f, v); v *= (k >= 4) ? o : 1.f;
  o = dpp_take<NGM_DPP_ROW_SHR(8), 0xf>(1.f, v); v *= (k >= 8) ? o : 1.f;
  o = dpp_take<NGM_DPP_ROW_BCAST15, 0xa>(1.f, v); v *= (k > pr) ? o : 1.f;
  o = dpp_take<NGM_DPP_ROW_BCAST31, 0xc>(1.f, v); v *= (k > lane - 32) ? o : 1.f;
  return v;
}
__device__ __forceinline__ float lane_value(float v, int l) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l)); }
// value of lane-1 (lane 0 keeps `ident`)
__device__ __forceinline__ float lane_prev(float v, float ident) { return dpp_take<NGM_DPP_WAVE_SHR1, 0xf>(ident, v); }
// reverse affine scan: composes x -> A + B*x from the segment END towards lower lanes.
// kr = number of samples after this one inside the ray (S-1-k); result maps Q_end to Q_before(lane).
__device__ __forceinline__ void seg_rscan_affine(float& A, float& B, int kr, int lane) {
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const float oA = __shfl_down(A, d, 64);
    const float oB = __shfl_down(B, d, 64);
    if (lane + d < 64 && kr >= d) { A = fmaf(B, oA, A); B = B * oB; }
  }
}

// idx / S for 0 <= idx < 2^24 by reciprocal multiplication, corrected at the segment borders
__device__ __forceinline__ int fdiv_idx32(int idx, float inv_s, int S) {
  int q = (int)(((float)idx + 0.5f) * inv_s);
  if (q * S > idx) --q;
  if ((q + 1) * S <= idx) ++q;
  return q;
}
// The same over the 32 samples of a tile held by lanes 0..31 (lanes 32..63 must hold the identity A = 0, B = 1), lane
// exchange by DPP only: row_shl inside the two 16-lane rows, then row 0 composes with lane 16's map where its segment runs
// past lane 15.  For a wave that has its SIMD to itself (k_field_bwd_b3): five dependent ds_bpermute round trips would
// be ~500 clocks of nothing per tile.
#define NGM_DPP_ROW_SHL(d) (0x100 + (d))
#define NGM_DPP_WAVE_SHL1 0x130
__device__ __forceinline__ void seg_rscan_affine32(float& A, float& B, int kr, int lane) {
#define NGM_RS32_STEP(d)                                                                  \
  {                                                                                       \
    const float oA = dpp_take<NGM_DPP_ROW_SHL(d), 0xf>(0.f, A), oB = dpp_take<NGM_DPP_ROW_SHL(d), 0xf>(1.f, B); \
    if (kr >= (d)) { A = fmaf(B, oA, A); B = B * oB; }                                    \
  }
  NGM_RS32_STEP(1) NGM_RS32_STEP(2) NGM_RS32_STEP(4) NGM_RS32_STEP(8)
#undef NGM_RS32_STEP
  const float A16 = lane_value(A, 16), B16 = lane_value(B, 16);
  if (lane < 16 && kr >= 16 - lane) { A = fmaf(B, A16, A); B = B * B16; }
}
// ... and over all 64 lanes: the four rows scan themselves, then row 2 composes with lane 48's map, row 1 with lane 32's
// (already extended), row 0 with lane 16's -- where the lane's segment runs past the end of its row.
__device__ __forceinline__ void seg_rscan_affine64(float& A, float& B, int kr, int lane) {
#define NGM_RS64_STEP(d)                                                                  \
  {                                                                                       \
    const float oA = dpp_take<NGM_DPP_ROW_SHL(d), 0xf>(0.f, A), oB = dpp_take<NGM_DPP_ROW_SHL(d), 0xf>(1.f, B); \
    if (kr >= (d)) { A = fmaf(B, oA, A); B = B * oB; }                                    \
  }
  NGM_RS64_STEP(1) NGM_RS64_STEP(2) NGM_RS64_STEP(4) NGM_RS64_STEP(8)
#undef NGM_RS64_STEP
#pragma unroll
  for (int row = 2; row >= 0; --row) {
    const int head = 16 * (row + 1);                       // first lane of the next row
    const float An = lane_value(A, head), Bn = lane_value(B, head);
    if ((lane >> 4) == row && kr >= head - lane) { A = fmaf(B, An, A); B = B * Bn; }
  }
}
// value of lane + 1 (lane 63 keeps `ident`)
__device__ __forceinline__ float lane_next(float v, float ident) { return dpp_take<NGM_DPP_WAVE_SHL1, 0xf>(ident, v); }

// sum over the 64 lanes (uniform result): DPP row scans, row totals carried by row_bcast, lane 63 read back
__device__ __forceinline__ float wave_sum(float v) {
  v += dpp_take<NGM_DPP_ROW_SHR(1), 0xf>(0.f, v);
  v += dpp_take<NGM_DPP_ROW_SHR(2), 0xf>(0.f, v);
  v += dpp_take<NGM_DPP_ROW_SHR(4), 0xf>(0.f, v);
  v += dpp_take<NGM_DPP_ROW_SHR(8), 0xf>(0.f, v);
  v += dpp_take<NGM_DPP_ROW_BCAST15, 0xa>(0.f, v);
  v += dpp_take<NGM_DPP_ROW_BCAST31, 0xc>(0.f, v);
  return lane_value(v, 63);
}

// loss scalars from the global sums (rm.py:1803-1871); out[0..5] = combined, termination, photometric, depth, freespace,
// tsdf.  An EMPTY selection gives NaN for its term and for `combined`, exactly as the reference's `.mean()` of an empty
// tensor does (rm.py:1803-1835; 0 * NaN = NaN for a zero weight too) -- the VALUES only: the gradient of an empty mean is
// empty, so the normalisers of the backward (k = 0 for an empty selection) already were the reference's.  The free-space /
// TSDF terms exist only with a non-zero weight (rm.py:624, 632, 1847-1871; callers zero the weights when the rays carry no gt).
// photometric gaussian_nll: the reference returns the L1 loss instead whenever the mean NLL exceeds 2 (losses.py:34-35) -- a
// decision on the GLOBAL mean, i.e. on the (all-reduced) sums; loss value and gradient seeds both follow it
__device__ __forceinline__ bool photo_nll_uses_l1(const ngm_render_cfg& rc, const float* sums) {
  const float n_m = sums[NGM_LS_PHOTO_CNT];
  return rc.photometric_mode == NGM_PHOTO_GAUSSIAN_NLL && n_m > 0 && sums[NGM_LS_PHOTO_SUM] / (3.0f * n_m) > 2.0f;   // NaN > 2 is false
}
__device__ __forceinline__ void loss_values_from_sums(const ngm_render_cfg& rc, const float* sums, float* out) {
  const float n_m = sums[NGM_LS_PHOTO_CNT], n_d = sums[NGM_LS_DEPTH_CNT], n_t = sums[NGM_LS_TERM_CNT],
              n_fs = sums[NGM_LS_FS_CNT], n_ts = sums[NGM_LS_TSDF_CNT];
  const float nan = __builtin_nanf("");
  const float lt = n_t > 0 ? sums[NGM_LS_TERM_SUM] / n_t : nan;
  float lp = n_m > 0 ? sums[NGM_LS_PHOTO_SUM] / (3.0f * n_m) : nan;
  if (photo_nll_uses_l1(rc, sums)) lp = sums[NGM_LS_PHOTO_L1_SUM] / (3.0f * n_m);       // losses.py:34-35
  const float ld = n_d > 0 ? sums[NGM_LS_DEPTH_SUM] / n_d : nan;
  const bool has_fs = rc.w_freespace != 0.f, has_ts = rc.w_tsdf != 0.f;
  const float lf = has_fs ? (n_fs > 0 ? sums[NGM_LS_FS_SUM] / n_fs : nan) : 0.f;
  const float ls = has_ts ? (n_ts > 0 ? sums[NGM_LS_TSDF_SUM] / n_ts : nan) : 0.f;
  out[1] = lt; out[2] = lp; out[3] = ld; out[4] = lf; out[5] = ls;
  float total = rc.w_termination * lt + rc.w_photometric * lp + rc.w_depth * ld;
  if (has_fs) total += rc.w_freespace * lf;
  if (has_ts) total += rc.w_tsdf * ls;
  out[0] = total;
  out[6] = 0.f; out[7] = 0.f;
}

// geometry value written over samples behind the camera (rm.py:614-622)
__device__ __forceinline__ float behind_camera_geometry(int mode) {
  return (mode == NGM_GEO_OCCUPANCY || mode == NGM_GEO_DENSITY) ? -100.0f : 1.0f;
}

// density mode (rm.py:746-749): occ_k = 1 - exp(-(t_{k+1} - t_k) relu(g_k)); the caller drops the last sample
__device__ __forceinline__ float occ_density(float g, float dl, float* docc_dg) {
  const float e = expf(-dl * fmaxf(g, 0.f));
  if (docc_dg) *docc_dg = (g > 0.f) ? dl * e : 0.f;
  return 1.0f - e;
}

// The same quantities for the compositing backward fused into the MLP backward, where a wave has its SIMD to itself and
// every instruction of this phase is exposed: one v_exp_f32 and one v_rcp_f32 instead of two IEEE divisions and two expf
// (~55 -> ~12 instructions).  With e = exp(-x), s = 1 / (1 + e): sigmoid(x) = s, sigmoid(-x) = e s, so
// nrgbd: occ = 4 e s^2, d occ / d g = gamma occ (e - 1) s; occupancy: occ = s, d occ / d g = gamma e s^2.
// |x| is clamped at 80 (e stays finite; the exact forms are 0 or 1 to fp32 there); a NaN geometry output stays NaN (fminf /
// fmaxf return the non-NaN operand: unguarded, a diverged field would render a finite, saturated occupancy where the exact
// form and the reference give NaN -- ADVICE r5).  Agrees with occ_pointwise to ~3e-7 relative.  The forward kernels use
// occ_pointwise_fwd (this form without the derivative); k_composite_bwd / k_stash_bwd differentiate the exact form
// occ_pointwise: the two differ by ~3e-7 relative, far inside the gradient bar (2e-3).
__device__ __forceinline__ float occ_clamped_arg(float x) {
  const float c = fminf(fmaxf(x, -80.0f), 80.0f);
  return (x != x) ? x : c;
}
__device__ __forceinline__ float occ_pointwise_fast(int mode, float gamma, float g, float* docc_dg) {
  const float x = occ_clamped_arg(gamma * g);
  const float e = __builtin_amdgcn_exp2f(x * -1.4426950408889634f);
  const float s = __builtin_amdgcn_rcpf(1.0f + e);
  if (mode == NGM_GEO_NRGBD) {
    const float o = 4.0f * e * s * s;
    *docc_dg = gamma * o * ((e - 1.0f) * s);
    return o;
  }
  *docc_dg = gamma * e * s * s;
  return s;
}

// The forward's form of the same (no derivative): round 5 -- the compositing kernels are bound by their vector instruction
// stream (profiles/r05_pmc_stages_sq.json), and the two IEEE divisions + two expf of the exact form are a seventh of it.
// ~3e-7 from the exact form (hardware exp2 / rcp: 1 ulp each), far inside the forward tolerance (2e-4 / 2e-5); exactly 1 at
// g = 0 (nrgbd) like the exact form.
__device__ __forceinline__ float occ_pointwise_fwd(int mode, float gamma, float g) {
  const float x = occ_clamped_arg(gamma * g);
  const float e = __builtin_amdgcn_exp2f(x * -1.4426950408889634f);
  const float s = __builtin_amdgcn_rcpf(1.0f + e);
  return (mode == NGM_GEO_NRGBD) ? 4.0f * e * s * s : s;
}

// occupancy probability of one sample (rm.py:746-762) and its derivative w.r.t. the geometry value.
// density/neus need the neighbouring sample and are handled by the callers.
__device__ __forceinline__ float occ_pointwise(int mode, float gamma, float g, float* docc_dg) {
  const float x = gamma * g;
  if (mode == NGM_GEO_NRGBD) {
    const float s = ngm_sigmoid(x), sm = ngm_sigmoid(-x);
    const float o = 4.0f * s * sm;
    if (docc_dg) *docc_dg = gamma * o * (sm - s);
    return o;
  }
  const float s = ngm_sigmoid(x);  // occupancy
  if (docc_dg) *docc_dg = gamma * s * (1.0f - s);
  return s;
}
