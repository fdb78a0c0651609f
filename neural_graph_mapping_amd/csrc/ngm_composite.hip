// HBM-bound stages of the hot path (gfx950): standalone ray sampler, volume-rendering quadrature
// forward / backward (wave segmented prefix scans over flat samples), backward of the compositing on
// the fused kernel's per-sample stash, and the sparse per-field Adam update.
#include "ngm_device.h"
#include "ngm_launch.h"
#include <algorithm>

#define WAVE_SYNC()                                        \
  do {                                                     \
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); \
    __builtin_amdgcn_wave_barrier();                       \
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); \
  } while (0)

__device__ __forceinline__ int fdiv_idx2(int idx, float inv_s, int S) {
  int q = (int)(((float)idx + 0.5f) * inv_s);
  if (q * S > idx) --q;
  if ((q + 1) * S <= idx) ++q;
  return q;
}

// ================================================================================================
// K1 standalone sampler: one thread per source element; rank-scatter into the sorted slot.
// ================================================================================================
struct SamplerArgs {
  ngm_render_cfg rc;
  ngm_rays rays;
  int S;
  float* points_cam; float* distances; float* dirs; float* points_world;
};

// one thread per source element (rays longer than SR_MAXS samples): every element draws its own jitter and, for its rank,
// up to three draws of the other stratum again
__global__ void k_sample_rays_elem(SamplerArgs a) {
#pragma clang fp contract(off)
  const int64_t total = (int64_t)a.rays.F * a.rays.R * a.S;
  const int S_c = a.rc.num_samples_coarse, S_g = a.S - S_c;
  const uint64_t poff = philox_launch_offset(a.rays);
  for (int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; g < total; g += (int64_t)gridDim.x * blockDim.x) {
    const int64_t ray = g / a.S;
    const int e = (int)(g - ray * a.S);
    const RayGeom rg = ray_geom(a.rc, a.rays, ray, S_g > 0);
    float t; int rank;
    sample_rank(a.rc, a.rays, poff, rg, ray, e, S_c, S_g, &t, &rank);
    const int64_t o = ray * a.S + rank;
    if (a.distances) a.distances[o] = t;
    if (a.points_cam) {
      a.points_cam[3 * o] = rg.dx * t; a.points_cam[3 * o + 1] = rg.dy * t;
      a.points_cam[3 * o + 2] = rg.dz * t;
    }
    if (a.points_world) {
      float wx, wy, wz;
      sample_world_point(a.rays, rg, ray, t, &wx, &wy, &wz);
      a.points_world[3 * o] = wx; a.points_world[3 * o + 1] = wy; a.points_world[3 * o + 2] = wz;
    }
    if (a.dirs && e == 0) { a.dirs[3 * ray] = rg.dx; a.dirs[3 * ray + 1] = rg.dy; a.dirs[3 * ray + 2] = rg.dz; }
  }
}

// A wave per batch of whole rays (as the fused forward does it): the ray-level quantities once per ray, ONE draw per element
// into an LDS plane, then the closed-form ranks read their neighbours' draws from there -- the per-element kernel above
// spends four Philox blocks and a ray set-up (three IEEE divisions, a root) on every sample.  Same arithmetic, same bits.
#define SR_MAXS 1024
struct SampWaveLds { float u[SR_MAXS]; float rt[32][8]; };
__global__ __launch_bounds__(NGM_BLOCK) void k_sample_rays(SamplerArgs a, int rays_per_wave) {
#pragma clang fp contract(off)
  __shared__ SampWaveLds lds[NGM_WAVES_PER_BLOCK];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  SampWaveLds& wl = lds[wave];
  const int64_t N = (int64_t)a.rays.F * a.rays.R;
  const int64_t gw = (int64_t)blockIdx.x * NGM_WAVES_PER_BLOCK + wave;
  const int64_t r_beg = min(N, gw * rays_per_wave), r_end = min(N, r_beg + rays_per_wave);
  const int S = a.S, S_c = a.rc.num_samples_coarse, S_g = S - S_c;
  const float inv_s = 1.0f / (float)S;
  const uint64_t poff = philox_launch_offset(a.rays);
  const int BR = max(1, min(32, SR_MAXS / S));
  for (int64_t rb = r_beg; rb < r_end; rb += BR) {
    const int nb = (int)min<int64_t>(BR, r_end - rb);
    const int nsamp = nb * S;
    if (lane < nb) {
      const int64_t ray = rb + lane;
      const RayGeom rg = ray_geom(a.rc, a.rays, ray, S_g > 0);
      float* rt = wl.rt[lane];
      rt[0] = rg.dx; rt[1] = rg.dy; rt[2] = rg.dz; rt[3] = rg.near; rt[4] = rg.far; rt[5] = rg.gnear; rt[6] = rg.gfar; rt[7] = rg.gt;
      if (a.dirs) { a.dirs[3 * ray] = rg.dx; a.dirs[3 * ray + 1] = rg.dy; a.dirs[3 * ray + 2] = rg.dz; }
    }
    jitter_fill(a.rays, poff, 0, rb, nb, S_c, wl.u, S, lane);                  // one Philox block per four elements
    if (S_g > 0) jitter_fill(a.rays, poff, 1, rb, nb, S_g, wl.u + S_c, S, lane);
    WAVE_SYNC();
    for (int idx = lane; idx < nsamp; idx += 64) {
      const int rl = fdiv_idx2(idx, inv_s, S), e = idx - rl * S;
      const int64_t ray = rb + rl;
      const float* rt = wl.rt[rl];
      RayGeom rg;
      rg.dx = rt[0]; rg.dy = rt[1]; rg.dz = rt[2]; rg.near = rt[3]; rg.far = rt[4]; rg.gnear = rt[5]; rg.gfar = rt[6]; rg.gt = rt[7];
      float t; int rank;
      sample_rank(a.rc, a.rays, poff, rg, ray, e, S_c, S_g, &t, &rank, wl.u + rl * S);
      const int64_t o = ray * S + rank;
      if (a.distances) a.distances[o] = t;
      if (a.points_cam) { a.points_cam[3 * o] = rg.dx * t; a.points_cam[3 * o + 1] = rg.dy * t; a.points_cam[3 * o + 2] = rg.dz * t; }
      if (a.points_world) {
        float wx, wy, wz;
        sample_world_point(a.rays, rg, ray, t, &wx, &wy, &wz);
        a.points_world[3 * o] = wx; a.points_world[3 * o + 1] = wy; a.points_world[3 * o + 2] = wz;
      }
    }
    WAVE_SYNC();
  }
}

static int comp_grid(int64_t N, int S, int* rays_per_wave);
int ngm_launch_sampler(const ngm_render_cfg* rc, const ngm_rays* rays, int S, float* points_cam, float* distances,
                       float* dirs, float* points_world, hipStream_t st) {
  NgmProfScope prof_(NGM_K_SAMPLER, st);
  SamplerArgs a{*rc, *rays, S, points_cam, distances, dirs, points_world};
  const int64_t N = (int64_t)rays->F * rays->R;
  if (S <= SR_MAXS) {
    int rpw;
    const int blocks = comp_grid(N, S, &rpw);
    hipLaunchKernelGGL(k_sample_rays, dim3(std::max(blocks, 1)), dim3(NGM_BLOCK), 0, st, a, rpw);
    return 0;
  }
  const int64_t total = N * S;
  const int blocks = (int)std::min<int64_t>((total + 255) / 256, 4096);
  hipLaunchKernelGGL(k_sample_rays_elem, dim3(std::max(blocks, 1)), dim3(256), 0, st, a);
  return 0;
}

// ------------------------------------------------------------------------------------------------
// Camera.sample_ijs_uniform, weighted-bin branch (camera.py:277-289): per sample, a bin drawn from the ray's bin weights
// (bin = first index whose running sum + 1e-3 reaches the first draw: torch.cumsum, torch.searchsorted(right=False)), then
// a uniform position inside it (second draw).  The running sum is torch.cumsum's on the CPU: accumulated sequentially in
// fp64, each prefix rounded to fp32 (ATen's cumsum_cpu_kernel, acc_type<float> = double) -- the reference's CUDA scan sums
// in fp32 in tree order; the bin of a sample changes only when its draw falls within that rounding of a cumulative weight.
// NOT sorted: the reference does not sort this branch's output either.  A draw beyond the last cumulative weight (weights
// that do not sum to 1 - 1e-3 or more: torch.gather is out of range there and raises) takes the last bin.
// A wave per batch of whole rays: lane = ray builds the ray's direction, its cumulative weights (+ 1e-3) and its boundaries
// in LDS once; lane = sample then draws (in-kernel: ONE Philox block per sample, words 0 and 1), finds its bin by bisection
// (= searchsorted's lower bound on a non-decreasing sequence) and reads the bin from LDS.  Rays with more bins than the
// LDS budget holds go through the one-thread-per-sample kernel below (sequential search of the running sum).
// ------------------------------------------------------------------------------------------------
struct WeightedSamplerArgs {
  ngm_render_cfg rc;
  ngm_rays rays;
  int S, B;
  const float* boundaries; const float* weights;
  float* points_cam; float* distances; float* dirs;
};
__device__ __forceinline__ void weighted_draws(const ngm_rays& rays, uint64_t poff, int64_t ray, int S, int e, float* u_bin, float* u_off) {
  if (rays.u_coarse) { *u_bin = rays.u_coarse[ray * S + e]; *u_off = rays.u_guided[ray * S + e]; }
  else philox_uniform2(rays.philox_seed, poff, (uint64_t)(ray * S + e), 0u, u_bin, u_off);
}
__global__ void k_sample_rays_weighted_elem(WeightedSamplerArgs a) {
#pragma clang fp contract(off)
  const int64_t total = (int64_t)a.rays.F * a.rays.R * a.S;
  const uint64_t poff = philox_launch_offset(a.rays);
  for (int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; g < total; g += (int64_t)gridDim.x * blockDim.x) {
    const int64_t ray = g / a.S;
    const int e = (int)(g - ray * a.S);
    const RayGeom rg = ray_geom(a.rc, a.rays, ray, false);
    float u_bin, u_off;
    weighted_draws(a.rays, poff, ray, a.S, e, &u_bin, &u_off);
    const float* w = a.weights + ray * a.B;
    const float* bd = a.boundaries + ray * (a.B + 1);
    double cum = 0.0;
    int bin = a.B - 1;
    for (int b = 0; b < a.B; ++b) {
      cum += (double)w[b];
      if ((float)cum + 1e-3f >= u_bin) { bin = b; break; }
    }
    const float start = bd[bin], size = bd[bin + 1] - start;
    const float t = start + size * u_off;
    if (a.distances) a.distances[g] = t;
    if (a.points_cam) { a.points_cam[3 * g] = rg.dx * t; a.points_cam[3 * g + 1] = rg.dy * t; a.points_cam[3 * g + 2] = rg.dz * t; }
    if (a.dirs && e == 0) { a.dirs[3 * ray] = rg.dx; a.dirs[3 * ray + 1] = rg.dy; a.dirs[3 * ray + 2] = rg.dz; }
  }
}
// per wave: [RB][B] cumulative weights + 1e-3, [RB][B + 1] boundaries, [RB][4] directions
__global__ __launch_bounds__(NGM_BLOCK) void k_sample_rays_weighted(WeightedSamplerArgs a, int rays_per_wave, int RB) {
#pragma clang fp contract(off)
  extern __shared__ __attribute__((aligned(16))) float sw_lds[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int S = a.S, B = a.B;
  float* const cw = sw_lds + (size_t)wave * RB * (2 * B + 1 + 4);
  float* const bdl = cw + (size_t)RB * B;
  float* const dir = bdl + (size_t)RB * (B + 1);
  const int64_t N = (int64_t)a.rays.F * a.rays.R;
  const int64_t gw = (int64_t)blockIdx.x * NGM_WAVES_PER_BLOCK + wave;
  const int64_t r_beg = min(N, gw * rays_per_wave), r_end = min(N, r_beg + rays_per_wave);
  const float inv_s = 1.0f / (float)S;
  const uint64_t poff = philox_launch_offset(a.rays);
  for (int64_t rb = r_beg; rb < r_end; rb += RB) {
    const int nb = (int)min<int64_t>(RB, r_end - rb);
    if (lane < nb) {
      const int64_t ray = rb + lane;
      const RayGeom rg = ray_geom(a.rc, a.rays, ray, false);
      dir[4 * lane] = rg.dx; dir[4 * lane + 1] = rg.dy; dir[4 * lane + 2] = rg.dz;
      if (a.dirs) { a.dirs[3 * ray] = rg.dx; a.dirs[3 * ray + 1] = rg.dy; a.dirs[3 * ray + 2] = rg.dz; }
      const float* w = a.weights + ray * B;
      const float* bd = a.boundaries + ray * (B + 1);
      double cum = 0.0;
      for (int b = 0; b < B; ++b) {
        cum += (double)w[b];
        cw[lane * B + b] = (float)cum + 1e-3f;
        bdl[lane * (B + 1) + b] = bd[b];
      }
      bdl[lane * (B + 1) + B] = bd[B];
    }
    WAVE_SYNC();
    const int nsamp = nb * S;
    for (int idx = lane; idx < nsamp; idx += 64) {
      const int rl = fdiv_idx2(idx, inv_s, S), e = idx - rl * S;
      const int64_t ray = rb + rl;
      float u_bin, u_off;
      weighted_draws(a.rays, poff, ray, S, e, &u_bin, &u_off);
      const float* c = cw + rl * B;
      int lo = 0, hi = B;                       // lower bound: first b with c[b] >= u_bin
      while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (c[mid] < u_bin) lo = mid + 1; else hi = mid;
      }
      const int bin = min(lo, B - 1);
      const float start = bdl[rl * (B + 1) + bin], size = bdl[rl * (B + 1) + bin + 1] - start;
      const float t = start + size * u_off;
      const int64_t g = ray * S + e;
      if (a.distances) a.distances[g] = t;
      if (a.points_cam) {
        const float dx = dir[4 * rl], dy = dir[4 * rl + 1], dz = dir[4 * rl + 2];
        a.points_cam[3 * g] = dx * t; a.points_cam[3 * g + 1] = dy * t; a.points_cam[3 * g + 2] = dz * t;
      }
    }
    WAVE_SYNC();
  }
}
static int comp_grid(int64_t N, int S, int* rays_per_wave);
int ngm_launch_sampler_weighted(const ngm_render_cfg* rc, const ngm_rays* rays, int S, int B, const float* boundaries,
                                const float* weights, float* points_cam, float* distances, float* dirs, hipStream_t st) {
  NgmProfScope prof_(NGM_K_SAMPLER, st);
  WeightedSamplerArgs a{*rc, *rays, S, B, boundaries, weights, points_cam, distances, dirs};
  const int64_t N = (int64_t)rays->F * rays->R;
  const int64_t per_ray = 2 * (int64_t)B + 1 + 4;                 // floats of LDS per ray
  const int64_t budget = 4096;                                    // floats per wave (16 KB; 4 waves per block, 2+ blocks per CU)
  if (per_ray <= budget) {
    int RB = (int)std::min<int64_t>(64, budget / per_ray);
    int rpw;
    const int blocks = comp_grid(N, S, &rpw);
    if (rpw < RB) RB = std::max(1, rpw);
    const size_t lds = (size_t)NGM_WAVES_PER_BLOCK * RB * per_ray * sizeof(float);
    hipLaunchKernelGGL(k_sample_rays_weighted, dim3(std::max(blocks, 1)), dim3(NGM_BLOCK), lds, st, a, rpw, RB);
    return 0;
  }
  const int64_t total = N * S;
  const int blocks = (int)std::min<int64_t>((total + 255) / 256, 8192);
  hipLaunchKernelGGL(k_sample_rays_weighted_elem, dim3(std::max(blocks, 1)), dim3(256), 0, st, a);
  return 0;
}

// ================================================================================================
// K4 quadrature forward (rm.py:709-799), all four geometry modes.
// Each wave owns a contiguous run of rays and walks their flat samples 64 at a time; transmittance
// is a segmented inclusive product scan carried from step to step; the per-ray sums are segmented
// sum scans whose segment tails accumulate into a small LDS ray table.
// ================================================================================================
#define CQ_BR 32
#define CQ_MAXS 1024
// KEEP: colours and depths of the batch stay in LDS planes for the variance pass (batches of 256 samples, so that the planes
// cost no occupancy: 6.5 KB per wave); !KEEP: rays longer than 256 samples re-read them from memory in batches of 1024.
template <bool KEEP>
struct CompWaveLds {
  static constexpr int MAXS = KEEP ? 256 : CQ_MAXS;
  float ra[CQ_BR][12];
  float wbuf[MAXS];
  float cbuf[KEEP ? 4 : 1][KEEP ? MAXS : 1];       // colour (3), depth
};

struct Rgb3 { float x, y, z; };   // 12-byte element: one global_load_dwordx3 per lane instead of three strided dword loads
__device__ __forceinline__ Rgb3 load_rgb(const float* colors, int64_t g) { return *reinterpret_cast<const Rgb3*>(colors + 3 * g); }

// packed sources (eval path = _render_ijs(use_vmap=False)): the (N,S,4) field outputs, or -- fused with the kNN blend
// (k_knn_blend, models.py:384-401) -- the (point, neighbour) pair records it would have been formed from
__device__ __forceinline__ bool comp_packed(const CompositeArgs& a) { return a.out4 || a.pair_field; }
static inline bool comp_packed_host(const CompositeArgs& a) { return a.out4 || a.pair_field; }
__device__ __forceinline__ float4 packed_out4(const CompositeArgs& a, int64_t g) {
  if (!a.pair_field) return a.out4[g];
  const int K = a.pair_K;
  float4 o = make_float4(a.outside_value, a.outside_value, a.outside_value, a.outside_value);   // models.py:401
  if (a.pair_field[g * K] >= 0) {
    o = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int k = 0; k < K; ++k) {
      const float w = a.pair_w[g * K + k];
      const float4 v = a.pair_out[g * K + k];
      o.x = fmaf(w, v.x, o.x); o.y = fmaf(w, v.y, o.y); o.z = fmaf(w, v.z, o.z); o.w = fmaf(w, v.w, o.w);
    }
  }
  return o;
}
// camera-frame z of sample g of `ray`: stored, or direction.z * distance as the sampler forms it
__device__ __forceinline__ float packed_pz(const CompositeArgs& a, int64_t ray, int64_t g) {
#pragma clang fp contract(off)
  if (a.pcam) return a.pcam[3 * g + 2];
  return a.ray_dir[3 * ray + 2] * a.dists[g];
}
// samples behind the camera get a constant (rm.py:614-622)
__device__ __forceinline__ float packed_geom(const CompositeArgs& a, const float4& o, float pz) {
  if (a.rc.overwrite_behind_camera && pz > 0.f) return behind_camera_geometry(a.rc.geometry_mode);
  return o.w;
}
__device__ __forceinline__ float geom_at(const CompositeArgs& a, int64_t ray, int64_t g) {
  if (!comp_packed(a)) return a.geoms[g];
  const float pz = (a.rc.overwrite_behind_camera) ? packed_pz(a, ray, g) : 0.f;
  if (a.rc.overwrite_behind_camera && pz > 0.f) return behind_camera_geometry(a.rc.geometry_mode);
  return a.pair_field ? packed_out4(a, g).w : a.out4[g].w;
}
// gself (optional): the geometry value of sample k itself, already formed by the caller
__device__ __forceinline__ float occ_at(const CompositeArgs& a, int64_t ray, int k, int S, float* docc = nullptr,
                                        const float* gself = nullptr) {
  const int mode = a.rc.geometry_mode;
  const int64_t g = ray * S + k;
  const float gm = gself ? *gself : geom_at(a, ray, g);
  if (mode == NGM_GEO_NRGBD || mode == NGM_GEO_OCCUPANCY)
    return docc ? occ_pointwise(mode, a.rc.geometry_factor, gm, docc) : occ_pointwise_fwd(mode, a.rc.geometry_factor, gm);   // forward: one form everywhere
  if (k >= S - 1) return 0.f;   // density / neus drop the last sample (rm.py:749,758)
  if (mode == NGM_GEO_DENSITY) return occ_density(gm, a.dists[g + 1] - a.dists[g], docc);
  const float isd = a.isds ? a.isds[ray] : 1.0f;
  const float t0 = ngm_sigmoid(isd * a.rc.geometry_factor * gm);
  const float t1 = ngm_sigmoid(isd * a.rc.geometry_factor * geom_at(a, ray, g + 1));
  return fmaxf((t0 - t1) / (t0 + 1e-5f), 0.f);
}

// NeuS (rm.py:753-758): occ_k = max((u_k - u_{k+1}) / (u_k + 1e-5), 0), u = sigmoid(isd * gamma * g).
// Partial derivatives w.r.t. g_k, g_{k+1} and the ray's inverse standard deviation.
__device__ __forceinline__ void neus_partials(const CompositeArgs& a, int64_t ray, int k, int S, float* d_self, float* d_next,
                                              float* d_isd) {
  *d_self = *d_next = *d_isd = 0.f;
  if (k >= S - 1) return;
  const int64_t g = ray * S + k;
  const float isd = a.isds ? a.isds[ray] : 1.0f, gam = a.rc.geometry_factor;
  const float g0 = geom_at(a, ray, g), g1 = geom_at(a, ray, g + 1);
  const float u0 = ngm_sigmoid(isd * gam * g0), u1 = ngm_sigmoid(isd * gam * g1);
  const float den = u0 + 1e-5f;
  if ((u0 - u1) / den <= 0.f) return;                     // clamp_min(…, 0): zero gradient (torch: grad 0 at the kink too)
  const float P = (1e-5f + u1) / (den * den), M = -1.0f / den;
  const float s0 = u0 * (1.0f - u0), s1 = u1 * (1.0f - u1);
  *d_self = P * isd * gam * s0;
  *d_next = M * isd * gam * s1;
  *d_isd = P * gam * g0 * s0 + M * gam * g1 * s1;
}

// Pair-record source (PK = neighbours per point): the blended outputs and camera-frame z of CH consecutive wave steps,
// gathered together.  A step's loads form a dependent chain (pair_field -> weights and outputs of the inside points) and
// the steps of a ray form another (the transmittance carry), so a wave that loads step by step sits out two memory
// latencies per 64 samples; here the CH first-level loads are issued at once, then all second-level ones (outside points
// read a dummy line: no branch between the loads), then the steps run from registers.
template <int PK, int CH>
__device__ __forceinline__ void pairs_gather(const CompositeArgs& a, int64_t rb, int S, float inv_s, int nsamp, int base0,
                                             int lane, float4 (&o4)[CH], float (&pz)[CH]) {
#pragma clang fp contract(off)
  constexpr int PKn = PK > 0 ? PK : 1;
  int pf[CH];
  float tt[CH], dz[CH];
  int64_t gg[CH];
#pragma unroll
  for (int c = 0; c < CH; ++c) {
    const int idx = base0 + 64 * c + lane;
    const bool valid = idx < nsamp;
    const int idc = valid ? idx : 0;
    const int rl = fdiv_idx2(idc, inv_s, S);
    gg[c] = rb * S + idc;
    const int f = a.pair_field[gg[c] * PKn];
    pf[c] = valid ? f : -1;
    tt[c] = a.dists[gg[c]];
    dz[c] = a.ray_dir[3 * (rb + rl) + 2];
  }
  float w[CH][PKn];
  float4 v[CH][PKn];
#pragma unroll
  for (int c = 0; c < CH; ++c)
#pragma unroll
    for (int k = 0; k < PKn; ++k) {
      const int64_t j = pf[c] >= 0 ? gg[c] * PKn + k : 0;
      w[c][k] = a.pair_w[j];
      v[c][k] = a.pair_out[j];
    }
#pragma unroll
  for (int c = 0; c < CH; ++c) {
    float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int k = 0; k < PKn; ++k) {      // the blend of k_knn_blend / packed_out4, same order
      o.x = fmaf(w[c][k], v[c][k].x, o.x); o.y = fmaf(w[c][k], v[c][k].y, o.y);
      o.z = fmaf(w[c][k], v[c][k].z, o.z); o.w = fmaf(w[c][k], v[c][k].w, o.w);
    }
    o4[c] = pf[c] >= 0 ? o : make_float4(a.outside_value, a.outside_value, a.outside_value, a.outside_value);   // models.py:401
    pz[c] = dz[c] * tt[c];
  }
}

// PK > 0: the packed source is the pair records of the kNN evaluation with PK neighbours per point (ngm_render_eval_knn)
template <bool KEEP, int PK>
__global__ __launch_bounds__(NGM_BLOCK) void k_composite_fwd(CompositeArgs a, int rays_per_wave) {
  __shared__ CompWaveLds<KEEP> lds[NGM_WAVES_PER_BLOCK];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  CompWaveLds<KEEP>& wl = lds[wave];
  const int64_t gw = (int64_t)blockIdx.x * NGM_WAVES_PER_BLOCK + wave;
  const int64_t r_beg = min(a.N, gw * rays_per_wave), r_end = min(a.N, r_beg + rays_per_wave);
  const int S = a.S;
  const int mode = a.rc.geometry_mode;
  const int S_eff = (mode == NGM_GEO_DENSITY || mode == NGM_GEO_NEUS) ? S - 1 : S;
  const float inv_s = 1.0f / (float)S;
  const bool packed = comp_packed(a);
  constexpr int CH = PK > 4 ? 2 : PK > 0 ? 4 : 1;    // wave steps gathered together (pair-record source; 5..8 neighbours: registers)
  const int BR = max(1, min(CQ_BR, CompWaveLds<KEEP>::MAXS / S));
  for (int64_t rb = r_beg; rb < r_end; rb += BR) {
    const int nb = (int)min<int64_t>(BR, r_end - rb);
    const int nsamp = nb * S;
    if (lane < nb) {
#pragma unroll
      for (int c = 0; c < 12; ++c) wl.ra[lane][c] = 0.f;
    }
    WAVE_SYNC();
    float carry = 1.0f;
    // ---- pass 1: weights, means
    auto step1 = [&](int base, const float4& o4g, float pzg) __attribute__((always_inline)) {
      const int idx = base + lane;
      const bool valid = idx < nsamp;
      const int rl = valid ? fdiv_idx2(idx, inv_s, S) : 0;
      const int k = valid ? idx - rl * S : 0;
      const int64_t g = (rb + rl) * S + k;
      const bool act = valid && k < S_eff;
      float4 o4 = o4g;
      float pz = pzg, gself = 0.f;
      if constexpr (PK == 0) {
        if (act && packed) { o4 = packed_out4(a, g); pz = packed_pz(a, rb + rl, g); }
      }
      if (act && packed) gself = packed_geom(a, o4, pz);
      const float occ = act ? occ_at(a, rb + rl, k, S, nullptr, packed ? &gself : nullptr) : 0.f;
      float q = seg_scan_mul(1.0f - occ, k, lane);
      if (k > lane) q *= carry;
      const float up = lane_prev(q, carry);
      const float T_excl = (k == 0) ? 1.0f : (lane == 0 ? carry : up);
      carry = lane_value(q, 63);
      const float w = act ? occ * T_excl : 0.f;
      float c0 = 0, c1 = 0, c2 = 0, dp = 0;
      if (act) {
        if (packed) { c0 = a.rc.color_factor * o4.x; c1 = a.rc.color_factor * o4.y; c2 = a.rc.color_factor * o4.z; dp = -pz; }
        else { const Rgb3 c = load_rgb(a.colors, g); c0 = c.x; c1 = c.y; c2 = c.z; dp = a.depths[g]; }
      }
      if (valid) {
        wl.wbuf[idx] = w;
        if constexpr (KEEP) { wl.cbuf[0][idx] = c0; wl.cbuf[1][idx] = c1; wl.cbuf[2][idx] = c2; wl.cbuf[3][idx] = dp; }
      }
      if (act && a.weights) a.weights[(rb + rl) * S_eff + k] = w;
      const float s0 = seg_scan_add(w * c0, k, lane), s1 = seg_scan_add(w * c1, k, lane),
                  s2 = seg_scan_add(w * c2, k, lane), s3 = seg_scan_add(w * dp, k, lane),
                  s4 = seg_scan_add(w, k, lane);
      const bool tail = valid && (k == S - 1 || lane == 63 || idx == nsamp - 1);
      if (tail) { float* ra = wl.ra[rl]; ra[0] += s0; ra[1] += s1; ra[2] += s2; ra[3] += s3; ra[4] += s4; }
      WAVE_SYNC();
    };
    for (int base0 = 0; base0 < nsamp; base0 += 64 * CH) {
      float4 o4c[CH];
      float pzc[CH];
      if constexpr (PK > 0) pairs_gather<PK, CH>(a, rb, S, inv_s, nsamp, base0, lane, o4c, pzc);
      else { o4c[0] = make_float4(0.f, 0.f, 0.f, 0.f); pzc[0] = 0.f; }
#pragma unroll
      for (int c = 0; c < CH; ++c)
        if (base0 + 64 * c < nsamp) step1(base0 + 64 * c, o4c[c], pzc[c]);
    }
    // ---- pass 2: the variances around the finished means (rm.py:781-790)
    auto step2 = [&](int base, const float4& o4g, float pzg) __attribute__((always_inline)) {
      const int idx = base + lane;
      const bool valid = idx < nsamp;
      const int rl = valid ? fdiv_idx2(idx, inv_s, S) : 0;
      const int k = valid ? idx - rl * S : 0;
      const int64_t g = (rb + rl) * S + k;
      const bool act = valid && k < S_eff;
      const float* ra = wl.ra[rl];
      const float w = valid ? wl.wbuf[idx] : 0.f;
      float e0 = 0, e1 = 0, e2 = 0, e3 = 0;
      if constexpr (KEEP) {
        // from the LDS planes: no second read of colours / depths
        if (act) { e0 = ra[0] - wl.cbuf[0][idx]; e1 = ra[1] - wl.cbuf[1][idx]; e2 = ra[2] - wl.cbuf[2][idx]; e3 = ra[3] - wl.cbuf[3][idx]; }
      } else if (act) {
        if (packed) {
          float4 o = o4g;
          float pz = pzg;
          if constexpr (PK == 0) { o = packed_out4(a, g); pz = packed_pz(a, rb + rl, g); }
          e0 = ra[0] - a.rc.color_factor * o.x; e1 = ra[1] - a.rc.color_factor * o.y; e2 = ra[2] - a.rc.color_factor * o.z;
          e3 = ra[3] + pz;
        } else { const Rgb3 c = load_rgb(a.colors, g); e0 = ra[0] - c.x; e1 = ra[1] - c.y; e2 = ra[2] - c.z; e3 = ra[3] - a.depths[g]; }
      }
      const float v0 = seg_scan_add(w * (e0 * e0), k, lane), v1 = seg_scan_add(w * (e1 * e1), k, lane),
                  v2 = seg_scan_add(w * (e2 * e2), k, lane), v3 = seg_scan_add(w * (e3 * e3), k, lane);
      const bool tail = valid && (k == S - 1 || lane == 63 || idx == nsamp - 1);
      WAVE_SYNC();
      if (tail) { float* rw = wl.ra[rl]; rw[5] += v0; rw[6] += v1; rw[7] += v2; rw[8] += v3; }
      WAVE_SYNC();
    };
    for (int base0 = 0; base0 < nsamp; base0 += 64 * CH) {
      float4 o4c[CH];
      float pzc[CH];
      if constexpr (PK > 0 && !KEEP) pairs_gather<PK, CH>(a, rb, S, inv_s, nsamp, base0, lane, o4c, pzc);
      else {
#pragma unroll
        for (int c = 0; c < CH; ++c) { o4c[c] = make_float4(0.f, 0.f, 0.f, 0.f); pzc[c] = 0.f; }
      }
#pragma unroll
      for (int c = 0; c < CH; ++c)
        if (base0 + 64 * c < nsamp) step2(base0 + 64 * c, o4c[c], pzc[c]);
    }
    if (lane < nb) {
      const int64_t ray = rb + lane;
      const float* ra = wl.ra[lane];
      if (a.rgbd) reinterpret_cast<float4*>(a.rgbd)[ray] = make_float4(ra[0], ra[1], ra[2], ra[3]);
      if (a.C) { a.C[3 * ray] = ra[0]; a.C[3 * ray + 1] = ra[1]; a.C[3 * ray + 2] = ra[2]; }
      if (a.D) a.D[ray] = ra[3];
      if (a.Cv) { a.Cv[3 * ray] = ra[5]; a.Cv[3 * ray + 1] = ra[6]; a.Cv[3 * ray + 2] = ra[7]; }
      if (a.Dv) a.Dv[ray] = ra[8];
      if (a.term) a.term[ray] = 1.0f - (1.0f - ra[4]);
    }
    WAVE_SYNC();
  }
}

// ---- whole-ray variant: S a multiple of 64, pointwise geometry modes (nrgbd, occupancy) ------------------------------------
// Every wave step then lies inside one ray (k % 64 == lane), so the segment conditions of the scans above are what the DPP
// bounds already enforce: the scans are the same lane exchanges without the compare / select per step (bit for bit the
// same sums), the per-ray sums stay in registers, the planes of the ray (weights, colours, depth) stay in LDS for the
// variance pass, and the loads of CH steps are issued together.  A fifth of the generic kernel's instructions per sample.
__device__ __forceinline__ float wave_scan_add(float v) {
  v += dpp_take<NGM_DPP_ROW_SHR(1), 0xf>(0.f, v); v += dpp_take<NGM_DPP_ROW_SHR(2), 0xf>(0.f, v);
  v += dpp_take<NGM_DPP_ROW_SHR(4), 0xf>(0.f, v); v += dpp_take<NGM_DPP_ROW_SHR(8), 0xf>(0.f, v);
  v += dpp_take<NGM_DPP_ROW_BCAST15, 0xa>(0.f, v); v += dpp_take<NGM_DPP_ROW_BCAST31, 0xc>(0.f, v);
  return v;
}
__device__ __forceinline__ float wave_scan_mul(float v) {
  v *= dpp_take<NGM_DPP_ROW_SHR(1), 0xf>(1.f, v); v *= dpp_take<NGM_DPP_ROW_SHR(2), 0xf>(1.f, v);
  v *= dpp_take<NGM_DPP_ROW_SHR(4), 0xf>(1.f, v); v *= dpp_take<NGM_DPP_ROW_SHR(8), 0xf>(1.f, v);
  v *= dpp_take<NGM_DPP_ROW_BCAST15, 0xa>(1.f, v); v *= dpp_take<NGM_DPP_ROW_BCAST31, 0xc>(1.f, v);
  return v;
}

// N sum scans at once, value-minor inside a step: a register's DPP read then sits N - 1 instructions behind its write and
// needs no wait states
template <int N>
__device__ __forceinline__ void wave_scan_add_n(float (&v)[N]) {
#define NGM_WSA_STEP(CTRL, MASK) _Pragma("unroll") for (int i_ = 0; i_ < N; ++i_) v[i_] += dpp_take<CTRL, MASK>(0.f, v[i_]);
  NGM_WSA_STEP(NGM_DPP_ROW_SHR(1), 0xf) NGM_WSA_STEP(NGM_DPP_ROW_SHR(2), 0xf) NGM_WSA_STEP(NGM_DPP_ROW_SHR(4), 0xf)
  NGM_WSA_STEP(NGM_DPP_ROW_SHR(8), 0xf) NGM_WSA_STEP(NGM_DPP_ROW_BCAST15, 0xa) NGM_WSA_STEP(NGM_DPP_ROW_BCAST31, 0xc)
#undef NGM_WSA_STEP
}

struct StepIn { float c0, c1, c2, dp, gm; };
// SRC 0: separate tensors; 1..8: pair records with SRC neighbours per point; 9: packed (N,S,4) outputs + camera-frame points
template <int SRC, int CH>
__device__ __forceinline__ void whole_gather(const CompositeArgs& a, int64_t ray, int S, int base0, int lane, StepIn (&in)[CH]) {
  if constexpr (SRC >= 1 && SRC <= 8) {
    float4 o4[CH];
    float pz[CH];
    pairs_gather<SRC, CH>(a, ray, S, 1.0f / (float)S, S, base0, lane, o4, pz);
#pragma unroll
    for (int c = 0; c < CH; ++c)
      in[c] = StepIn{a.rc.color_factor * o4[c].x, a.rc.color_factor * o4[c].y, a.rc.color_factor * o4[c].z, -pz[c],
                     packed_geom(a, o4[c], pz[c])};
  } else {
#pragma unroll
    for (int c = 0; c < CH; ++c) {
      const int k = min(base0 + 64 * c + lane, S - 1);        // steps past the ray's end re-read its last sample (unused)
      const int64_t g = ray * S + k;
      if constexpr (SRC == 9) {
        const float4 o = a.out4[g];
        const float pz = a.pcam[3 * g + 2];
        in[c] = StepIn{a.rc.color_factor * o.x, a.rc.color_factor * o.y, a.rc.color_factor * o.z, -pz, packed_geom(a, o, pz)};
      } else {
        const Rgb3 col = load_rgb(a.colors, g);
        in[c] = StepIn{col.x, col.y, col.z, a.depths[g], a.geoms[g]};
      }
    }
  }
}

template <int SRC>
__global__ __launch_bounds__(NGM_BLOCK) void k_composite_fwd_whole(CompositeArgs a, int rays_per_wave) {
  extern __shared__ __attribute__((aligned(16))) float cw_lds[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int S = a.S;
  float* const pw = cw_lds + (size_t)wave * 5 * S;      // planes: weight, colour (3), depth
  float* const p0 = pw + S; float* const p1 = p0 + S; float* const p2 = p1 + S; float* const p3 = p2 + S;
  const int64_t gw = (int64_t)blockIdx.x * NGM_WAVES_PER_BLOCK + wave;
  const int64_t r_beg = min(a.N, gw * rays_per_wave), r_end = min(a.N, r_beg + rays_per_wave);
  const int mode = a.rc.geometry_mode;
  constexpr int CH = (SRC >= 1 && SRC <= 4) ? 5 : (SRC >= 5 && SRC <= 8) ? 2 : 4;
  // most wave steps of an image lie outside every field: all 64 geometry values are the constant outside value, and so is
  // their occupancy -- computed once here (two expf, two IEEE divisions otherwise, per sample)
  const float gm_ref = a.outside_value;
  const float occ_ref = occ_pointwise_fwd(mode, a.rc.geometry_factor, gm_ref);
  for (int64_t ray = r_beg; ray < r_end; ++ray) {
    // per-LANE partial sums over the ray's steps, ONE cross-lane reduction per ray and sum (a DPP scan per step and sum was a
    // third of this kernel's vector instructions: it is VALU-bound, profiles/r05_pmc_stages_sq.json)
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f, s4 = 0.f;
    float carry = 1.0f;
    for (int base0 = 0; base0 < S; base0 += 64 * CH) {
      StepIn in[CH];
      whole_gather<SRC, CH>(a, ray, S, base0, lane, in);
#pragma unroll
      for (int c = 0; c < CH; ++c) {
        const int k = base0 + 64 * c + lane;
        if (base0 + 64 * c < S) {
          const bool all_ref = __builtin_amdgcn_ballot_w64(in[c].gm != gm_ref) == 0ull;      // (wave-uniform; same function, same input)
          const float occ = all_ref ? occ_ref : occ_pointwise_fwd(mode, a.rc.geometry_factor, in[c].gm);
          float q = wave_scan_mul(1.0f - occ);
          q *= carry;                                              // (carry = 1 in the ray's first step)
          const float up = lane_prev(q, carry);
          const float T_excl = (k == 0) ? 1.0f : (lane == 0 ? carry : up);
          carry = lane_value(q, 63);
          const float w = occ * T_excl;
          pw[k] = w; p0[k] = in[c].c0; p1[k] = in[c].c1; p2[k] = in[c].c2; p3[k] = in[c].dp;
          if (a.weights) a.weights[ray * S + k] = w;
          s0 = fmaf(w, in[c].c0, s0); s1 = fmaf(w, in[c].c1, s1); s2 = fmaf(w, in[c].c2, s2); s3 = fmaf(w, in[c].dp, s3);
          s4 += w;
        }
      }
    }
    float mv[5] = {s0, s1, s2, s3, s4};
    wave_scan_add_n<5>(mv);
    const float m0 = lane_value(mv[0], 63), m1 = lane_value(mv[1], 63), m2 = lane_value(mv[2], 63), m3 = lane_value(mv[3], 63),
                m4 = lane_value(mv[4], 63);
    WAVE_SYNC();
    float sv[4] = {0.f, 0.f, 0.f, 0.f};                            // variances around the finished means (rm.py:781-790)
    for (int base = 0; base < S; base += 64) {
      const int k = base + lane;
      const float w = pw[k];
      const float e0 = m0 - p0[k], e1 = m1 - p1[k], e2 = m2 - p2[k], e3 = m3 - p3[k];
      sv[0] = fmaf(w * e0, e0, sv[0]); sv[1] = fmaf(w * e1, e1, sv[1]); sv[2] = fmaf(w * e2, e2, sv[2]); sv[3] = fmaf(w * e3, e3, sv[3]);
    }
    wave_scan_add_n<4>(sv);
    const float v0 = lane_value(sv[0], 63), v1 = lane_value(sv[1], 63), v2 = lane_value(sv[2], 63), v3 = lane_value(sv[3], 63);
    if (lane == 0) {
      if (a.rgbd) reinterpret_cast<float4*>(a.rgbd)[ray] = make_float4(m0, m1, m2, m3);
      if (a.C) { a.C[3 * ray] = m0; a.C[3 * ray + 1] = m1; a.C[3 * ray + 2] = m2; }
      if (a.D) a.D[ray] = m3;
      if (a.Cv) { a.Cv[3 * ray] = v0; a.Cv[3 * ray + 1] = v1; a.Cv[3 * ray + 2] = v2; }
      if (a.Dv) a.Dv[ray] = v3;
      if (a.term) a.term[ray] = 1.0f - (1.0f - m4);
    }
    WAVE_SYNC();
  }
}

static int comp_grid(int64_t N, int S, int* rays_per_wave) {
  // HBM-bound: enough waves to hide memory latency (256 CUs x 24 waves; LDS allows 7 blocks of 4 waves per CU),
  // whole rays per wave
  const int64_t target_waves = 256 * 24;
  int64_t rpw = (N + target_waves - 1) / target_waves;
  if (rpw < 1) rpw = 1;
  *rays_per_wave = (int)rpw;
  const int64_t waves = (N + rpw - 1) / rpw;
  return (int)((waves + NGM_WAVES_PER_BLOCK - 1) / NGM_WAVES_PER_BLOCK);
}

int ngm_launch_composite_fwd(const CompositeArgs& a, hipStream_t st) {
  NgmProfScope prof_(NGM_K_COMPOSITE_FWD, st);
  if (a.S > CQ_MAXS || a.S < 1) return NGM_E_UNSUPPORTED;
  int rpw;
  const int blocks = comp_grid(a.N, a.S, &rpw);
  const bool keep = a.S <= CompWaveLds<true>::MAXS;
  const int pk = a.pair_field ? a.pair_K : 0;
  if (pk && (pk > 8 || !a.pair_w || !a.pair_out || !a.ray_dir || !a.dists || a.pcam || a.out4)) return NGM_E_INVALID;
  const int gm = a.rc.geometry_mode;
  if (a.S % 64 == 0 && (gm == NGM_GEO_NRGBD || gm == NGM_GEO_OCCUPANCY)) {
    // whole-ray steps: the specialised kernel (same sums, a fifth of the instructions)
    const size_t lds = (size_t)NGM_WAVES_PER_BLOCK * 5 * a.S * sizeof(float);
    const int src = pk ? pk : (a.out4 ? 9 : 0);
    static const hipError_t attr = [] {
      hipError_t e = hipSuccess;
#define NGM_CWA(SRC_) do { const hipError_t x = hipFuncSetAttribute(reinterpret_cast<const void*>(k_composite_fwd_whole<SRC_>), hipFuncAttributeMaxDynamicSharedMemorySize, NGM_WAVES_PER_BLOCK * 5 * CQ_MAXS * 4); if (x != hipSuccess) e = x; } while (0)
      NGM_CWA(0); NGM_CWA(1); NGM_CWA(2); NGM_CWA(3); NGM_CWA(4); NGM_CWA(5); NGM_CWA(6); NGM_CWA(7); NGM_CWA(8); NGM_CWA(9);
#undef NGM_CWA
      return e;
    }();
    if (attr != hipSuccess) return NGM_E_HIP;
#define NGM_CW(SRC_) hipLaunchKernelGGL((k_composite_fwd_whole<SRC_>), dim3(std::max(blocks, 1)), dim3(NGM_BLOCK), lds, st, a, rpw)
    switch (src) { case 0: NGM_CW(0); break; case 1: NGM_CW(1); break; case 2: NGM_CW(2); break; case 3: NGM_CW(3); break;
                   case 4: NGM_CW(4); break; case 5: NGM_CW(5); break; case 6: NGM_CW(6); break; case 7: NGM_CW(7); break;
                   case 8: NGM_CW(8); break; default: NGM_CW(9); break; }
#undef NGM_CW
    return 0;
  }
#define NGM_CF(PK_)                                                                                                     \
  do {                                                                                                                  \
    if (keep) hipLaunchKernelGGL((k_composite_fwd<true, PK_>), dim3(std::max(blocks, 1)), dim3(NGM_BLOCK), 0, st, a, rpw);  \
    else hipLaunchKernelGGL((k_composite_fwd<false, PK_>), dim3(std::max(blocks, 1)), dim3(NGM_BLOCK), 0, st, a, rpw);      \
  } while (0)
  if (pk == 0) NGM_CF(0); else if (pk == 1) NGM_CF(1); else if (pk == 2) NGM_CF(2); else if (pk == 3) NGM_CF(3); else if (pk == 4) NGM_CF(4);
  else if (pk == 5) NGM_CF(5); else if (pk == 6) NGM_CF(6); else if (pk == 7) NGM_CF(7); else NGM_CF(8);
#undef NGM_CF
  return 0;
}

// ================================================================================================
// quadrature backward (pointwise geometry modes): dL/dcolour_k = w_k dC ; dL/docc_k = T_k (a_k - Q_k)
// with a_k = dC.c_k + dD d_k + dterm and the suffix recursion Q_{k-1} = a_k o_k + (1-o_k) Q_k, evaluated
// as a REVERSE segmented scan of affine maps (exact also when 1-o_k = 0, unlike the division form).
// ================================================================================================
// per-wave LDS planes of CQ_MAXS floats each: exclusive transmittance (later dL/docc), occ, d occ_k / d geom_k and, in
// neus mode only, d occ_k / d geom_{k+1}, d occ_k / d isd and the per-ray d isd accumulators.  The plane count is
// chosen at launch (3 or 5): the two extra planes would otherwise halve the occupancy of this HBM-bound kernel.
struct CompBwdLds {
  float* tex; float* occ; float* doc; float* dnx; float* dis; float* risd;
  __device__ __forceinline__ CompBwdLds(float* base, bool neus) {
    tex = base; occ = base + CQ_MAXS; doc = base + 2 * CQ_MAXS;
    dnx = neus ? base + 3 * CQ_MAXS : nullptr; dis = neus ? base + 4 * CQ_MAXS : nullptr;
    risd = neus ? base + 5 * CQ_MAXS : nullptr;
  }
  static __host__ __device__ int floats(bool neus) { return neus ? 5 * CQ_MAXS + CQ_BR : 3 * CQ_MAXS; }
};

__global__ __launch_bounds__(NGM_BLOCK) void k_composite_bwd(CompositeArgs a, int rays_per_wave) {
  extern __shared__ __attribute__((aligned(16))) float cb_lds_raw[];
  const bool neus = a.rc.geometry_mode == NGM_GEO_NEUS;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  CompBwdLds wl(cb_lds_raw + wave * CompBwdLds::floats(neus), neus);
  const int64_t gw = (int64_t)blockIdx.x * NGM_WAVES_PER_BLOCK + wave;
  const int64_t r_beg = min(a.N, gw * rays_per_wave), r_end = min(a.N, r_beg + rays_per_wave);
  const int S = a.S;
  const float inv_s = 1.0f / (float)S;
  const int BR = max(1, min(CQ_BR, CQ_MAXS / S));
  for (int64_t rb = r_beg; rb < r_end; rb += BR) {
    const int nb = (int)min<int64_t>(BR, r_end - rb);
    const int nsamp = nb * S;
    // forward sweep: occupancy + exclusive transmittance
    float carry = 1.0f;
    for (int base = 0; base < nsamp; base += 64) {
      const int idx = base + lane;
      const bool valid = idx < nsamp;
      const int rl = valid ? fdiv_idx2(idx, inv_s, S) : 0;
      const int k = valid ? idx - rl * S : 0;
      float dodg0 = 0.f, dnext0 = 0.f, disd0 = 0.f;
      const float occ = valid ? occ_at(a, rb + rl, k, S, &dodg0) : 0.f;
      if (neus && valid) neus_partials(a, rb + rl, k, S, &dodg0, &dnext0, &disd0);
      float q = seg_scan_mul(1.0f - occ, k, lane);
      if (k > lane) q *= carry;
      const float up = lane_prev(q, carry);
      const float T_excl = (k == 0) ? 1.0f : (lane == 0 ? carry : up);
      carry = lane_value(q, 63);
      if (valid) {
        wl.tex[idx] = T_excl; wl.occ[idx] = occ; wl.doc[idx] = dodg0;
        if (neus) { wl.dnx[idx] = dnext0; wl.dis[idx] = disd0; }
      }
    }
    if (neus && lane < nb) wl.risd[lane] = 0.f;
    WAVE_SYNC();
    // reverse sweep
    float carryQ = 0.f;
    const int nsteps = (nsamp + 63) / 64;
    for (int st = nsteps - 1; st >= 0; --st) {
      const int idx = st * 64 + lane;
      const bool valid = idx < nsamp;
      const int rl = valid ? fdiv_idx2(idx, inv_s, S) : 0;
      const int k = valid ? idx - rl * S : 0;
      const int kr = valid ? S - 1 - k : 0;
      const int64_t ray = rb + rl, g = ray * S + k;
      float c0 = 0, c1 = 0, c2 = 0, dp = 0, T = 0, occ = 0, dodg = 0;
      float dC0 = 0, dC1 = 0, dC2 = 0, dD = 0, dT = 0;
      if (valid) {
        const Rgb3 c = load_rgb(a.colors, g); c0 = c.x; c1 = c.y; c2 = c.z; dp = a.depths[g];
        T = wl.tex[idx]; occ = wl.occ[idx]; dodg = wl.doc[idx];
        if (a.dC) { dC0 = a.dC[3 * ray]; dC1 = a.dC[3 * ray + 1]; dC2 = a.dC[3 * ray + 2]; }
        if (a.dD) dD = a.dD[ray];
        if (a.dterm) dT = a.dterm[ray];
      }
      const float ak = dC0 * c0 + dC1 * c1 + dC2 * c2 + dD * dp + dT;
      float A = valid ? ak * occ : 0.f, B = valid ? 1.0f - occ : 1.0f;
      seg_rscan_affine64(A, B, kr, lane);
      const bool extends = valid && (kr > 63 - lane);
      const float Qend = extends ? carryQ : 0.f;
      const float nA = __shfl_down(A, 1, 64), nB = __shfl_down(B, 1, 64);
      const float Qk = (kr >= 1 && lane < 63) ? fmaf(nB, Qend, nA) : Qend;
      const float Qbefore = fmaf(B, Qend, A);
      carryQ = lane_value(Qbefore, 0);
      if (valid) {
        const float w = occ * T;
        if (a.d_colors) *reinterpret_cast<Rgb3*>(a.d_colors + 3 * g) = Rgb3{w * dC0, w * dC1, w * dC2};
        if (neus) wl.tex[idx] = T * (ak - Qk);       // dL/docc_k, combined with the neighbour's below
        else if (a.d_geoms) a.d_geoms[g] = T * (ak - Qk) * dodg;
      }
    }
    WAVE_SYNC();
    if (neus) {
      // occ_k depends on g_k and g_{k+1}: d g_k = docc_k * d_self_k + docc_{k-1} * d_next_{k-1}; d isd = sum_k docc_k * d_isd_k
      for (int base = 0; base < nsamp; base += 64) {
        const int idx = base + lane;
        const bool valid = idx < nsamp;
        const int rl = valid ? fdiv_idx2(idx, inv_s, S) : 0;
        const int k = valid ? idx - rl * S : 0;
        float dg = 0.f, di = 0.f;
        if (valid) {
          const float docc = wl.tex[idx];
          dg = docc * wl.doc[idx];
          if (k > 0) dg = fmaf(wl.tex[idx - 1], wl.dnx[idx - 1], dg);
          di = docc * wl.dis[idx];
          if (a.d_geoms) a.d_geoms[(rb + rl) * S + k] = dg;
        }
        const float si = seg_scan_add(di, k, lane);
        const bool tail = valid && (k == S - 1 || lane == 63 || idx == nsamp - 1);
        if (tail) wl.risd[rl] += si;
        WAVE_SYNC();
      }
      if (lane < nb && a.d_isds) a.d_isds[rb + lane] = wl.risd[lane];
      WAVE_SYNC();
    }
  }
}


// Whole-ray steps (S = 64 NST, pointwise geometry modes, separate tensors): no segment bookkeeping, the ray's transmittances /
// occupancies / derivatives stay in registers between the two sweeps, every load of a ray is issued before its first use,
// one exp + one rcp per occupancy (occ_pointwise_fast, as the backward fused into the MLP backward).  The generic kernel above
// spends 288 vector instructions per 64 samples on a stream that is VALU-bound (profiles/r05_pmc_stages_sq.json).
__device__ __forceinline__ void wave_rscan_affine64(float& A, float& B) {
  // unsegmented reverse scan of affine maps x -> A + B x over the 64 lanes (lanes beyond a row's end read the identity)
#define NGM_WRS_STEP(d)                                                                  \
  {                                                                                       \
    const float oA = dpp_take<NGM_DPP_ROW_SHL(d), 0xf>(0.f, A), oB = dpp_take<NGM_DPP_ROW_SHL(d), 0xf>(1.f, B); \
    A = fmaf(B, oA, A); B = B * oB;                                                       \
  }
  NGM_WRS_STEP(1) NGM_WRS_STEP(2) NGM_WRS_STEP(4) NGM_WRS_STEP(8)
#undef NGM_WRS_STEP
  const int lane = (int)(threadIdx.x & 63);
#pragma unroll
  for (int row = 2; row >= 0; --row) {
    const int head = 16 * (row + 1);
    const float An = lane_value(A, head), Bn = lane_value(B, head);
    if ((lane >> 4) == row) { A = fmaf(B, An, A); B = B * Bn; }
  }
}
template <int NST>
__global__ __launch_bounds__(NGM_BLOCK) void k_composite_bwd_whole(CompositeArgs a, int rays_per_wave) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t gw = (int64_t)blockIdx.x * NGM_WAVES_PER_BLOCK + wave;
  const int64_t r_beg = min(a.N, gw * rays_per_wave), r_end = min(a.N, r_beg + rays_per_wave);
  constexpr int S = 64 * NST;
  const int mode = a.rc.geometry_mode;
  const float gamma = a.rc.geometry_factor;
  for (int64_t ray = r_beg; ray < r_end; ++ray) {
    const int64_t g0 = ray * S + lane;
    float T[NST], oc[NST], dg[NST], dp[NST];
    Rgb3 col[NST];
#pragma unroll
    for (int st = 0; st < NST; ++st) dg[st] = a.geoms[g0 + 64 * st];          // (geometry values, replaced by the derivatives below)
#pragma unroll
    for (int st = 0; st < NST; ++st) { col[st] = load_rgb(a.colors, g0 + 64 * st); dp[st] = a.depths[g0 + 64 * st]; }
    const float dC0 = a.dC ? a.dC[3 * ray] : 0.f, dC1 = a.dC ? a.dC[3 * ray + 1] : 0.f, dC2 = a.dC ? a.dC[3 * ray + 2] : 0.f;
    const float dD = a.dD ? a.dD[ray] : 0.f, dT = a.dterm ? a.dterm[ray] : 0.f;
    float carry = 1.0f;
#pragma unroll
    for (int st = 0; st < NST; ++st) {
      float d_;
      const float occ = occ_pointwise_fast(mode, gamma, dg[st], &d_);
      float q = wave_scan_mul(1.0f - occ);
      q *= carry;
      const float up = lane_prev(q, carry);
      T[st] = (st == 0 && lane == 0) ? 1.0f : (lane == 0 ? carry : up);
      carry = lane_value(q, 63);
      oc[st] = occ; dg[st] = d_;
    }
    float carryQ = 0.f;
#pragma unroll
    for (int st = NST - 1; st >= 0; --st) {
      const float ak = dC0 * col[st].x + dC1 * col[st].y + dC2 * col[st].z + dD * dp[st] + dT;
      float A = ak * oc[st], B = 1.0f - oc[st];
      wave_rscan_affine64(A, B);
      const float Qend = (st == NST - 1) ? 0.f : carryQ;
      const float nA = lane_next(A, 0.f), nB = lane_next(B, 1.0f);
      const float Qk = (lane < 63) ? fmaf(nB, Qend, nA) : Qend;
      carryQ = lane_value(fmaf(B, Qend, A), 0);
      const float w = oc[st] * T[st];
      const int64_t g = g0 + 64 * st;
      if (a.d_colors) *reinterpret_cast<Rgb3*>(a.d_colors + 3 * g) = Rgb3{w * dC0, w * dC1, w * dC2};
      if (a.d_geoms) a.d_geoms[g] = T[st] * (ak - Qk) * dg[st];
    }
  }
}

int ngm_launch_composite_bwd(const CompositeArgs& a, hipStream_t st) {
  NgmProfScope prof_(NGM_K_COMPOSITE_BWD, st);
  if (a.S > CQ_MAXS || a.S < 1) return NGM_E_UNSUPPORTED;
  int rpw;
  const int blocks = comp_grid(a.N, a.S, &rpw);
  const int gm = a.rc.geometry_mode;
  static const bool no_whole = getenv("NGM_NO_WHOLE_COMP_BWD") != nullptr;       // A/B and test knob: the generic kernel
  if (!no_whole && a.S % 64 == 0 && a.S <= 512 && (gm == NGM_GEO_NRGBD || gm == NGM_GEO_OCCUPANCY) && !comp_packed_host(a) &&
      a.colors && a.geoms && a.depths) {
#define NGM_CBW(N_) hipLaunchKernelGGL((k_composite_bwd_whole<N_>), dim3(std::max(blocks, 1)), dim3(NGM_BLOCK), 0, st, a, rpw)
    switch (a.S / 64) { case 1: NGM_CBW(1); break; case 2: NGM_CBW(2); break; case 3: NGM_CBW(3); break; case 4: NGM_CBW(4); break;
                        case 5: NGM_CBW(5); break; case 6: NGM_CBW(6); break; case 7: NGM_CBW(7); break; default: NGM_CBW(8); break; }
#undef NGM_CBW
    return 0;
  }
  const size_t lds = sizeof(float) * CompBwdLds::floats(a.rc.geometry_mode == NGM_GEO_NEUS) * NGM_WAVES_PER_BLOCK;
  (void)hipFuncSetAttribute((const void*)k_composite_bwd, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipLaunchKernelGGL(k_composite_bwd, dim3(std::max(blocks, 1)), dim3(NGM_BLOCK), lds, st, a, rpw);
  return 0;
}

// ================================================================================================
// Backward of compositing + losses on the fused forward's stash.  Reads (colour, geometry | t, T),
// overwrites stashA with dL/d(raw MLP outputs) for the MFMA backward kernel.
// ================================================================================================
__global__ __launch_bounds__(NGM_BLOCK) void k_stash_bwd(StashBwdArgs a, int rays_per_wave) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t N = (int64_t)a.F * a.R;
  const int64_t gw = (int64_t)blockIdx.x * NGM_WAVES_PER_BLOCK + wave;
  const int64_t r_beg = min(N, gw * rays_per_wave), r_end = min(N, r_beg + rays_per_wave);
  const int S = a.S;
  const float tau = a.rc.truncation_distance, gamma = a.rc.geometry_factor, cf = a.rc.color_factor;
  const int mode = a.rc.geometry_mode;
  // The streams of a step (stash, ray table, per-ray predictions / targets) are fetched ONE STEP AHEAD: the first step's
  // loads travel under the loss-partial reduction below, the next step's under this step's scans (a wave runs its
  // steps back to front with a carried suffix value, so nothing else overlaps them at one or two steps per wave).
  const int64_t nsamp_all = (r_end - r_beg) * S;
  const int64_t nsteps = (nsamp_all + 63) / 64;
  struct StepIn {
    float4 sa, r0, r1, pr, tg, vr; float2 sb; float term, tprob; unsigned char dm, tm;
  };
  // variance-weighted loss modes (losses.py:30-36, 64-75): the loss reads the rendered variances (rm.py:781-790) and its
  // gradient reaches the samples through them as well
  const bool nll = a.seed_mode == 0 ? (a.rc.photometric_mode == NGM_PHOTO_GAUSSIAN_NLL || a.rc.depth_mode != NGM_DEPTH_HUBER)
                                    : (a.d_cvars != nullptr || a.d_dvars != nullptr);     // explicit seeds on the variances
  auto fetch = [&](int64_t st, StepIn& in) __attribute__((always_inline)) {
    const int64_t idx = st * 64 + lane;
    in.sa = in.r0 = in.r1 = in.pr = in.tg = in.vr = make_float4(0.f, 0.f, 0.f, 0.f);
    in.sb = make_float2(0.f, 0.f); in.term = in.tprob = 0.f; in.dm = in.tm = 0;
    if (st >= 0 && idx < nsamp_all) {
      const int64_t rl = idx / S;
      const int64_t ray = r_beg + rl, g = ray * S + (idx - rl * S);
      in.sa = a.stashA[g];
      in.sb = a.stashB[g];
      in.r1 = reinterpret_cast<const float4*>(a.raytab)[2 * ray + 1];
      if (a.xyz_out) in.r0 = reinterpret_cast<const float4*>(a.raytab)[2 * ray];
      if (a.seed_mode == 0) {
        in.pr = reinterpret_cast<const float4*>(a.pred.rgbds)[ray];
        in.tg = reinterpret_cast<const float4*>(a.tg.rgbds)[ray];
        in.term = a.pred.term_probs[ray];
        if (nll) in.vr = make_float4(a.pred.color_vars[3 * ray], a.pred.color_vars[3 * ray + 1], a.pred.color_vars[3 * ray + 2],
                                     a.pred.depth_vars[ray]);
        in.dm = a.tg.depth_mask[ray];
        if (a.tg.term_mask) { in.tm = a.tg.term_mask[ray]; in.tprob = a.tg.term_probs[ray]; }
      } else if (nll) {            // explicit seeds on the variances: the forward's means and weight sum, the seeds themselves
        in.pr = reinterpret_cast<const float4*>(a.pred.rgbds)[ray];
        in.term = a.pred.term_probs[ray];
        in.vr = make_float4(a.d_cvars ? a.d_cvars[3 * ray] : 0.f, a.d_cvars ? a.d_cvars[3 * ray + 1] : 0.f,
                            a.d_cvars ? a.d_cvars[3 * ray + 2] : 0.f, a.d_dvars ? a.d_dvars[ray] : 0.f);
      }
    }
  };
  StepIn cur;
  fetch(nsteps - 1, cur);
  // global loss normalisers (after the caller's all-reduce, or -- deferred reduction -- summed here by every workgroup
  // from the forward's partials in k_loss_reduce's fixed order: identical in all workgroups, deterministic)
  __shared__ float s_red[16][17];
  __shared__ float s_sums[NGM_NUM_LOSS_SUMS];
  const float* sums = a.loss_sums;
  if (a.seed_mode == 0 && a.loss_partials) {             // kernel-uniform branch
    const int slot = threadIdx.x & 15, part = threadIdx.x >> 4;
    float s = 0.f;
    for (int b = part; b < a.n_partials; b += 16) s += a.loss_partials[(int64_t)b * NGM_NUM_LOSS_SUMS + slot];
    s_red[part][slot] = s;
    __syncthreads();
    if (threadIdx.x < NGM_NUM_LOSS_SUMS) {
      float t = 0.f;
#pragma unroll
      for (int p = 0; p < 16; ++p) t += s_red[p][threadIdx.x];
      s_sums[threadIdx.x] = t;
      if (blockIdx.x == 0 && a.sums_out) a.sums_out[threadIdx.x] = t;
    }
    __syncthreads();
    sums = s_sums;
    if (blockIdx.x == 0 && threadIdx.x == 0 && a.counter) *a.counter += 1ull;
  }
  float k_photo = 0, k_depth = 0, k_term = 0, k_fs = 0, k_ts = 0;
  if (a.seed_mode == 0) {
    const float n_m = sums[NGM_LS_PHOTO_CNT], n_d = sums[NGM_LS_DEPTH_CNT], n_t = sums[NGM_LS_TERM_CNT],
                n_fs = sums[NGM_LS_FS_CNT], n_ts = sums[NGM_LS_TSDF_CNT];
    k_photo = n_m > 0 ? a.rc.w_photometric / (3.0f * n_m) : 0.f;
    k_depth = n_d > 0 ? a.rc.w_depth / n_d : 0.f;
    k_term = n_t > 0 ? a.rc.w_termination * 2.0f / n_t : 0.f;
    k_fs = n_fs > 0 ? a.rc.w_freespace * 2.0f / n_fs : 0.f;
    k_ts = n_ts > 0 ? a.rc.w_tsdf * 2.0f / n_ts : 0.f;
  }
  const bool photo_l1 = a.seed_mode == 0 && (a.rc.photometric_mode == NGM_PHOTO_L1 || photo_nll_uses_l1(a.rc, sums));
  // the loss scalars ride along (one thread; saves a launch in the training step)
  if (a.loss_out && a.seed_mode == 0 && blockIdx.x == 0 && threadIdx.x == 0) loss_values_from_sums(a.rc, sums, a.loss_out);
  float carryQ = 0.f;
  for (int64_t st = nsteps - 1; st >= 0; --st) {
    const int64_t idx = st * 64 + lane;
    const bool valid = idx < nsamp_all;
    // idx / S with idx possibly large: exact integer division (one per lane per step)
    const int64_t rl = valid ? idx / S : 0;
    const int k = valid ? (int)(idx - rl * S) : 0;
    const int kr = valid ? S - 1 - k : 0;
    const int64_t ray = r_beg + rl, g = ray * S + k;
    float c0 = 0, c1 = 0, c2 = 0, geom = 0, t = 0, T = 0, dzc = 0, gt = 0;
    float dC0 = 0, dC1 = 0, dC2 = 0, dD = 0, dT = 0;
    // d loss / d variance (colour x 3, depth) and the means the variances are taken around; zero outside the nll modes
    float gV0 = 0, gV1 = 0, gV2 = 0, gVd = 0, mC0 = 0, mC1 = 0, mC2 = 0, mD = 0;
    const StepIn in = cur;
    fetch(st - 1, cur);                                  // no-op for st == 0
    if (valid) {
      const float4 sa = in.sa;
      const float2 sb = in.sb;
      c0 = sa.x; c1 = sa.y; c2 = sa.z; geom = sa.w; t = sb.x; T = sb.y;
      const float4 r1 = in.r1;
      dzc = r1.z; gt = r1.w;
      if (a.xyz_out) {             // hash encoding: the position k_hash_grad searches the simplex of (the forward's own fmaf)
        const float4 r0 = in.r0;
        typedef float v4f __attribute__((ext_vector_type(4)));
        const v4f p = {fmaf(t, r0.w, r0.x), fmaf(t, r1.x, r0.y), fmaf(t, r1.y, r0.z), 0.f};
        __builtin_nontemporal_store(p, reinterpret_cast<v4f*>(a.xyz_out + g));
      }
      if (a.seed_mode == 0) {
        const float4 pr = in.pr;
        const float4 tg = in.tg;
        const float term = in.term;
        const bool m = in.dm && (term > a.rc.term_threshold);
        if (m) {
          const float e0 = pr.x - tg.x, e1 = pr.y - tg.y, e2 = pr.z - tg.z;
          if (a.rc.photometric_mode == NGM_PHOTO_L2) {      // d mean(e^2) (losses.py:28-29)
            dC0 = 2.0f * k_photo * e0; dC1 = 2.0f * k_photo * e1; dC2 = 2.0f * k_photo * e2;
          } else if (photo_l1) {                             // d mean|e|  (losses.py:26-27; also gaussian_nll's L1 branch)
            dC0 = k_photo * ((e0 > 0.f) - (e0 < 0.f)); dC1 = k_photo * ((e1 > 0.f) - (e1 < 0.f));
            dC2 = k_photo * ((e2 > 0.f) - (e2 < 0.f));
          } else {                                           // gaussian_nll (losses.py:30-33): 0.5 e^2 / v + 0.5 log v
            const float4 v = in.vr;
            dC0 = k_photo * e0 / v.x; dC1 = k_photo * e1 / v.y; dC2 = k_photo * e2 / v.z;
            gV0 = k_photo * 0.5f * (1.0f / v.x - e0 * e0 / (v.x * v.x));
            gV1 = k_photo * 0.5f * (1.0f / v.y - e1 * e1 / (v.y * v.y));
            gV2 = k_photo * 0.5f * (1.0f / v.z - e2 * e2 / (v.z * v.z));
          }
          const float e = pr.w - tg.w, dl = a.rc.huber_delta;
          if (a.rc.depth_mode == NGM_DEPTH_GAUSSIAN_NLL) {   // losses.py:64-69
            const float v = in.vr.w + 1e-15f;
            dD = k_depth * e / v;
            gVd = k_depth * 0.5f * (1.0f / v - e * e / (v * v));
          } else if (a.rc.depth_mode == NGM_DEPTH_LAPLACIAN_NLL) {   // losses.py:70-75: |e| / sqrt(0.5 v + 1e-6) + 0.5 log(2 v + 1e-6)
            const float sv = 0.5f * in.vr.w + 1e-6f, rs = 1.0f / sqrtf(sv);
            dD = k_depth * ((e > 0.f) - (e < 0.f)) * rs;
            gVd = k_depth * (1.0f / (2.0f * in.vr.w + 1e-6f) - 0.25f * fabsf(e) * rs / sv);
          } else
            dD = k_depth * ((fabsf(e) < dl) ? e : dl * ((e > 0.f) - (e < 0.f)));
          if (nll) {
            // variances are taken around the finished means (rm.py:781-790): V = sum_k w_k (c_k - C)^2, so besides
            // d V / d w_k = (c_k - C)^2 and d V / d c_k = 2 w_k (c_k - C) there is d V / d C = -2 C (1 - W), W = sum w = the
            // termination probability (zero only where the weights sum to one); it joins the mean's own seed
            mC0 = pr.x; mC1 = pr.y; mC2 = pr.z; mD = pr.w;
            const float bg2 = 2.0f * (1.0f - term);
            dC0 -= gV0 * bg2 * mC0; dC1 -= gV1 * bg2 * mC1; dC2 -= gV2 * bg2 * mC2; dD -= gVd * bg2 * mD;
          }
        }
        if (a.tg.term_mask && in.tm) dT = k_term * (term - in.tprob);
      } else {
        const float4 d = reinterpret_cast<const float4*>(a.d_rgbds)[ray];
        dC0 = d.x; dC1 = d.y; dC2 = d.z; dD = d.w;
        if (a.d_term) dT = a.d_term[ray];
        if (nll) {                 // as above: d V / d C = -2 C (1 - W) joins the mean's seed
          gV0 = in.vr.x; gV1 = in.vr.y; gV2 = in.vr.z; gVd = in.vr.w;
          mC0 = in.pr.x; mC1 = in.pr.y; mC2 = in.pr.z; mD = in.pr.w;
          const float bg2 = 2.0f * (1.0f - in.term);
          dC0 -= gV0 * bg2 * mC0; dC1 -= gV1 * bg2 * mC1; dC2 -= gV2 * bg2 * mC2; dD -= gVd * bg2 * mD;
        }
      }
    }
    const float depth = -(dzc * t);
    float dodg = 0.f, occ = 0.f;
    // neus (rm.py:753-758): tno = sigmoid(isd gamma g), occ_k = max((tno_k - tno_{k+1}) / (tno_k + eps), 0), k < S - 1
    float tno = 0.f, tnx = 0.f, isd = 0.f;
    if (valid) {
      if (mode == NGM_GEO_DENSITY) { if (k < S - 1) occ = occ_density(geom, a.stashB[g + 1].x - t, &dodg); }   // last sample dropped
      else if (mode == NGM_GEO_NEUS) {
        const int64_t fld = ray / a.R;
        const int64_t row = a.field_index ? a.field_index[fld] : fld;
        isd = 1.0f / fabsf(a.neus_sd[row * a.neus_sd_stride]);
        tno = ngm_sigmoid(isd * gamma * geom);
        if (k < S - 1) {
          tnx = ngm_sigmoid(isd * gamma * a.stashA[g + 1].w);
          occ = fmaxf((tno - tnx) / (tno + 1e-5f), 0.f);
        }
      } else occ = occ_pointwise(mode, gamma, geom, &dodg);
    }
    // d loss / d w_k: through the means, the termination probability and (nll modes) the variances
    auto a_of = [&](float q0, float q1, float q2, float qd) __attribute__((always_inline)) {
      float v = dC0 * q0 + dC1 * q1 + dC2 * q2 + dD * qd + dT;
      if (nll) v += gV0 * (q0 - mC0) * (q0 - mC0) + gV1 * (q1 - mC1) * (q1 - mC1) + gV2 * (q2 - mC2) * (q2 - mC2) +
                    gVd * (qd - mD) * (qd - mD);
      return v;
    };
    const float ak = a_of(c0, c1, c2, depth);
    float A = valid ? ak * occ : 0.f, B = valid ? 1.0f - occ : 1.0f;
    seg_rscan_affine64(A, B, kr, lane);
    const bool extends = valid && (kr > 63 - lane);
    const float Qend = extends ? carryQ : 0.f;
    const float nA = __shfl_down(A, 1, 64), nB = __shfl_down(B, 1, 64);
    const float Qk = (kr >= 1 && lane < 63) ? fmaf(nB, Qend, nA) : Qend;
    const float Qbefore = fmaf(B, Qend, A);
    carryQ = lane_value(Qbefore, 0);
    float di_lane = 0.f;
    if (valid) {
      const float w = occ * T;
      float dg = T * (ak - Qk) * dodg;
      if (mode == NGM_GEO_NEUS) {
        // d loss / d tno_k = D_k docc_k/dtno_k + D_{k-1} docc_{k-1}/dtno_k with D_j = T_j (a_j - Q_j) = d loss / d occ_j.
        // D_{k-1} is rebuilt locally from the neighbour's saved values: Q_{k-1} = a_k occ_k + (1 - occ_k) Q_k is this
        // lane's own suffix value (the recursion of the reverse scan), T_{k-1}, colours, t of sample k - 1 come from the stash.
        float dtno = 0.f;
        if (k < S - 1 && tno > tnx) dtno = T * (ak - Qk) * (tnx + 1e-5f) / ((tno + 1e-5f) * (tno + 1e-5f));
        if (k > 0) {
          const float4 pa = a.stashA[g - 1];
          const float2 pb = a.stashB[g - 1];
          const float tnp = ngm_sigmoid(isd * gamma * pa.w);
          if (tnp > tno) {                                          // occ_{k-1} > 0: the clamp is inactive
            const float a_prev = a_of(pa.x, pa.y, pa.z, -(dzc * pb.x));
            const float Q_prev = fmaf(1.0f - occ, Qk, ak * occ);
            dtno -= pb.y * (a_prev - Q_prev) / (tnp + 1e-5f);
          }
        }
        const float ds = tno * (1.0f - tno);
        dg = dtno * isd * gamma * ds;
        di_lane = dtno * gamma * geom * ds;                        // d loss / d isd of this sample (also through the constant
      }                                                            // geometry of samples behind the camera, as in the reference)
      if (a.seed_mode == 0) {
        const float thr = (gt - tau) * (gt != 0.0f ? 1.0f : 0.0f);
        if (t < thr) dg += k_fs * (geom * tau - tau) * tau;
        const float dl = gt - t;
        if (fabsf(dl) < tau && gt != 0.0f) dg += k_ts * (geom * tau - dl) * tau;
      } else if (a.d_geom_samples) {
        dg += a.d_geom_samples[g];
      }
      if (a.rc.overwrite_behind_camera && dzc * t > 0.f) dg = 0.f;   // overwritten sample: no gradient reaches the MLP output
      {
        typedef float v4f __attribute__((ext_vector_type(4)));
        // d loss / d colour_k = w_k (d C + 2 d V (c_k - C)): the second term only in the nll modes (gV = 0 otherwise)
        const v4f dv = {cf * w * (dC0 + 2.0f * gV0 * (c0 - mC0)), cf * w * (dC1 + 2.0f * gV1 * (c1 - mC1)),
                        cf * w * (dC2 + 2.0f * gV2 * (c2 - mC2)), dg};
        // in place over the saved forward values, except in neus mode (neighbouring lanes / waves still read them)
        __builtin_nontemporal_store(dv, reinterpret_cast<v4f*>((a.d_out ? a.d_out : a.stashA) + g));   // read once, by the MLP backward
      }
    }
    if (mode == NGM_GEO_NEUS && a.d_isd_rays) {      // kernel-uniform: per-ray sums of d loss / d isd
      const float si = seg_scan_add(di_lane, k, lane);
      const bool tail = valid && (k == S - 1 || lane == 63 || idx == nsamp_all - 1);
      // a ray split over two steps adds twice, from the same wave in program order: deterministic
      if (tail) atomicAdd(a.d_isd_rays + ray, si);
    }
  }
}

int ngm_launch_stash_bwd(const StashBwdArgs& a, hipStream_t st) {
  NgmProfScope prof_(NGM_K_STASH_BWD, st);
  if (a.rc.geometry_mode == NGM_GEO_NEUS) {
    if (!a.neus_sd || !a.d_out) return NGM_E_UNSUPPORTED;             // per-field _neus_sd and a separate gradient buffer
    if (a.d_isd_rays) (void)hipMemsetAsync(a.d_isd_rays, 0, sizeof(float) * (size_t)a.F * a.R, st);
  }
  int rpw;
  const int blocks = comp_grid((int64_t)a.F * a.R, a.S, &rpw);   // one ray per wave up to 6144 rays (2 and 4 rays per wave: +3 / +9 us at M1)
  hipLaunchKernelGGL(k_stash_bwd, dim3(std::max(blocks, 1)), dim3(NGM_BLOCK), 0, st, a, rpw);
  return 0;
}

// ================================================================================================
// loss scalars from the global sums (rm.py:1803-1871); out[0..5] = combined, termination,
// photometric, depth, freespace, tsdf.  Empty selections: NaN, as the reference (loss_values_from_sums).
// ================================================================================================
__global__ void k_loss_values(ngm_render_cfg rc, const float* sums, float* out) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  loss_values_from_sums(rc, sums, out);
}
int ngm_launch_loss_values(const ngm_render_cfg* rc, const float* sums, float* out, hipStream_t st) {
  hipLaunchKernelGGL(k_loss_values, dim3(1), dim3(64), 0, st, *rc, sums, out);
  return 0;
}

// sum of per-workgroup loss partials: 16 slots x 16 strided lanes, then a fixed-order tree
// counter (optional): the iteration counter behind the Philox offset / Adam step, advanced here -- the forward that
// read it has finished (stream order) and the Adam kernel that reads it next has not started
__global__ void k_loss_reduce(const float* partials, int nblocks, float* sums, unsigned long long* counter) {
  __shared__ float red[16][17];
  const int slot = threadIdx.x & 15, part = threadIdx.x >> 4;   // 256 threads
  float s = 0.f;
  for (int b = part; b < nblocks; b += 16) s += partials[(int64_t)b * NGM_NUM_LOSS_SUMS + slot];
  red[part][slot] = s;
  __syncthreads();
  if (threadIdx.x < NGM_NUM_LOSS_SUMS) {
    float t = 0.f;
#pragma unroll
    for (int p = 0; p < 16; ++p) t += red[p][threadIdx.x];
    sums[threadIdx.x] = t;
  }
  if (counter && threadIdx.x == 0) *counter += 1ull;
}
int ngm_launch_loss_reduce(const float* partials, int nblocks, float* sums, uint64_t* counter, hipStream_t st) {
  NgmProfScope prof_(NGM_K_LOSS_REDUCE, st);
  hipLaunchKernelGGL(k_loss_reduce, dim3(1), dim3(256), 0, st, partials, nblocks, sums, reinterpret_cast<unsigned long long*>(counter));
  return 0;
}

// neus: d loss / d _neus_sd[f] = (sum over the field's rays of d loss / d isd) * d(1 / |sd|) / d sd (rm.py:641-644);
// one workgroup per field, fixed-order tree -> deterministic
__global__ void k_neus_sd_grad(const float* d_isd_rays, int R, const float* neus_sd, int64_t sd_stride,
                               const int64_t* field_index, float* d_sd) {
  __shared__ float red[256];
  const int f = blockIdx.x;
  float s = 0.f;
  for (int r = threadIdx.x; r < R; r += 256) s += d_isd_rays[(int64_t)f * R + r];
  red[threadIdx.x] = s;
  __syncthreads();
  for (int w = 128; w >= 1; w >>= 1) {
    if ((int)threadIdx.x < w) red[threadIdx.x] += red[threadIdx.x + w];
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    const float sd = neus_sd[(field_index ? field_index[f] : f) * sd_stride];
    d_sd[f] = red[0] * (-(sd > 0.f ? 1.0f : -1.0f) / (sd * sd));
  }
}
int ngm_launch_neus_sd_grad(const float* d_isd_rays, int F, int R, const float* neus_sd, int64_t sd_stride,
                            const int64_t* field_index, float* d_sd, hipStream_t st) {
  hipLaunchKernelGGL(k_neus_sd_grad, dim3(F), dim3(256), 0, st, d_isd_rays, R, neus_sd, sd_stride, field_index, d_sd);
  return 0;
}

// copy geometry / distance planes out of the stash (Prediction.freespace_geometry / tsdf_residuals)
__global__ void k_read_stash(const float4* sa, const float2* sb, int64_t n, float* geoms, float* dists) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    if (geoms) geoms[i] = sa[i].w;
    if (dists) dists[i] = sb[i].x;
  }
}
int ngm_launch_read_stash(const float4* sa, const float2* sb, int64_t n, float* geoms, float* dists, hipStream_t st) {
  const int blocks = (int)std::min<int64_t>((n + 255) / 256, 2048);
  hipLaunchKernelGGL(k_read_stash, dim3(std::max(blocks, 1)), dim3(256), 0, st, sa, sb, n, geoms, dists);
  return 0;
}

// ================================================================================================
// sparse per-field Adam (torch.optim.Adam, L2-coupled weight decay, shared step; rm.py:357-362)
// ================================================================================================
__global__ void k_adam_sparse(float* param, float* m, float* v, int64_t stride, const float* grad, int64_t gstride,
                              const int64_t* field_index, int F, int64_t numel, float lr_bc1, float inv_sqrt_bc2,
                              float beta1, float beta2, float eps, float wd) {
  const int f = blockIdx.y;
  const int64_t row = field_index ? field_index[f] : f;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < numel; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t o = row * stride + i;
    const float p = param[o];
    const float g = grad[(int64_t)f * gstride + i] + wd * p;
    const float mn = beta1 * m[o] + (1.0f - beta1) * g;
    const float vn = beta2 * v[o] + (1.0f - beta2) * g * g;
    m[o] = mn; v[o] = vn;
    const float denom = sqrtf(vn) * inv_sqrt_bc2 + eps;
    param[o] = p - lr_bc1 * (mn / denom);
  }
}
int ngm_launch_adam(float* param, float* m, float* v, int64_t stride, const float* grad, int64_t gstride,
                    const int64_t* field_index, int F, int64_t numel, int64_t step, float lr, float beta1, float beta2,
                    float eps, float wd, hipStream_t st) {
  NgmProfScope prof_(NGM_K_ADAM, st);
  const double bc1 = 1.0 - pow((double)beta1, (double)step), bc2 = 1.0 - pow((double)beta2, (double)step);
  const float lr_bc1 = (float)((double)lr / bc1), inv_sqrt_bc2 = (float)(1.0 / sqrt(bc2));
  dim3 grid((unsigned)std::max<int64_t>(1, std::min<int64_t>((numel + 255) / 256, 64)), (unsigned)F);
  hipLaunchKernelGGL(k_adam_sparse, grid, dim3(256), 0, st, param, m, v, stride, grad, gstride, field_index, F, numel,
                     lr_bc1, inv_sqrt_bc2, beta1, beta2, eps, wd);
  return 0;
}

// all tensors of the field set in one launch: blockIdx.y = field, blockIdx.z = tensor
struct AdamMultiK {
  ngm_adam_tensor t[2 * (NGM_MAX_LAYERS + 1) + 2];
  int n;
  const int64_t* field_index;
  const int64_t* step_dev;
  int64_t step;
  float lr, beta1, beta2, eps, wd;
  int64_t* advance_step;      // non-NULL: ++*advance_step once every block has read it
  uint64_t* advance_offset;   // non-NULL: ++*advance_offset likewise (Philox offset of the next iteration)
};
__device__ unsigned int g_adam_blocks_done = 0;
__global__ void k_adam_multi(AdamMultiK a) {
  const ngm_adam_tensor& t = a.t[blockIdx.z];
  const int f = blockIdx.y;
  const int64_t row = a.field_index ? a.field_index[f] : f;
  const double step = (double)(a.step_dev ? *a.step_dev : a.step);
  float lr_bc1 = 0.f, inv_sqrt_bc2 = 0.f;
  if ((int64_t)blockIdx.x * blockDim.x < t.numel) {     // the grid is sized for the largest tensor
    lr_bc1 = (float)((double)a.lr / (1.0 - pow((double)a.beta1, step)));
    inv_sqrt_bc2 = (float)(1.0 / sqrt(1.0 - pow((double)a.beta2, step)));
  }
  const bool vec = ((t.numel | t.stride | t.grad_stride) & 3) == 0 &&
                   ((reinterpret_cast<uintptr_t>(t.param) | reinterpret_cast<uintptr_t>(t.grad) |
                     reinterpret_cast<uintptr_t>(t.exp_avg) | reinterpret_cast<uintptr_t>(t.exp_avg_sq)) & 15) == 0;
  if (vec) {                                   // 16-byte accesses (the hash tables: 128 Ki floats per field)
    const int64_t n4 = t.numel >> 2;
    float4* P4 = reinterpret_cast<float4*>(t.param + row * t.stride);
    float4* M4 = reinterpret_cast<float4*>(t.exp_avg + row * t.stride);
    float4* V4 = reinterpret_cast<float4*>(t.exp_avg_sq + row * t.stride);
    const float4* G4 = reinterpret_cast<const float4*>(t.grad + (int64_t)f * t.grad_stride);
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
      float4 p = P4[i], m = M4[i], v = V4[i];
      const float4 g4 = G4[i];
      float* pp = &p.x; float* pm = &m.x; float* pv = &v.x; const float* pg = &g4.x;
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const float g = pg[c] + a.wd * pp[c];
        const float mn = a.beta1 * pm[c] + (1.0f - a.beta1) * g;
        const float vn = a.beta2 * pv[c] + (1.0f - a.beta2) * g * g;
        pm[c] = mn; pv[c] = vn;
        pp[c] = pp[c] - lr_bc1 * (mn / (sqrtf(vn) * inv_sqrt_bc2 + a.eps));
      }
      M4[i] = m; V4[i] = v; P4[i] = p;
      if (t.param_lp) {
#pragma unroll
        for (int c = 0; c < 4; ++c) ngm_stp(t.param_lp, row * t.stride + 4 * i + c, pp[c], t.lp_dtype);
      }
    }
  } else
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < t.numel; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t o = row * t.stride + i;
    const float p = t.param[o];
    const float g = t.grad[(int64_t)f * t.grad_stride + i] + a.wd * p;
    const float mn = a.beta1 * t.exp_avg[o] + (1.0f - a.beta1) * g;
    const float vn = a.beta2 * t.exp_avg_sq[o] + (1.0f - a.beta2) * g * g;
    t.exp_avg[o] = mn; t.exp_avg_sq[o] = vn;
    const float pn = p - lr_bc1 * (mn / (sqrtf(vn) * inv_sqrt_bc2 + a.eps));
    t.param[o] = pn;
    if (t.param_lp) ngm_stp(t.param_lp, o, pn, t.lp_dtype);
  }
  // end-of-iteration bookkeeping by the last block to finish (all blocks have read the step by then)
  if (a.advance_step || a.advance_offset) {
    __syncthreads();
    if (threadIdx.x == 0) {
      __threadfence();
      const unsigned int total = gridDim.x * gridDim.y * gridDim.z;
      if (atomicAdd(&g_adam_blocks_done, 1u) == total - 1) {
        if (a.advance_step) *a.advance_step += 1;
        if (a.advance_offset) *a.advance_offset += 1;
        g_adam_blocks_done = 0;
        __threadfence();
      }
    }
  }
}
int ngm_launch_adam_multi(const ngm_adam_tensor* tensors, int n, const int64_t* field_index, int F, int64_t step,
                          const int64_t* step_dev, float lr, float beta1, float beta2, float eps, float wd,
                          int64_t* advance_step, uint64_t* advance_offset, hipStream_t st) {
  NgmProfScope prof_(NGM_K_ADAM, st);
  AdamMultiK a;
  int64_t mx = 1;
  for (int i = 0; i < n; ++i) { a.t[i] = tensors[i]; mx = std::max<int64_t>(mx, tensors[i].numel); }
  a.n = n; a.field_index = field_index; a.step_dev = step_dev; a.step = step;
  a.lr = lr; a.beta1 = beta1; a.beta2 = beta2; a.eps = eps; a.wd = wd;
  a.advance_step = advance_step; a.advance_offset = advance_offset;
  // blocks per (field, tensor): enough to put the large tensors (hash tables) on every CU
  const int64_t want = std::max<int64_t>(1, (4 * 256 + (int64_t)F - 1) / F);
  dim3 grid((unsigned)std::min<int64_t>((mx / 4 + 255) / 256 + 1, std::max<int64_t>(16, want)), (unsigned)F, (unsigned)n);
  hipLaunchKernelGGL(k_adam_multi, grid, dim3(256), 0, st, a);
  return 0;
}
__global__ void k_step_advance(int64_t* step_dev, uint64_t* off_dev) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    if (step_dev) *step_dev += 1;
    if (off_dev) *off_dev += 1;
  }
}
int ngm_launch_step_advance(int64_t* step_dev, uint64_t* off_dev, hipStream_t st) {
  hipLaunchKernelGGL(k_step_advance, dim3(1), dim3(64), 0, st, step_dev, off_dev);
  return 0;
}
