// Eval path: NeuralFieldSet.forward(use_vmap=False) (models.py:347-405) on gfx950.
//
// The reference finds the K nearest field centres of every query point, evaluates each touched field
// in a Python loop over boolean masks and blends with softmax(-distance_factor * dist).  Here:
//   k_knn_grid    : field centres binned into a uniform grid on the device (cell = the cover-grid spacing 2 r / sqrt 3 of
//                   rm.py:299, never smaller than the mask radius; one workgroup: bounding box, counts, scan, fill), then per
//                   cell of the grid extended by one ring the list of centres a point of that cell can be inside of
//   k_knn_assign  : exact K-nearest (K <= 8) per point: the cell's list decides the inside test (most samples of an image
//                   fall on an empty list and stop there) and seeds the neighbour list; where fewer than K centres are within
//                   the radius the block of cells around the point grows ring by ring until the K-th neighbour found is
//                   provably the K-th nearest; softmax weights; per-workgroup LDS histogram -> one global atomic per field
//                   per workgroup.  No limit on the number of fields.
//   k_knn_offsets : exclusive scans over fields (segment offsets, tile offsets)
//   k_knn_scatter : (point,k) pairs bucketed by field (counting sort)
//   k_knn_eval    : one workgroup per 2048-pair tile of ONE field: that field's weights resident in
//                   LDS, MLP on the matrix cores (same eval_64 as the train kernels)
//   k_knn_blend   : out[p] = sum_k w_k out_{p,k}, or outside_value on all channels
// ngm_render_eval_knn (render_image's block loop, rm.py:402-437) runs the same stages per block of rays with the points
// GENERATED inside the assignment (k_knn_raytab + the sampler's arithmetic; only the sample distances are kept) and the blend
// done inside the quadrature (k_composite_fwd* on the pair records): no point, blended output or camera-frame point in memory.
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

#ifndef KNN_TILE
#define KNN_TILE 512
#endif
#define KNN_MAXK 8
// copies of the per-field pair counters: 4096 workgroups flushing their histograms into ~100 addresses serialise on the
// device-scope atomics; sixteen copies, summed by k_knn_offsets
#ifndef KNN_COUNT_REPL
#define KNN_COUNT_REPL 16
#endif
#ifndef KNN_ASSIGN_MAX_WG
#define KNN_ASSIGN_MAX_WG 4096
#endif
// the candidate lists are kept per SUB-cell (edge c / KNN_SUB) of the grid extended by one ring: a list holds the centres within
// the mask radius of its box, and a box of half the edge sees half as many (13 instead of 26 on a cover-grid map) -- the
// assignment walks its list for every point near a field
#define KNN_SUB 2
#define KNN_NEAR_PER_FIELD (27 * KNN_SUB * KNN_SUB * KNN_SUB)   // a centre is listed by sub-cells of its own and the 26 adjacent cells only
#ifndef KNN_EVAL_B3_THREADS
#define KNN_EVAL_B3_THREADS 512
#endif
#define CQ_MAXS_EVAL 1024      // samples per ray the quadrature kernel takes (CQ_MAXS of ngm_composite.hip)

struct KnnArgs {
  ngm_field_cfg fc;
  ngm_params pr;
  int NF, K;
  int64_t P;
  const float* points; const float* pos; const float* quat;
  float distance_factor, outside_value, radius;
  float* out;
  // workspace
  int* pair_field;      // (P*K)  field of the pair, -1 if the point is outside every field
  float* pair_w;        // (P*K)
  int* counts;          // (KNN_COUNT_REPL, NF) pairs per field, replicated: workgroup b adds into copy b % KNN_COUNT_REPL
  int* cursor;          // (NF)
  int* seg_off;         // (NF+1)
  int* tile_off;        // (NF+1)
  int* sorted;          // (P*K)  pair ids grouped by field
  float4* pair_out;     // (P*K)
  // uniform grid over the field centres (built per call by k_knn_grid)
  struct KnnGridHdr* grid;   // origin, cell size, dimensions (device)
  int* cell_start;      // (max_cells + 1) exclusive prefix of the per-cell counts
  int* cell_fill;       // (max_cells) fill cursors
  float4* cell_c;       // (NF) centres grouped by cell: (x, y, z, field index as int bits)
  // per cell of the grid extended by one ring: the centres a point of that cell can be within the mask radius of (box
  // distance < radius; at most 27 cells per centre) -- the inside test and, where the map is dense, the whole search
  int* near_start;      // (KNN_SUB^3 * 4 * max_cells + 1) exclusive prefix over the sub-cells of the extended grid
  float4* near_c;       // (KNN_NEAR_PER_FIELD * NF)
  int max_cells;
  float cell_size;      // requested cell edge (>= mask radius); the kernel may coarsen it to fit max_cells
  int hist_in_lds;      // per-workgroup field histograms fit the LDS (else: global atomics per pair)
  int near_inline;      // k_knn_grid builds the candidate lists itself (small maps); else k_knn_near_* do
  // point source: `points` (P,3), or -- gen != 0, ngm_render_eval_knn -- point p is sample p % S of ray p / S of one block of
  // eval-style rays (single stratum): k_knn_raytab leaves the ray-level quantities in `ray_dir` / `ray_tab`, the assignment draws
  // the sample's distance and forms the point exactly as k_sample_rays would, leaves the distance in `dist`; the evaluation
  // re-forms the point from those
  int gen, S;
  ngm_render_cfg rc;
  ngm_rays rays;
  float* dist;          // (P)
  float* ray_dir;       // (P / S, 3)
  float4* ray_tab;      // (P / S) near, far - near, (far - near) / S of the ray's stratum
};
struct KnnGridHdr { float x0, y0, z0, c, inv_c; int nx, ny, nz, ncells, pad; };

// wave-wide min / max (uniform result): DPP row scans + row broadcasts, lane 63 read back (no LDS-pipe shuffles)
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_src(float ident, float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(ident), __float_as_int(v), CTRL, ROW_MASK, 0xf, false));
}
__device__ __forceinline__ float wave_min_f(float v) {
  const float id = INFINITY;
  v = fminf(v, dpp_src<0x111, 0xf>(id, v)); v = fminf(v, dpp_src<0x112, 0xf>(id, v));
  v = fminf(v, dpp_src<0x114, 0xf>(id, v)); v = fminf(v, dpp_src<0x118, 0xf>(id, v));
  v = fminf(v, dpp_src<0x142, 0xa>(id, v)); v = fminf(v, dpp_src<0x143, 0xc>(id, v));
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}
__device__ __forceinline__ float wave_max_f(float v) { return -wave_min_f(-v); }

// ---- uniform grid over the field centres -------------------------------------------------------------------------------
// One workgroup of 1024 threads (the map is a few hundred to a few ten thousand centres): bounding box -> cell size / grid
// dimensions (coarsened until the grid fits max_cells) -> per-cell counts -> exclusive scan -> centres grouped by cell.
// The order of the centres inside a cell is whatever the atomics give: k_knn_assign breaks distance ties by field index,
// so its result does not depend on it.
// exclusive scan of arr[0 .. n) in place by one workgroup of 1024 threads, arr[n] = total: contiguous segments per thread,
// block scan of the segment sums (sscan: 1024 ints of LDS)
__device__ __forceinline__ void block_exclusive_scan_1024(int* arr, int n, int* sscan) {
  const int t = threadIdx.x;
  const int per = (n + 1023) / 1024, b0 = min(n, t * per), b1 = min(n, b0 + per);
  int sum = 0;
  for (int i = b0; i < b1; ++i) sum += arr[i];
  sscan[t] = sum;
  __syncthreads();
  for (int d = 1; d < 1024; d <<= 1) {
    const int v = (t >= d) ? sscan[t - d] : 0;
    __syncthreads();
    sscan[t] += v;
    __syncthreads();
  }
  int run = sscan[t] - sum;
  for (int i = b0; i < b1; ++i) { const int cnt = arr[i]; arr[i] = run; run += cnt; }
  if (t == 1023) arr[n] = sscan[1023];
  __threadfence();
  __syncthreads();
}

// Candidate list of sub-cell e of the extended grid (the grid plus one ring): the centres closer to its box than the mask radius (a
// little more: the points are binned in fp32).  Four out of five samples of an image are nowhere near a field and fall
// on an empty list; for the others the list replaces the walk over the 27 cells.  fill = false: count only.
__device__ __forceinline__ int knn_near_cells(const KnnGridHdr& h) { return (h.nx + 2) * (h.ny + 2) * (h.nz + 2) * KNN_SUB * KNN_SUB * KNN_SUB; }
__device__ __forceinline__ int knn_near_visit(const KnnArgs& a, const KnnGridHdr& h, int e, bool fill) {
  const int ex = (h.nx + 2) * KNN_SUB, ey = (h.ny + 2) * KNN_SUB;
  const float reach = a.radius * 1.001f + 1e-3f * h.c, reach2 = reach * reach;
  // sub-cell (sx, sy, sz) in units of c / KNN_SUB from the grid origin, -KNN_SUB .. (n + 1) KNN_SUB - 1; its coarse cell
  const int sx = e % ex - KNN_SUB, sy = (e / ex) % ey - KNN_SUB, sz = e / (ex * ey) - KNN_SUB;
  const int cx = (sx + KNN_SUB) / KNN_SUB - 1, cy = (sy + KNN_SUB) / KNN_SUB - 1, cz = (sz + KNN_SUB) / KNN_SUB - 1;
  const float w = h.c * (1.0f / KNN_SUB);
  const float bx0 = h.x0 + (float)sx * w, by0 = h.y0 + (float)sy * w, bz0 = h.z0 + (float)sz * w;
  int cnt = 0;
  const int base = fill ? a.near_start[e] : 0;
  for (int nz = max(cz - 1, 0); nz <= min(cz + 1, h.nz - 1); ++nz)
    for (int ny = max(cy - 1, 0); ny <= min(cy + 1, h.ny - 1); ++ny) {
      const int xs = max(cx - 1, 0), xe = min(cx + 1, h.nx - 1);
      if (xs > xe) continue;
      const int rowb = (nz * h.ny + ny) * h.nx;
      for (int j = a.cell_start[rowb + xs]; j < a.cell_start[rowb + xe + 1]; ++j) {
        const float4 c = a.cell_c[j];
        const float dx = fmaxf(fmaxf(bx0 - c.x, c.x - (bx0 + w)), 0.f), dy = fmaxf(fmaxf(by0 - c.y, c.y - (by0 + w)), 0.f),
                    dz = fmaxf(fmaxf(bz0 - c.z, c.z - (bz0 + w)), 0.f);
        if (dx * dx + dy * dy + dz * dz < reach2) { if (fill) a.near_c[base + cnt] = c; ++cnt; }
      }
    }
  return cnt;
}

// a.near_inline: the candidate lists are built by this workgroup as well (small maps: one launch); otherwise by the three
// kernels below on the whole device (a map of 40 000 centres: 2 ms -> a fraction of that)
__global__ __launch_bounds__(1024) void k_knn_grid(KnnArgs a) {
  __shared__ float red[6][16];
  __shared__ int sscan[1024];
  __shared__ KnnGridHdr h;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  float lo[3] = {3.0e38f, 3.0e38f, 3.0e38f}, hi[3] = {-3.0e38f, -3.0e38f, -3.0e38f};
  for (int f = t; f < a.NF; f += 1024)
#pragma unroll
    for (int d = 0; d < 3; ++d) { const float v = a.pos[3 * f + d]; lo[d] = fminf(lo[d], v); hi[d] = fmaxf(hi[d], v); }
#pragma unroll
  for (int d = 0; d < 3; ++d) {
    const float l = wave_min_f(lo[d]), u = wave_max_f(hi[d]);
    if (lane == 0) { red[d][wave] = l; red[3 + d][wave] = u; }
  }
  __syncthreads();
  if (t == 0) {
    float L[3], U[3];
    for (int d = 0; d < 3; ++d) {
      L[d] = red[d][0]; U[d] = red[3 + d][0];
      for (int w = 1; w < 16; ++w) { L[d] = fminf(L[d], red[d][w]); U[d] = fmaxf(U[d], red[3 + d][w]); }
      if (!(L[d] <= U[d])) { L[d] = 0.f; U[d] = 0.f; }                // no (finite) centre
    }
    float c = a.cell_size;
    int n[3];
    for (int it = 0; it < 200; ++it) {
      double cells = 1.0, ext = 1.0;
      for (int d = 0; d < 3; ++d) {
        const float e = (U[d] - L[d]) / c;
        n[d] = (e < 1.0e6f) ? (int)e + 1 : 1000001;
        cells *= (double)n[d]; ext *= (double)(n[d] + 2);
      }
      if (cells <= (double)a.max_cells && ext <= 4.0 * (double)a.max_cells) break;
      c *= 1.26f;                                                      // cells / 2 per step
    }
    h.x0 = L[0]; h.y0 = L[1]; h.z0 = L[2]; h.c = c; h.inv_c = 1.0f / c;
    h.nx = n[0]; h.ny = n[1]; h.nz = n[2]; h.ncells = n[0] * n[1] * n[2]; h.pad = 0;
    *a.grid = h;
  }
  __syncthreads();
  const int nc = h.ncells;
  for (int i = t; i <= nc; i += 1024) a.cell_start[i] = 0;
  for (int i = t; i < nc; i += 1024) a.cell_fill[i] = 0;
  for (int i = t; i < KNN_COUNT_REPL * a.NF; i += 1024) a.counts[i] = 0;     // pairs per field: zero for the first assignment (k_knn_offsets re-zeroes)
  __threadfence();            // the counts are changed by L2 atomics below: no stale L1 line may serve the reads after them
  __syncthreads();
  auto cell_of = [&](int f) {
    const int ix = min(max((int)((a.pos[3 * f] - h.x0) * h.inv_c), 0), h.nx - 1);
    const int iy = min(max((int)((a.pos[3 * f + 1] - h.y0) * h.inv_c), 0), h.ny - 1);
    const int iz = min(max((int)((a.pos[3 * f + 2] - h.z0) * h.inv_c), 0), h.nz - 1);
    return (iz * h.ny + iy) * h.nx + ix;
  };
  for (int f = t; f < a.NF; f += 1024) atomicAdd(&a.cell_start[cell_of(f)], 1);
  __threadfence();
  __syncthreads();
  block_exclusive_scan_1024(a.cell_start, nc, sscan);
  for (int f = t; f < a.NF; f += 1024) {
    const int cidx = cell_of(f);
    const int slot = a.cell_start[cidx] + atomicAdd(&a.cell_fill[cidx], 1);
    a.cell_c[slot] = make_float4(a.pos[3 * f], a.pos[3 * f + 1], a.pos[3 * f + 2], __int_as_float(f));
  }
  if (!a.near_inline) return;
  __threadfence();
  __syncthreads();
  const int ne = knn_near_cells(h);
  for (int e = t; e < ne; e += 1024) a.near_start[e] = knn_near_visit(a, h, e, false);
  __threadfence();
  __syncthreads();
  block_exclusive_scan_1024(a.near_start, ne, sscan);
  for (int e = t; e < ne; e += 1024) (void)knn_near_visit(a, h, e, true);
}
// the candidate lists of a large map, on the whole device: count -> scan (one workgroup) -> fill
__global__ __launch_bounds__(256) void k_knn_near_count(KnnArgs a) {
  const KnnGridHdr h = *a.grid;
  const int ne = knn_near_cells(h);
  for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < ne; e += gridDim.x * blockDim.x) a.near_start[e] = knn_near_visit(a, h, e, false);
}
__global__ __launch_bounds__(1024) void k_knn_near_scan(KnnArgs a) {
  __shared__ int sscan[1024];
  const KnnGridHdr h = *a.grid;
  block_exclusive_scan_1024(a.near_start, knn_near_cells(h), sscan);
}
__global__ __launch_bounds__(256) void k_knn_near_fill(KnnArgs a) {
  const KnnGridHdr h = *a.grid;
  const int ne = knn_near_cells(h);
  for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < ne; e += gridDim.x * blockDim.x) (void)knn_near_visit(a, h, e, true);
}

// ---- point source -------------------------------------------------------------------------------------------------------
// generated points (a.gen): the sampler's arithmetic (ray_geom, strat_t, sample_world_point of ngm_device.h), so that the fused
// image path gives the points -- bit for bit -- that k_sample_rays would have written
// per ray of the block, once: direction (camera frame) and the stratum's (near, span, span / S) -- the ray-level part of
// ray_geom / strat_t, which every one of the ray's S samples would otherwise repeat (three IEEE divisions, a root, two
// int64 conversions)
__global__ __launch_bounds__(256) void k_knn_raytab(KnnArgs a) {
#pragma clang fp contract(off)
  const int64_t nr = a.P / a.S;
  for (int64_t ray = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; ray < nr; ray += (int64_t)gridDim.x * blockDim.x) {
    const RayGeom rg = ray_geom(a.rc, a.rays, ray, false);
    a.ray_dir[3 * ray] = rg.dx; a.ray_dir[3 * ray + 1] = rg.dy; a.ray_dir[3 * ray + 2] = rg.dz;
    const float span = rg.far - rg.near;
    a.ray_tab[ray] = make_float4(rg.near, span, span / (float)a.S, 0.f);
  }
}
__device__ __forceinline__ void knn_point_draw(const KnnArgs& a, uint64_t poff, int64_t p, float* x, float* y, float* z) {
#pragma clang fp contract(off)
  const uint32_t ray = (uint32_t)p / (uint32_t)a.S;
  const int e = (int)((uint32_t)p - ray * (uint32_t)a.S);
  RayGeom rg;
  rg.dx = a.ray_dir[3 * ray]; rg.dy = a.ray_dir[3 * ray + 1]; rg.dz = a.ray_dir[3 * ray + 2];
  const float4 st = a.ray_tab[ray];                      // near, span, delta
  // strat_t (ngm_device.h) on the ray's precomputed span and delta: t = (delta u + lin_e span) + near
  // jitter() of the sampler with its element index ray * S + e == p taken as is (no 64-bit multiply per sample)
  const float u = a.rays.u_coarse ? a.rays.u_coarse[p] : philox_uniform(a.rays.philox_seed, poff, (uint64_t)p, 0u);
  const float b = strat_lin(a.rays.lin_coarse, a.S, e) * st.y;
  const float du = st.z * u;
  const float sm = du + b;
  const float t = sm + st.x;
  a.dist[p] = t;
  sample_world_point(a.rays, rg, ray, t, x, y, z);
}
__device__ __forceinline__ void knn_point_again(const KnnArgs& a, int64_t p, float* x, float* y, float* z) {
  const uint32_t ray = (uint32_t)p / (uint32_t)a.S;
  RayGeom rg;
  rg.dx = a.ray_dir[3 * ray]; rg.dy = a.ray_dir[3 * ray + 1]; rg.dz = a.ray_dir[3 * ray + 2];
  sample_world_point(a.rays, rg, ray, a.dist[p], x, y, z);
}

// K nearest centres of every point, exact.  The (2 R + 1)^3 block of cells around the point's cell holds every centre
// within R cell edges c of it.  Phase A decides the inside test of models.py:369 from the candidate list of the point's cell
// (k_knn_grid: every centre a point of that cell can be within the mask radius of) -- four out of five samples of an image
// fall on an empty list and stop here -- and seeds the neighbour list with the centres within the radius.  Phase B (inside
// points with fewer than K centres within the radius) finds the K nearest: cells whose nearest face is already farther
// than the current K-th best are skipped, rows are narrowed in x the same way, and the block grows ring by ring until the
// K-th neighbour found lies within R c (R = 1 wherever the map is as dense as its cover grid).  Ties in distance go to the
// lower field index, as in a loop over the fields in order: the brute-force result bit for bit.
// GL: the grid (cell offsets + centres) staged in LDS -- maps up to a few thousand fields; larger ones read it through L1 / L2.
// KK: the number of neighbours as a compile-time constant (the list insertion is the hot code: a wave executes it for the
// union of its lanes' candidates)
template <bool GL, int KK>
__global__ __launch_bounds__(256) void k_knn_assign(KnnArgs a) {
  extern __shared__ __attribute__((aligned(16))) int knn_lds[];
  const KnnGridHdr h = *a.grid;
  int* hist = knn_lds;                                  // NF (hist_in_lds)
  const int hist_n = a.hist_in_lds ? a.NF : 0;
  const int* cs = a.cell_start;
  const float4* cc = a.cell_c;
  if (a.hist_in_lds)
    for (int i = threadIdx.x; i < a.NF; i += blockDim.x) hist[i] = 0;
  if constexpr (GL) {
    float4* lcc = reinterpret_cast<float4*>(knn_lds + ((hist_n + 3) & ~3));
    int* lcs = reinterpret_cast<int*>(lcc + a.NF);
    for (int i = threadIdx.x; i < a.NF; i += blockDim.x) lcc[i] = a.cell_c[i];
    for (int i = threadIdx.x; i <= h.ncells; i += blockDim.x) lcs[i] = a.cell_start[i];
    cs = lcs; cc = lcc;
  }
  __syncthreads();
  // KK <= 8: K = KK, everything unrolled over the list.  KK = 16 (round 6): the instance for K = 9..16 -- the list has 16 slots,
  // the run-time K says how many of them count (the reference takes any num_knn, models.py:354-366)
  const int K = (KK > KNN_MAXK) ? a.K : KK;
  auto kth = [&](const auto& arr) __attribute__((always_inline)) {       // arr[K - 1] without indexing registers by a run-time value
    auto v = arr[KK - 1];
    if constexpr (KK > KNN_MAXK) {
#pragma unroll
      for (int k = 0; k < KK; ++k) v = (k == K - 1) ? arr[k] : v;
    }
    return v;
  };
  const int maxR = max(h.nx, max(h.ny, h.nz));
  const float c2 = h.c * h.c;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  const uint64_t poff = a.gen ? philox_launch_offset(a.rays) : 0ull;
  for (int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; p < a.P; p += stride) {
    float x, y, z;
    if (a.gen) knn_point_draw(a, poff, p, &x, &y, &z);
    else { x = a.points[3 * p]; y = a.points[3 * p + 1]; z = a.points[3 * p + 2]; }
    const float gx = (x - h.x0) * h.inv_c, gy = (y - h.y0) * h.inv_c, gz = (z - h.z0) * h.inv_c;
    // cell of the point; far outside the grid (or NaN) it is clamped to two rings beyond: nothing is within c then
    const float fx = fminf(fmaxf(floorf(gx), -2.0f), (float)h.nx + 1.0f), fy = fminf(fmaxf(floorf(gy), -2.0f), (float)h.ny + 1.0f),
                fz = fminf(fmaxf(floorf(gz), -2.0f), (float)h.nz + 1.0f);
    const int ix = (int)fx, iy = (int)fy, iz = (int)fz;
    // the centres this point can be within the mask radius of: the list of its cell (none two rings outside the grid)
    int nj0 = 0, nj1 = 0;
    if (!(ix < -1 || ix > h.nx || iy < -1 || iy > h.ny || iz < -1 || iz > h.nz)) {
      // sub-cell of the point inside its (extended) cell; a point that fp32 puts just across a sub-cell face is covered by
      // the margin of the lists' reach
      const int ux = min(max((int)floorf(gx * KNN_SUB), ix * KNN_SUB), ix * KNN_SUB + KNN_SUB - 1),
                uy = min(max((int)floorf(gy * KNN_SUB), iy * KNN_SUB), iy * KNN_SUB + KNN_SUB - 1),
                uz = min(max((int)floorf(gz * KNN_SUB), iz * KNN_SUB), iz * KNN_SUB + KNN_SUB - 1);
      const int e = ((uz + KNN_SUB) * ((h.ny + 2) * KNN_SUB) + (uy + KNN_SUB)) * ((h.nx + 2) * KNN_SUB) + (ux + KNN_SUB);
      nj0 = a.near_start[e]; nj1 = a.near_start[e + 1];
    }
    if (nj0 == nj1) {                                            // nowhere near a field: outside, no search
#pragma unroll
      for (int k = 0; k < KK; ++k)
        if (k < K) a.pair_field[p * K + k] = -1;
      continue;
    }
    float bd[KK]; int bi[KK];
    float worst; int worst_i;
    // scan the block of radius R; cells / row ends whose nearest face is not closer than `lim2` (world units squared; a
    // fixed limit, or the running K-th best when `dynamic`) are skipped -- they cannot change the result
    auto scan = [&](int R, float lim2, bool dynamic) __attribute__((always_inline)) {
#pragma unroll
      for (int k = 0; k < KK; ++k) { bd[k] = INFINITY; bi[k] = -1; }
      worst = INFINITY; worst_i = -1;
      const int xs0 = max(ix - R, 0), xe0 = min(ix + R, h.nx - 1);
      if (xs0 > xe0) return;
      for (int oz = 0; oz <= 2 * R; ++oz) {
        // own plane first, then alternating outwards: the running bound shrinks early
        const int cz = iz + ((oz & 1) ? (oz + 1) / 2 : -(oz / 2));
        if (cz < 0 || cz >= h.nz) continue;
        const float lz = fmaxf(fmaxf((float)cz - gz, gz - (float)(cz + 1)), 0.f);
        for (int oy = 0; oy <= 2 * R; ++oy) {
          const int cy = iy + ((oy & 1) ? (oy + 1) / 2 : -(oy / 2));
          if (cy < 0 || cy >= h.ny) continue;
          const float ly = fmaxf(fmaxf((float)cy - gy, gy - (float)(cy + 1)), 0.f);
          const float lat2 = (ly * ly + lz * lz) * c2 * 0.9999f;
          const float lim = dynamic ? fminf(worst, lim2) : lim2;
          if (!(lat2 < lim)) continue;
          // cells of this row that can hold a centre closer than lim: |x - centre.x| < sqrt(lim - lat2)
          int xs = xs0, xe = xe0;
          if (lim < INFINITY) {
            const float rem = __builtin_amdgcn_sqrtf(lim - lat2) * h.inv_c * 1.001f + 1e-4f;     // (1-ulp hardware root: the margin covers it)
            xs = max(xs, (int)floorf(gx - rem)); xe = min(xe, (int)floorf(gx + rem));
            if (xs > xe) continue;
          }
          const int rowb = (cz * h.ny + cy) * h.nx;
          const int je = cs[rowb + xe + 1];
          for (int j = cs[rowb + xs]; j < je; ++j) {
            const float4 c = cc[j];
            const float dx = x - c.x, dy = y - c.y, dz = z - c.z;
            float d = dx * dx + dy * dy + dz * dz;
            int id = __float_as_int(c.w);
            if (d < worst || (d == worst && id < worst_i)) {
              // sorted insertion; equal distances keep the lower field index first
#pragma unroll
              for (int k = 0; k < KK; ++k) {
                if (k < K && (d < bd[k] || (d == bd[k] && id < bi[k]))) {
                  const float td = bd[k]; const int ti = bi[k]; bd[k] = d; bi[k] = id; d = td; id = ti;
                }
              }
              worst = kth(bd);
              worst_i = kth(bi);
            }
          }
        }
      }
    };
    // ---- phase A: is any centre within the mask radius?  Every such centre is in the cell's list; the ones within the
    // radius also start the neighbour list -- where the map is as dense as its cover grid the K nearest are among them and
    // phase B is not needed at all.  A wave walks the union of its lanes' lists and nearly every candidate is within the
    // radius of SOME lane, so the insertion is branch-free: (distance bits, field index) as one 64-bit key -- distances are
    // non-negative, so the integer order of the keys is the order (distance, then lower field index) of the list -- and a
    // chain of K min / max pairs; candidates beyond the radius carry the all-ones key of an empty slot.
    const float rad2 = a.radius * a.radius * 1.0002f;
    unsigned long long key[KK];
#pragma unroll
    for (int k = 0; k < KK; ++k) key[k] = ~0ull;
    for (int j = nj0; j < nj1; ++j) {
      const float4 c = a.near_c[j];
      const float dx = x - c.x, dy = y - c.y, dz = z - c.z;
      const float d = dx * dx + dy * dy + dz * dz;
      unsigned long long kv = ((unsigned long long)__float_as_uint(d) << 32) | (unsigned long long)__float_as_uint(c.w);
      kv = (d < rad2) ? kv : ~0ull;
#pragma unroll
      for (int k = 0; k < KK; ++k) {
        const unsigned long long lo = kv < key[k] ? kv : key[k], hi = kv < key[k] ? key[k] : kv;
        key[k] = lo; kv = hi;
      }
    }
#pragma unroll
    for (int k = 0; k < KK; ++k) {
      bi[k] = (int)(unsigned)(key[k] & 0xffffffffull);
      bd[k] = (key[k] == ~0ull) ? INFINITY : __uint_as_float((unsigned)(key[k] >> 32));
    }
    if constexpr (KK > KNN_MAXK) {       // slots beyond K are not part of the list (phase B's insertion never touches them)
#pragma unroll
      for (int k = 0; k < KK; ++k)
        if (k >= K) { bd[k] = INFINITY; bi[k] = -1; }
    }
    worst = kth(bd); worst_i = kth(bi);
    const float dmin = bd[0];                                    // (no centre within the radius: infinity, outside)
    bool inside = sqrtf(dmin) < a.radius;                        // models.py:369 (the same squared distance as the list's bd[0])
    // every centre closer than the radius has been seen: a K-th neighbour inside the radius is the K-th nearest
    if (inside && !(worst < a.radius * a.radius)) {
      // ---- phase B: the K nearest of an inside point with fewer than K centres within the radius
      for (int R = 1;; ++R) {
        scan(R, INFINITY, true);
        // exact when the K-th found is within the block's guaranteed reach R c (a little less: fp32 cell arithmetic) or the
        // block has swallowed the grid
        const float reach = (float)R * h.c * 0.9999f;
        if (worst <= reach * reach || R > maxR + 2) break;
      }
    }
    if (inside) inside = sqrtf(bd[0]) < a.radius;                 // models.py:369 on the distance the blend uses
    if (!inside) {                                                // outside every field: the blend writes outside_value
#pragma unroll
      for (int k = 0; k < KK; ++k)
        if (k < K) a.pair_field[p * K + k] = -1;     // (k_knn_blend reads the weights of inside points only)
      continue;
    }
    // softmax(-distance_factor * dist) over the K neighbours (models.py:384); dist[0] is the distance phase A tested
    float dist[KK], mx = -INFINITY, e[KK], sum = 0.f;
#pragma unroll
    for (int k = 0; k < KK; ++k)
      if (k < K) { dist[k] = sqrtf(bd[k]); mx = fmaxf(mx, -a.distance_factor * dist[k]); }
#pragma unroll
    for (int k = 0; k < KK; ++k)
      if (k < K) { e[k] = expf(-a.distance_factor * dist[k] - mx); sum += e[k]; }
#pragma unroll
    for (int k = 0; k < KK; ++k) {
      if (k < K) {
        a.pair_field[p * K + k] = bi[k];
        a.pair_w[p * K + k] = e[k] / sum;
        if (bi[k] >= 0) { if (a.hist_in_lds) atomicAdd(&hist[bi[k]], 1); else atomicAdd(&a.counts[(int64_t)(blockIdx.x % KNN_COUNT_REPL) * a.NF + bi[k]], 1); }
      }
    }
  }
  if (a.hist_in_lds) {
    __syncthreads();
    for (int i = threadIdx.x; i < a.NF; i += blockDim.x)
      if (hist[i]) atomicAdd(&a.counts[(int64_t)(blockIdx.x % KNN_COUNT_REPL) * a.NF + i], hist[i]);
  }
}

// exclusive scans over the fields (segment offsets, tile offsets): one wave, 64 fields per round
__global__ void k_knn_offsets(KnnArgs a) {
  const int lane = threadIdx.x;          // launched with 64 threads
  int so = 0, to = 0;
  for (int f0 = 0; f0 < a.NF; f0 += 64) {
    const int f = f0 + lane;
    int c = 0;
    if (f < a.NF) {
#pragma unroll
      for (int r = 0; r < KNN_COUNT_REPL; ++r) { c += a.counts[(int64_t)r * a.NF + f]; a.counts[(int64_t)r * a.NF + f] = 0; }   // summed, and ready for the next block
    }
    const int t = (c + KNN_TILE - 1) / KNN_TILE;
    int sc = c, st = t;                  // inclusive wave scans
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const int oc = __shfl_up(sc, d, 64), ot = __shfl_up(st, d, 64);
      if (lane >= d) { sc += oc; st += ot; }
    }
    if (f < a.NF) { a.seg_off[f] = so + sc - c; a.tile_off[f] = to + st - t; a.cursor[f] = 0; }
    so += __shfl(sc, 63, 64); to += __shfl(st, 63, 64);
  }
  if (lane == 0) { a.seg_off[a.NF] = so; a.tile_off[a.NF] = to; }
}

// Counting-sort scatter in two levels: every workgroup ranks its SC_ITEMS * 256 pairs per field with LDS atomics,
// reserves one contiguous range per field with ONE global atomic, then writes.  (One global atomic per pair on the
// per-field cursor serialised badly -- neighbouring pixels hit the same field -- and took 80 % of render_image.)
#ifndef SC_ITEMS
#define SC_ITEMS 32
#endif
__global__ __launch_bounds__(256) void k_knn_scatter(KnnArgs a) {
  extern __shared__ int sc_lds[];
  int* hist = sc_lds;            // NF: pairs of this workgroup per field, then the reserved global base
  const int64_t n = a.P * a.K;
  const int64_t base = (int64_t)blockIdx.x * (SC_ITEMS * 256);
  if (!a.hist_in_lds) {          // maps too large for an LDS histogram (> ~38 000 fields): one global atomic per pair
#pragma unroll
    for (int k = 0; k < SC_ITEMS; ++k) {
      const int64_t i = base + k * 256 + threadIdx.x;
      const int fl = (i < n) ? a.pair_field[i] : -1;
      if (fl >= 0) a.sorted[a.seg_off[fl] + atomicAdd(&a.cursor[fl], 1)] = (int)i;
    }
    return;
  }
  for (int i = threadIdx.x; i < a.NF; i += blockDim.x) hist[i] = 0;
  __syncthreads();
  int fld[SC_ITEMS], rnk[SC_ITEMS];
#pragma unroll
  for (int k = 0; k < SC_ITEMS; ++k) {
    const int64_t i = base + k * 256 + threadIdx.x;
    fld[k] = (i < n) ? a.pair_field[i] : -1;
    rnk[k] = (fld[k] >= 0) ? atomicAdd(&hist[fld[k]], 1) : 0;
  }
  __syncthreads();
  for (int f = threadIdx.x; f < a.NF; f += blockDim.x) {
    const int c = hist[f];
    hist[f] = a.seg_off[f] + (c ? atomicAdd(&a.cursor[f], c) : 0);
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < SC_ITEMS; ++k)
    if (fld[k] >= 0) a.sorted[hist[fld[k]] + rnk[k]] = (int)(base + k * 256 + threadIdx.x);
}

// Persistent workgroups: workgroup b owns a contiguous run of the (field-ordered) KNN_TILE-pair tiles, so it stages a
// field's weights once per field it meets (one or two per run), not once per tile, and the runs are equal to within one
// small tile.  NT: threads per workgroup -- the bf16-split variant keeps 83 KB of weights per workgroup, so one workgroup
// per CU: eight waves (two per SIMD, as in the training forward) instead of four share them.
template <int MI, int MH, int L, bool NEED_COS, int HASH, int SKIP, bool B3 = false, int NT = NGM_BLOCK>
__global__ __launch_bounds__(NT) void k_knn_eval(KnnArgs a) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int total_tiles = a.tile_off[a.NF];
  const int per = (total_tiles + (int)gridDim.x - 1) / (int)gridDim.x;
  const int t0 = blockIdx.x * per, t1 = min(total_tiles, t0 + per);
  if (t0 >= t1) return;
  // field of the first tile: last f with tile_off[f] <= t0
  int lo = 0, hi = a.NF - 1;
  while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (a.tile_off[mid] <= t0) lo = mid; else hi = mid - 1; }
  int f = lo, cur = -1;
  ngm_u32x4* const b3w = reinterpret_cast<ngm_u32x4*>(sm + FieldLds<MI, MH, L, SKIP == 2>::TOTAL);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float div, off;
  scale_consts(a.fc.scale_mode, a.fc.field_radius, &div, &off);
  float px = 0, py = 0, pz = 0, qw = 1, qx = 0, qy = 0, qz = 0;
  HashCtx hc = make_hash_ctx(a.fc, a.pr, 0, nullptr);
  TriCtx tc = make_tri_ctx(a.fc, a.pr, 0, nullptr);
  for (int t = t0; t < t1; ++t) {
    while (a.tile_off[f + 1] <= t) ++f;                          // (fields without pairs own no tile)
    if (f != cur) {
      if (cur >= 0) __syncthreads();                             // every wave is done with the previous field's weights
      const int64_t row = a.pr.field_index ? a.pr.field_index[f] : f;
      {
        FieldStage<MI, MH, L, SKIP == 2> fstage;     // every parameter load in flight at once, then the permuting LDS writes
        fstage.issue(a.fc, a.pr, row);
        fstage.commit(sm, a.fc);
      }
      __syncthreads();
      if constexpr (B3) {        // ngm_matmul_mode BF16X3: bf16 weight planes behind the fp32 fragments (ngm_field.h)
        b3_build_planes<MI, MH, L>(sm, b3w);
        __syncthreads();
      }
      px = a.pos[3 * f]; py = a.pos[3 * f + 1]; pz = a.pos[3 * f + 2];
      qw = a.quat[4 * f]; qx = a.quat[4 * f + 1]; qy = a.quat[4 * f + 2]; qz = a.quat[4 * f + 3];
      hc = make_hash_ctx(a.fc, a.pr, row, nullptr);
      tc = make_tri_ctx(a.fc, a.pr, row, nullptr);
      cur = f;
    }
    const int tile = t - a.tile_off[f];
    const int beg = a.seg_off[f] + tile * KNN_TILE, end = min(a.seg_off[f + 1], beg + KNN_TILE);
    for (int base = beg + wave * 64; base < end; base += NT) {
      const int idx = base + lane;
      const bool valid = idx < end;
      float x = 0, y = 0, z = 0;
      int pair = 0;
      if (valid) {
        pair = a.sorted[idx];
        const int64_t p = pair / a.K;
        float wx, wy, wz;
        if (a.gen) knn_point_again(a, p, &wx, &wy, &wz);
        else { wx = a.points[3 * p]; wy = a.points[3 * p + 1]; wz = a.points[3 * p + 2]; }
        const Vec3 v = scaled_local_point(Vec3{wx, wy, wz}, true, px, py, pz, qw, qx, qy, qz, div, off);   // models.py:377-381
        x = v.x; y = v.y; z = v.z;
      }
      const float4 o = eval_64<MI, MH, L, NEED_COS, HASH, SKIP, B3>(sm, lane, x, y, z, &hc, nullptr, nullptr, b3w, &tc);
      if (valid) a.pair_out[pair] = o;      // (write-through here measured slower: 11.7 vs 11.4 ms per image -- the quadrature reads these soon, from L2)
    }
  }
}

__global__ __launch_bounds__(256) void k_knn_blend(KnnArgs a) {
  for (int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; p < a.P; p += (int64_t)gridDim.x * blockDim.x) {
    float4 o = make_float4(a.outside_value, a.outside_value, a.outside_value, a.outside_value);   // models.py:401
    if (a.pair_field[p * a.K] >= 0) {
      o = make_float4(0.f, 0.f, 0.f, 0.f);
      for (int k = 0; k < a.K; ++k) {
        const float w = a.pair_w[p * a.K + k];
        const float4 v = a.pair_out[p * a.K + k];
        o.x = fmaf(w, v.x, o.x); o.y = fmaf(w, v.y, o.y); o.z = fmaf(w, v.z, o.z); o.w = fmaf(w, v.w, o.w);
      }
    }
    reinterpret_cast<float4*>(a.out)[p] = o;
  }
}

// cells of the uniform grid over the centres: ~8 per field (the cover grid of rm.py:299 puts about one centre per cell; rooms
// are not cubes), bounded so that the build stays a one-workgroup job
static int knn_max_cells(int num_fields) { return (int)std::min<int64_t>(std::max<int64_t>(8 * (int64_t)num_fields, 512), 1 << 18); }
int64_t ngm_knn_workspace_bytes(int num_fields, int64_t P, int K) {
  const int64_t n = P * K;
  const int64_t mc = knn_max_cells(num_fields);
  return 256 * 12 + 4 * (n + 255) + 4 * (n + 255) + 4 * (int64_t)((3 + KNN_COUNT_REPL) * num_fields + 64) + 4 * (n + 255) + 16 * (n + 16) +
         256 + 4 * (2 * mc + 2) + 16 * ((int64_t)num_fields + 16) + 16 * KNN_SUB * KNN_SUB * KNN_SUB * mc + 16 * KNN_NEAR_PER_FIELD * (int64_t)num_fields + 1024;
}
// + the distances of the generated samples and the ray directions of one block of rays (ngm_render_eval_knn)
int64_t ngm_knn_render_workspace_bytes(int num_fields, int ray_block, int S, int K) {
  const int64_t P = (int64_t)ray_block * S;
  return ngm_knn_workspace_bytes(num_fields, P, K) + 4 * (P + 64) + 28 * ((int64_t)ray_block + 64) + 1024;
}

// compute units of the device: the persistent evaluation kernel runs one workgroup per CU in its 83 KB bf16-split variant
// (eight waves), four per CU in the fp32 variants (<= 35 KB, four waves each)
static int knn_num_cus() {
  static int n = 0;
  if (!n) {
    int dev = 0;
    hipDeviceProp_t prop;
    n = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0)
            ? prop.multiProcessorCount : 256;
    (void)hipGetLastError();
  }
  return n;
}

template <int MI, int MH, int L>
static int launch_eval(const KnnArgs& a, int max_tiles, hipStream_t st) {
  const int grid = std::max(1, std::min(max_tiles, 4 * knn_num_cus())), grid_b3 = std::max(1, std::min(max_tiles, knn_num_cus()));
#define NGM_KE(NC, HS, SK)                                                                                             \
  do {                                                                                                                 \
    const size_t lds = FieldLds<MI, MH, L, (SK) == 2>::TOTAL * sizeof(float);                                          \
    (void)hipFuncSetAttribute((const void*)k_knn_eval<MI, MH, L, NC, HS, SK>, hipFuncAttributeMaxDynamicSharedMemorySize, \
                              (int)lds);                                                                               \
    hipLaunchKernelGGL((k_knn_eval<MI, MH, L, NC, HS, SK>), dim3(grid), dim3(NGM_BLOCK), lds, st, a);                  \
  } while (0)
  const int sk = a.fc.skip_mode;
  g_ngm_last_matmul[2] = NGM_MATMUL_F32;
  if constexpr (MI == 2 && MH == 2 && L <= 2) {
    if ((a.fc.matmul_mode == NGM_MATMUL_BF16X3 || a.fc.matmul_mode == NGM_MATMUL_AUTO) && sk == NGM_SKIP_NO &&
        (a.fc.encoding == NGM_ENC_FOURIER || a.fc.encoding == NGM_ENC_NONE)) {
      g_ngm_last_matmul[2] = NGM_MATMUL_BF16X3;
      const size_t lds = FieldLds<MI, MH, L>::TOTAL * sizeof(float) + (size_t)B3Lds<MI, MH, L>::TOTAL * 16;
      (void)hipFuncSetAttribute((const void*)k_knn_eval<MI, MH, L, false, 0, 0, true, KNN_EVAL_B3_THREADS>,
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      hipLaunchKernelGGL((k_knn_eval<MI, MH, L, false, 0, 0, true, KNN_EVAL_B3_THREADS>), dim3(grid_b3), dim3(KNN_EVAL_B3_THREADS), lds, st, a);
      return 0;
    }
  }
  if (a.fc.encoding == NGM_ENC_PERMUTO) {
    if constexpr (MI == 1) { if (sk == NGM_SKIP_ADD) NGM_KE(false, 1, 1); else if (sk == NGM_SKIP_CONCAT) NGM_KE(false, 1, 2); else NGM_KE(false, 1, 0); }
    else return NGM_E_UNSUPPORTED;
  } else if (a.fc.encoding == NGM_ENC_TRIPLANE) {
    if (sk == NGM_SKIP_ADD) NGM_KE(false, 2, 1); else if (sk == NGM_SKIP_CONCAT) NGM_KE(false, 2, 2); else NGM_KE(false, 2, 0);
  } else if (a.fc.encoding == NGM_ENC_NERF) {
    if (sk == NGM_SKIP_ADD) NGM_KE(true, 0, 1); else if (sk == NGM_SKIP_CONCAT) NGM_KE(true, 0, 2); else NGM_KE(true, 0, 0);
  } else {
    if (sk == NGM_SKIP_ADD) NGM_KE(false, 0, 1); else if (sk == NGM_SKIP_CONCAT) NGM_KE(false, 0, 2); else NGM_KE(false, 0, 0);
  }
#undef NGM_KE
  return 0;
}

// workspace carving shared by the two entry points; `a.NF`, `a.K`, `a.P` (the largest P of the call) are set
static void knn_carve(KnnArgs& a, void* workspace, char** end) {
  const int64_t n = a.P * a.K;
  char* w = reinterpret_cast<char*>(((int64_t)workspace + 255) / 256 * 256);
  auto carve = [&](int64_t bytes) { char* p = w; w += (bytes + 255) / 256 * 256; return p; };
  a.pair_field = reinterpret_cast<int*>(carve(4 * n));
  a.pair_w = reinterpret_cast<float*>(carve(4 * n));
  a.counts = reinterpret_cast<int*>(carve(4 * (int64_t)KNN_COUNT_REPL * a.NF));
  a.cursor = reinterpret_cast<int*>(carve(4 * a.NF));
  a.seg_off = reinterpret_cast<int*>(carve(4 * (a.NF + 1)));
  a.tile_off = reinterpret_cast<int*>(carve(4 * (a.NF + 1)));
  a.sorted = reinterpret_cast<int*>(carve(4 * n));
  a.pair_out = reinterpret_cast<float4*>(carve(16 * n));
  a.max_cells = knn_max_cells(a.NF);
  a.grid = reinterpret_cast<KnnGridHdr*>(carve(256));
  a.cell_start = reinterpret_cast<int*>(carve(4 * ((int64_t)a.max_cells + 1)));
  a.cell_fill = reinterpret_cast<int*>(carve(4 * (int64_t)a.max_cells));
  a.cell_c = reinterpret_cast<float4*>(carve(16 * (int64_t)a.NF));
  a.near_start = reinterpret_cast<int*>(carve(4 * (KNN_SUB * KNN_SUB * KNN_SUB * 4 * (int64_t)a.max_cells + 1)));
  a.near_c = reinterpret_cast<float4*>(carve(16 * KNN_NEAR_PER_FIELD * (int64_t)a.NF));
  *end = w;
}

// grid build (optional) -> assignment -> offsets -> scatter -> per-field evaluation of the pairs; the blend is the caller's
static int knn_stages(KnnArgs& a, bool build_grid, hipStream_t st) {
  const int64_t n = a.P * a.K;
  // (the pair counts are zero here: k_knn_grid zeroes them, k_knn_offsets re-zeroes them after use)
  const int pb = (int)std::min<int64_t>((a.P + 255) / 256, KNN_ASSIGN_MAX_WG);
  // per-workgroup field histograms (assignment, scatter) in LDS while they fit (150 KB = 38 400 fields); beyond: global atomics
  const size_t lds_h = (size_t)a.NF * 4;
  a.hist_in_lds = lds_h <= 150 * 1024 ? 1 : 0;
  static const hipError_t attr_a = [] {
    hipError_t e = hipSuccess;
#define NGM_KA(G, KK_, B) do { const hipError_t x = hipFuncSetAttribute(reinterpret_cast<const void*>(k_knn_assign<G, KK_>), hipFuncAttributeMaxDynamicSharedMemorySize, B); if (x != hipSuccess) e = x; } while (0)
    NGM_KA(false, 1, 150 * 1024); NGM_KA(false, 2, 150 * 1024); NGM_KA(false, 3, 150 * 1024); NGM_KA(false, 4, 150 * 1024);
    NGM_KA(true, 1, 64 * 1024); NGM_KA(true, 2, 64 * 1024); NGM_KA(true, 3, 64 * 1024); NGM_KA(true, 4, 64 * 1024);
    NGM_KA(false, 5, 150 * 1024); NGM_KA(false, 6, 150 * 1024); NGM_KA(false, 7, 150 * 1024); NGM_KA(false, 8, 150 * 1024);
    NGM_KA(true, 5, 64 * 1024); NGM_KA(true, 6, 64 * 1024); NGM_KA(true, 7, 64 * 1024); NGM_KA(true, 8, 64 * 1024);
    NGM_KA(false, 16, 150 * 1024); NGM_KA(true, 16, 64 * 1024);
#undef NGM_KA
    return e;
  }();
  // grid in LDS while histogram + centres + cell offsets stay within 64 KB (two workgroups per CU keep their latency hiding)
  const size_t lds_grid = (((size_t)a.NF + 3) & ~(size_t)3) * 4 + (size_t)a.NF * 16 + ((size_t)a.max_cells + 1) * 4;
  const bool grid_in_lds = a.hist_in_lds && lds_grid <= 64 * 1024;
  static const hipError_t attr_s = hipFuncSetAttribute(reinterpret_cast<const void*>(k_knn_scatter),
                                                       hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
  if (attr_a != hipSuccess || attr_s != hipSuccess) return NGM_E_HIP;
  const int K = a.K;
  {
    NgmProfScope prof_(NGM_K_KNN_ASSIGN, st);
    if (build_grid) {
      a.near_inline = a.NF <= 2048 ? 1 : 0;
      hipLaunchKernelGGL(k_knn_grid, dim3(1), dim3(1024), 0, st, a);
      if (!a.near_inline) {
        const int nb = (int)std::min<int64_t>((KNN_SUB * KNN_SUB * KNN_SUB * 4 * (int64_t)a.max_cells + 255) / 256, 2048);
        hipLaunchKernelGGL(k_knn_near_count, dim3(nb), dim3(256), 0, st, a);
        hipLaunchKernelGGL(k_knn_near_scan, dim3(1), dim3(1024), 0, st, a);
        hipLaunchKernelGGL(k_knn_near_fill, dim3(nb), dim3(256), 0, st, a);
      }
    }
#define NGM_KL(KK_)                                                                                                    \
    do {                                                                                                               \
      if (grid_in_lds) hipLaunchKernelGGL((k_knn_assign<true, KK_>), dim3(std::max(pb, 1)), dim3(256), lds_grid, st, a);  \
      else hipLaunchKernelGGL((k_knn_assign<false, KK_>), dim3(std::max(pb, 1)), dim3(256), a.hist_in_lds ? lds_h : 0, st, a); \
    } while (0)
    if (K == 1) NGM_KL(1); else if (K == 2) NGM_KL(2); else if (K == 3) NGM_KL(3); else if (K == 4) NGM_KL(4);
    else if (K == 5) NGM_KL(5); else if (K == 6) NGM_KL(6); else if (K == 7) NGM_KL(7); else if (K == 8) NGM_KL(8);
    else if (K <= 16) NGM_KL(16); else return NGM_E_UNSUPPORTED;
#undef NGM_KL
  }
  if (hipGetLastError() != hipSuccess) return NGM_E_HIP;      // an over-sized LDS request fails here, not four launches later
  hipLaunchKernelGGL(k_knn_offsets, dim3(1), dim3(64), 0, st, a);
  const int nb = (int)((n + SC_ITEMS * 256 - 1) / (SC_ITEMS * 256));
  hipLaunchKernelGGL(k_knn_scatter, dim3(std::max(nb, 1)), dim3(256), a.hist_in_lds ? lds_h : 0, st, a);
  const int max_tiles = (int)std::min<int64_t>((n + KNN_TILE - 1) / KNN_TILE + a.NF, 0x7fffffff);
  const FieldShape s = field_shape(&a.fc);
  int le = NGM_E_UNSUPPORTED;
  {
    NgmProfScope prof_eval_(NGM_K_KNN_EVAL, st);
    if (s.MI == 2 && s.MH == 2 && s.L == 2) le = launch_eval<2, 2, 2>(a, max_tiles, st);
#ifndef NGM_FAST_BUILD
    else if (s.MI == 2 && s.MH == 2 && s.L == 1) le = launch_eval<2, 2, 1>(a, max_tiles, st);
    else if (s.MI == 1 && s.MH == 1 && s.L == 1) le = launch_eval<1, 1, 1>(a, max_tiles, st);
    else if (s.MI == 1 && s.MH == 1 && s.L == 2) le = launch_eval<1, 1, 2>(a, max_tiles, st);
    else if (s.MI == 2 && s.MH == 2 && s.L == 3) le = launch_eval<2, 2, 3>(a, max_tiles, st);
#endif
  }
  return le;
}

static void knn_common(KnnArgs& a, const ngm_field_cfg* fc, const ngm_params* pr, int num_fields, const float* pos,
                       const float* quat, int K, float distance_factor, float outside_value, float mask_radius) {
  memset(&a, 0, sizeof(a));
  a.fc = *fc; a.pr = *pr; a.NF = num_fields; a.K = K; a.pos = pos; a.quat = quat;
  a.distance_factor = distance_factor; a.outside_value = outside_value; a.radius = mask_radius;
  // cell edge: the spacing of the reference's cover grid (2 r / sqrt 3, rm.py:299), never below the mask radius (the inside
  // test is decided in the first ring); a non-positive radius (no field contains anything) still needs a positive edge
  a.cell_size = std::max(std::max(1.1547005f * fc->field_radius, mask_radius), 1e-6f);
}

int ngm_launch_knn(const ngm_field_cfg* fc, const ngm_params* pr, int num_fields, int64_t P, const float* points,
                   const float* pos, const float* quat, int K, float distance_factor, float outside_value, float mask_radius,
                   float* out, void* workspace, int64_t workspace_bytes, hipStream_t st) {
  if (workspace_bytes < ngm_knn_workspace_bytes(num_fields, P, K) || !workspace) return NGM_E_WORKSPACE;
  if (num_fields < 1 || P * K > 0x7fffffff) return NGM_E_UNSUPPORTED;
  KnnArgs a;
  knn_common(a, fc, pr, num_fields, pos, quat, K, distance_factor, outside_value, mask_radius);
  a.P = P; a.points = points; a.out = out;
  char* end;
  knn_carve(a, workspace, &end);
  const int le = knn_stages(a, true, st);
  if (le) return le;
  const int pb = (int)std::min<int64_t>((P + 255) / 256, 4096);
  hipLaunchKernelGGL(k_knn_blend, dim3(std::max(pb, 1)), dim3(256), 0, st, a);
  return 0;
}

// render_image's block loop (rm.py:402-437) as one call: per block of `ray_block` rays the samples are drawn inside the
// assignment, evaluated per field and blended inside the quadrature -- between the stages only the pair records, the sample
// distances and the ray directions touch memory; the grid over the centres is built once.  Block b draws from the Philox
// stream (seed + b * ray_block, ray index within the block), as a caller looping over the blocks with the staged entry
// points (ngm_sample_rays_world -> ngm_field_eval_knn -> ngm_composite_fwd_packed) would.
int ngm_launch_render_eval_knn(const ngm_field_cfg* fc, const ngm_render_cfg* rc, const ngm_params* pr, int num_fields,
                               const float* pos, const float* quat, const ngm_rays* rays, int K, float distance_factor,
                               float outside_value, float mask_radius, int ray_block, const ngm_prediction* pred,
                               void* workspace, int64_t workspace_bytes, hipStream_t st) {
  const int S = rc->num_samples_coarse;
  const int64_t total = (int64_t)rays->F * rays->R;
  if (ray_block < 1 || S < 1 || S > CQ_MAXS_EVAL || (int64_t)ray_block * S * K > 0x7fffffff || num_fields < 1) return NGM_E_UNSUPPORTED;
  if (workspace_bytes < ngm_knn_render_workspace_bytes(num_fields, ray_block, S, K) || !workspace) return NGM_E_WORKSPACE;
  KnnArgs a;
  knn_common(a, fc, pr, num_fields, pos, quat, K, distance_factor, outside_value, mask_radius);
  a.P = (int64_t)ray_block * S;
  char* w;
  knn_carve(a, workspace, &w);
  a.dist = reinterpret_cast<float*>(w); w += (4 * a.P + 255) / 256 * 256;
  a.ray_dir = reinterpret_cast<float*>(w); w += (12 * (int64_t)ray_block + 255) / 256 * 256;
  a.ray_tab = reinterpret_cast<float4*>(w);
  a.gen = 1; a.S = S; a.rc = *rc;
  for (int64_t r0 = 0; r0 < total; r0 += ray_block) {
    const int nr = (int)std::min<int64_t>(ray_block, total - r0);
    ngm_rays rb = *rays;
    rb.F = 1; rb.R = nr; rb.gt = nullptr; rb.u_guided = nullptr;
    rb.ijs = rays->ijs + 2 * r0;
    if (rays->c2w_per_ray) rb.c2ws = rays->c2ws + 16 * r0;
    if (rays->near) rb.near = rays->near + r0;
    if (rays->far) rb.far = rays->far + r0;
    if (rays->u_coarse) rb.u_coarse = rays->u_coarse + r0 * S;
    rb.philox_seed = rays->philox_seed + (uint64_t)r0;
    a.rays = rb;
    a.P = (int64_t)nr * S;
    hipLaunchKernelGGL(k_knn_raytab, dim3((nr + 255) / 256), dim3(256), 0, st, a);
    const int le = knn_stages(a, r0 == 0, st);
    if (le) return le;
    CompositeArgs c;
    memset(&c, 0, sizeof(c));
    c.rc = *rc; c.N = nr; c.S = S; c.dists = a.dist; c.ray_dir = a.ray_dir;
    c.pair_field = a.pair_field; c.pair_w = a.pair_w; c.pair_out = a.pair_out; c.pair_K = K; c.outside_value = outside_value;
    c.rgbd = pred->rgbds ? pred->rgbds + 4 * r0 : nullptr;
    c.Cv = pred->color_vars ? pred->color_vars + 3 * r0 : nullptr;
    c.Dv = pred->depth_vars ? pred->depth_vars + r0 : nullptr;
    c.term = pred->term_probs ? pred->term_probs + r0 : nullptr;
    const int ce = ngm_launch_composite_fwd(c, st);
    if (ce) return ce;
  }
  return 0;
}
