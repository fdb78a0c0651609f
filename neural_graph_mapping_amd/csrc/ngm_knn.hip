// Eval path: NeuralFieldSet.forward(use_vmap=False) (models.py:347-405) on gfx950.
//
// The reference finds the K nearest field centres of every query point, evaluates each touched field
// in a Python loop over boolean masks and blends with softmax(-distance_factor * dist).  Here:
//   k_knn_assign  : exact K-nearest (K <= 4, all centres in LDS, per-wave candidate list), radius test, softmax weights;
//                   per-workgroup LDS histogram -> one global atomic per field per workgroup
//   k_knn_offsets : exclusive scans over fields (segment offsets, tile offsets)
//   k_knn_scatter : (point,k) pairs bucketed by field (counting sort)
//   k_knn_eval    : one workgroup per 4096-pair tile of ONE field: that field's weights resident in
//                   LDS, MLP on the matrix cores (same eval_64 as the train kernels)
//   k_knn_blend   : out[p] = sum_k w_k out_{p,k}, or outside_value on all channels
#include "ngm_field.h"
#include "ngm_launch.h"

#include <algorithm>

#define WAVE_SYNC()                                        \
  do {                                                     \
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); \
    __builtin_amdgcn_wave_barrier();                       \
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); \
  } while (0)

#define KNN_TILE 4096
#define KNN_MAXK 4

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
  int* counts;          // (NF)   pairs per field
  int* cursor;          // (NF)
  int* seg_off;         // (NF+1)
  int* tile_off;        // (NF+1)
  int* sorted;          // (P*K)  pair ids grouped by field
  float4* pair_out;     // (P*K)
};

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

// K nearest centres, exact, with a per-wave candidate list.  The 64 points of a wave are consecutive samples of a
// ray (or neighbours of a dense grid): they sit in a ball (c0, rho).  If D bounds the K-th nearest distance of c0
// from above, the K nearest centres of EVERY point of the wave lie within D + 2 rho of c0 (triangle inequality), so
// only those centres -- typically a handful of a few hundred -- are compared per point.  D = the K-th smallest of the
// 64 lanes' minima over disjoint subsets of the centres (K distinct centres, hence an upper bound).  The list keeps
// the field order, so distances, tie-breaking and results are bit-identical to the brute-force loop; incoherent
// points only make the list long.
__global__ __launch_bounds__(256) void k_knn_assign(KnnArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smf[];
  float4* cpos = reinterpret_cast<float4*>(smf);       // NF centres, one 16-byte broadcast read each
  int* hist = reinterpret_cast<int*>(smf + 4 * a.NF);  // NF
  int* cand_all = hist + a.NF;                         // 4 waves x NF candidate indices
  for (int i = threadIdx.x; i < a.NF; i += blockDim.x) cpos[i] = make_float4(a.pos[3 * i], a.pos[3 * i + 1], a.pos[3 * i + 2], 0.f);
  for (int i = threadIdx.x; i < a.NF; i += blockDim.x) hist[i] = 0;
  __syncthreads();
  const int K = a.K;
  const int lane = threadIdx.x & 63;
  int* cand = cand_all + (threadIdx.x >> 6) * a.NF;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t p0 = (int64_t)blockIdx.x * blockDim.x; p0 < a.P; p0 += stride) {   // uniform trip count per wave
    const int64_t p = p0 + threadIdx.x;
    const bool live = p < a.P;
    float x = 0.f, y = 0.f, z = 0.f;
    if (live) { x = a.points[3 * p]; y = a.points[3 * p + 1]; z = a.points[3 * p + 2]; }
    // ---- bounding ball of the wave's points
    const float big = 3.0e38f;
    const float lx = wave_min_f(live ? x : big), hx = wave_max_f(live ? x : -big);
    const float ly = wave_min_f(live ? y : big), hy = wave_max_f(live ? y : -big);
    const float lz = wave_min_f(live ? z : big), hz = wave_max_f(live ? z : -big);
    int ncand = 0;
    if (lx <= hx) {                                            // at least one live lane (wave-uniform)
      const float cx = 0.5f * (lx + hx), cy = 0.5f * (ly + hy), cz = 0.5f * (lz + hz);
      const float ex = x - cx, ey = y - cy, ez = z - cz;
      const float rho = sqrtf(wave_max_f(live ? ex * ex + ey * ey + ez * ez : 0.f)) * 1.0001f + 1e-12f;
      // ---- upper bound D of the K-th nearest distance of c0
      float dmin = INFINITY;
      for (int f = lane; f < a.NF; f += 64) {
        const float4 c = cpos[f];
        const float dx = cx - c.x, dy = cy - c.y, dz = cz - c.z;
        dmin = fminf(dmin, dx * dx + dy * dy + dz * dz);
      }
      float D2 = INFINITY;
      for (int k = 0; k < K; ++k) {
        D2 = wave_min_f(dmin);
        const unsigned long long who = __ballot(dmin == D2);
        if (who && lane == __ffsll((long long)who) - 1) dmin = INFINITY;   // pop one lane's minimum per round
      }
      // fewer than K centres: D2 is inf and every centre stays a candidate.  Slack: the bound is compared in squared
      // distances computed in fp32 -> widen by a relative 1e-4 (costs nothing, keeps the list a superset).
      const float reach = (sqrtf(D2) + 2.0f * rho) * 1.0001f;
      const float reach2 = reach * reach;
      // ---- candidate list in field order
      for (int f0 = 0; f0 < a.NF; f0 += 64) {
        const int f = f0 + lane;
        bool keep = false;
        if (f < a.NF) {
          const float4 c = cpos[f];
          const float dx = cx - c.x, dy = cy - c.y, dz = cz - c.z;
          keep = !(dx * dx + dy * dy + dz * dz > reach2);      // NaN-safe: keeps the centre
        }
        const unsigned long long m = __ballot(keep);
        if (keep) cand[ncand + __popcll(m & ((1ull << lane) - 1ull))] = f;
        ncand += __popcll(m);
      }
    }
    WAVE_SYNC();
    if (live) {
    float bd[KNN_MAXK]; int bi[KNN_MAXK];
#pragma unroll
    for (int k = 0; k < KNN_MAXK; ++k) { bd[k] = INFINITY; bi[k] = -1; }
    float worst = INFINITY;                                   // bd[K-1]: most centres are farther and skip the insertion
    for (int ci = 0; ci < ncand; ++ci) {
      const int f = cand[ci];
      const float4 c = cpos[f];
      const float dx = x - c.x, dy = y - c.y, dz = z - c.z;
      float d = dx * dx + dy * dy + dz * dz;
      if (d < worst) {
        int id = f;
        // sorted insertion (stable: equal distances keep the lower field index first)
#pragma unroll
        for (int k = 0; k < KNN_MAXK; ++k) {
          if (k < K && d < bd[k]) { const float td = bd[k]; const int ti = bi[k]; bd[k] = d; bi[k] = id; d = td; id = ti; }
        }
        worst = (K == 1) ? bd[0] : (K == 2) ? bd[1] : (K == 3) ? bd[2] : bd[3];
      }
    }
    float dist[KNN_MAXK];
#pragma unroll
    for (int k = 0; k < KNN_MAXK; ++k) dist[k] = sqrtf(bd[k]);
    const bool inside = dist[0] < a.radius;                      // models.py:369
    // softmax(-distance_factor * dist) over the K neighbours (models.py:384)
    float mx = -INFINITY, e[KNN_MAXK], sum = 0.f;
#pragma unroll
    for (int k = 0; k < KNN_MAXK; ++k) if (k < K) mx = fmaxf(mx, -a.distance_factor * dist[k]);
#pragma unroll
    for (int k = 0; k < KNN_MAXK; ++k) { e[k] = (k < K) ? expf(-a.distance_factor * dist[k] - mx) : 0.f; sum += e[k]; }
#pragma unroll
    for (int k = 0; k < KNN_MAXK; ++k) {
      if (k < K) {
        a.pair_field[p * K + k] = inside ? bi[k] : -1;
        a.pair_w[p * K + k] = e[k] / sum;
        if (inside) atomicAdd(&hist[bi[k]], 1);
      }
    }
    }
    WAVE_SYNC();     // the next round overwrites the candidate list
  }
  __syncthreads();
  for (int i = threadIdx.x; i < a.NF; i += blockDim.x)
    if (hist[i]) atomicAdd(&a.counts[i], hist[i]);
}

// exclusive scans over the fields (segment offsets, tile offsets): one wave, 64 fields per round
__global__ void k_knn_offsets(KnnArgs a) {
  const int lane = threadIdx.x;          // launched with 64 threads
  int so = 0, to = 0;
  for (int f0 = 0; f0 < a.NF; f0 += 64) {
    const int f = f0 + lane;
    const int c = (f < a.NF) ? a.counts[f] : 0;
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
#define SC_ITEMS 8
__global__ __launch_bounds__(256) void k_knn_scatter(KnnArgs a) {
  extern __shared__ int sc_lds[];
  int* hist = sc_lds;            // NF: pairs of this workgroup per field, then the reserved global base
  const int64_t n = a.P * a.K;
  for (int i = threadIdx.x; i < a.NF; i += blockDim.x) hist[i] = 0;
  __syncthreads();
  const int64_t base = (int64_t)blockIdx.x * (SC_ITEMS * 256);
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

template <int MI, int MH, int L, bool NEED_COS, int HASH, int SKIP, bool B3 = false>
__global__ __launch_bounds__(NGM_BLOCK) void k_knn_eval(KnnArgs a) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int total_tiles = a.tile_off[a.NF];
  if ((int)blockIdx.x >= total_tiles) return;
  // field of this tile: last f with tile_off[f] <= blockIdx.x
  int lo = 0, hi = a.NF - 1;
  while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (a.tile_off[mid] <= (int)blockIdx.x) lo = mid; else hi = mid - 1; }
  const int f = lo;
  const int tile = blockIdx.x - a.tile_off[f];
  const int beg = a.seg_off[f] + tile * KNN_TILE, end = min(a.seg_off[f + 1], beg + KNN_TILE);
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
  const float px = a.pos[3 * f], py = a.pos[3 * f + 1], pz = a.pos[3 * f + 2];
  const float qw = a.quat[4 * f], qx = a.quat[4 * f + 1], qy = a.quat[4 * f + 2], qz = a.quat[4 * f + 3];
  const HashCtx hc = make_hash_ctx(a.fc, a.pr, row, nullptr);
  const TriCtx tc = make_tri_ctx(a.fc, a.pr, row, nullptr);
  for (int base = beg + wave * 64; base < end; base += NGM_BLOCK) {
    const int idx = base + lane;
    const bool valid = idx < end;
    float x = 0, y = 0, z = 0;
    int pair = 0;
    if (valid) {
      pair = a.sorted[idx];
      const int64_t p = pair / a.K;
      Vec3 v{a.points[3 * p] - px, a.points[3 * p + 1] - py, a.points[3 * p + 2] - pz};   // models.py:377-381
      v = quat_rotate_inv(qw, qx, qy, qz, v);
      x = v.x / div + off; y = v.y / div + off; z = v.z / div + off;
    }
    const float4 o = eval_64<MI, MH, L, NEED_COS, HASH, SKIP, B3>(sm, lane, x, y, z, &hc, nullptr, nullptr, b3w, &tc);
    if (valid) a.pair_out[pair] = o;
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

int64_t ngm_knn_workspace_bytes(int num_fields, int64_t P, int K) {
  const int64_t n = P * K;
  return 256 * 8 + 4 * (n + 255) + 4 * (n + 255) + 4 * (int64_t)(4 * num_fields + 64) + 4 * (n + 255) + 16 * (n + 16);
}

template <int MI, int MH, int L>
static int launch_eval(const KnnArgs& a, int grid, hipStream_t st) {
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
      (void)hipFuncSetAttribute((const void*)k_knn_eval<MI, MH, L, false, 0, 0, true>,
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      hipLaunchKernelGGL((k_knn_eval<MI, MH, L, false, 0, 0, true>), dim3(grid), dim3(NGM_BLOCK), lds, st, a);
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

int ngm_launch_knn(const ngm_field_cfg* fc, const ngm_params* pr, int num_fields, int64_t P, const float* points,
                   const float* pos, const float* quat, int K, float distance_factor, float outside_value, float mask_radius,
                   float* out, void* workspace, int64_t workspace_bytes, hipStream_t st) {
  if (workspace_bytes < ngm_knn_workspace_bytes(num_fields, P, K) || !workspace) return NGM_E_WORKSPACE;
  // k_knn_assign keeps centres + histogram + 4 candidate lists of every field in LDS: 36 B per field of the 160 KiB
  if (num_fields > 4096 || P * K > 0x7fffffff) return NGM_E_UNSUPPORTED;
  KnnArgs a;
  a.fc = *fc; a.pr = *pr; a.NF = num_fields; a.K = K; a.P = P; a.points = points; a.pos = pos; a.quat = quat;
  a.distance_factor = distance_factor; a.outside_value = outside_value; a.radius = mask_radius; a.out = out;
  const int64_t n = P * K;
  char* w = reinterpret_cast<char*>(((int64_t)workspace + 255) / 256 * 256);
  auto carve = [&](int64_t bytes) { char* p = w; w += (bytes + 255) / 256 * 256; return p; };
  a.pair_field = reinterpret_cast<int*>(carve(4 * n));
  a.pair_w = reinterpret_cast<float*>(carve(4 * n));
  a.counts = reinterpret_cast<int*>(carve(4 * num_fields));
  a.cursor = reinterpret_cast<int*>(carve(4 * num_fields));
  a.seg_off = reinterpret_cast<int*>(carve(4 * (num_fields + 1)));
  a.tile_off = reinterpret_cast<int*>(carve(4 * (num_fields + 1)));
  a.sorted = reinterpret_cast<int*>(carve(4 * n));
  a.pair_out = reinterpret_cast<float4*>(carve(16 * n));
  (void)hipMemsetAsync(a.counts, 0, 4 * (size_t)num_fields, st);
  const int pb = (int)std::min<int64_t>((P + 255) / 256, 4096);
  const size_t lds_a = (size_t)num_fields * (20 + 4 * 4);    // centres, histogram, 4 waves of candidate lists
  if (lds_a > 48 * 1024) {
    static const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void*>(k_knn_assign),
                                                       hipFuncAttributeMaxDynamicSharedMemorySize, 4096 * 36);
    if (attr != hipSuccess) return NGM_E_HIP;
  }
  {
    NgmProfScope prof_(NGM_K_KNN_ASSIGN, st);
    hipLaunchKernelGGL(k_knn_assign, dim3(std::max(pb, 1)), dim3(256), lds_a, st, a);
  }
  if (hipGetLastError() != hipSuccess) return NGM_E_HIP;      // an over-sized LDS request fails here, not four launches later
  hipLaunchKernelGGL(k_knn_offsets, dim3(1), dim3(64), 0, st, a);
  const int nb = (int)((n + SC_ITEMS * 256 - 1) / (SC_ITEMS * 256));
  hipLaunchKernelGGL(k_knn_scatter, dim3(std::max(nb, 1)), dim3(256), (size_t)num_fields * 4, st, a);
  const int max_tiles = (int)((n + KNN_TILE - 1) / KNN_TILE) + num_fields;
  const FieldShape s = field_shape(fc);
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
  if (le) return le;
  hipLaunchKernelGGL(k_knn_blend, dim3(std::max(pb, 1)), dim3(256), 0, st, a);
  return 0;
}
