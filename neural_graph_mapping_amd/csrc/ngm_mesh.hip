// Mesh extraction on the device: marching cubes over a dense grid of field values, replacing the
// pytorch3d.ops.marching_cubes call of NeuralGraphMap._extract_mesh (run_mapping.py:2255-2298).
//
// HBM-bound integer/byte work: three streaming passes over the (nx, ny, nz) volume, no atomics, deterministic
// output order (vertices by (grid point, axis) of the crossed edge, faces by cell) through two exclusive scans:
//   k_mc_classify : per grid point the three "edge crossed" flags + per cell the triangle count of its corner case
//   k_scan_*      : vertex / face offsets, hand-written reduce-then-scan (the +1'th element holds the totals)
//   k_mc_emit     : vertices (linear interpolation along the edge) and faces (indices through the vertex offsets)
// The 256-case triangulation table is derived from the cube topology at first use (no typed-in table): face
// segments -> closed loops -> a triangulation without in-face diagonals; ambiguous faces cut off the inside corners,
// a rule both neighbours of a face evaluate identically, so the mesh is watertight.  PARITY UNPINNED against
// pytorch3d (not vendored); the CPU restatement is oracle/mesh_oracle.py.
#include <hip/hip_runtime.h>

#include <array>
#include <cstring>
#include <vector>

#include "ngm_launch.h"

namespace {

constexpr int MC_MAX_TRI = 5;
__device__ int8_t d_tri_table[256 * MC_MAX_TRI * 3];
__device__ int32_t d_tri_count[256];

struct McTables {
  int8_t table[256 * MC_MAX_TRI * 3];
  int32_t count[256];
  bool ok;
};

// ---- host: derive the table ---------------------------------------------------------------------------------
struct Cube {
  int corner[8][3];
  int edge[12][2];       // 4 edges along x, then y, then z, each ordered by the lower corner
  int edge_id[8][8];
  int face[6][4];        // corner cycles
  int edge_faces[12];    // bit mask of the (two) faces an edge lies on
  Cube() {
    for (int i = 0; i < 8; ++i) { corner[i][0] = i & 1; corner[i][1] = (i >> 1) & 1; corner[i][2] = (i >> 2) & 1; }
    memset(edge_id, -1, sizeof(edge_id));
    int n = 0;
    for (int ax = 0; ax < 3; ++ax)
      for (int a = 0; a < 8; ++a)
        if (!(a & (1 << ax))) { edge[n][0] = a; edge[n][1] = a | (1 << ax); edge_id[a][a | (1 << ax)] = edge_id[a | (1 << ax)][a] = n; ++n; }
    n = 0;
    for (int ax = 0; ax < 3; ++ax) {
      int o[2], k = 0;
      for (int b = 0; b < 3; ++b) if (b != ax) o[k++] = 1 << b;
      for (int side = 0; side < 2; ++side) {
        const int base = side << ax;
        face[n][0] = base; face[n][1] = base | o[0]; face[n][2] = base | o[0] | o[1]; face[n][3] = base | o[1];
        ++n;
      }
    }
    for (int e = 0; e < 12; ++e) {
      edge_faces[e] = 0;
      for (int f = 0; f < 6; ++f) {
        int hit = 0;
        for (int k = 0; k < 4; ++k) hit += (face[f][k] == edge[e][0]) + (face[f][k] == edge[e][1]);
        if (hit == 2) edge_faces[e] |= 1 << f;
      }
    }
  }
};

using Tri = std::array<int, 3>;
using Triangulation = std::vector<Tri>;

// every triangulation of the ordered polygon, fixed enumeration order: the triangle on the polygon edge
// (poly[0], poly[last]) picks its apex k, then the two sub-polygons recurse
static std::vector<Triangulation> all_triangulations(const std::vector<int>& poly) {
  std::vector<Triangulation> res;
  if (poly.size() < 3) { res.push_back({}); return res; }
  if (poly.size() == 3) { res.push_back({Tri{poly[0], poly[1], poly[2]}}); return res; }
  for (size_t k = 1; k + 1 < poly.size(); ++k) {
    const std::vector<int> lp(poly.begin(), poly.begin() + k + 1), rp(poly.begin() + k, poly.end());
    for (const auto& l : all_triangulations(lp))
      for (const auto& r : all_triangulations(rp)) {
        Triangulation t = l;
        t.push_back(Tri{poly[0], poly[k], poly.back()});
        t.insert(t.end(), r.begin(), r.end());
        res.push_back(t);
      }
  }
  return res;
}

static McTables derive_tables() {
  McTables T;
  memset(T.table, -1, sizeof(T.table));
  memset(T.count, 0, sizeof(T.count));
  T.ok = true;
  const Cube C;
  for (int cs = 0; cs < 256; ++cs) {
    int inside[8];
    for (int i = 0; i < 8; ++i) inside[i] = (cs >> i) & 1;
    int nbr[12][2], deg[12];
    memset(deg, 0, sizeof(deg));
    auto link = [&](int a, int b) { nbr[a][deg[a]++] = b; nbr[b][deg[b]++] = a; };
    for (int f = 0; f < 6; ++f) {
      int cross[4], nc = 0;
      for (int k = 0; k < 4; ++k)
        if (inside[C.face[f][k]] != inside[C.face[f][(k + 1) & 3]]) cross[nc++] = C.edge_id[C.face[f][k]][C.face[f][(k + 1) & 3]];
      if (nc == 2) link(cross[0], cross[1]);
      else if (nc == 4)
        for (int k = 0; k < 4; ++k)      // ambiguous face: every inside corner is cut off on its own
          if (inside[C.face[f][k]]) link(C.edge_id[C.face[f][k]][C.face[f][(k + 1) & 3]], C.edge_id[C.face[f][k]][C.face[f][(k + 3) & 3]]);
    }
    bool seen[12] = {};
    int ntri = 0;
    for (int start = 0; start < 12; ++start) {
      if (!deg[start] || seen[start]) continue;
      std::vector<int> loop{start};
      seen[start] = true;
      for (int prev = -1, cur = start;;) {
        const int nxt = (nbr[cur][0] == prev) ? nbr[cur][1] : nbr[cur][0];
        if (nxt == start) break;
        loop.push_back(nxt);
        seen[nxt] = true;
        prev = cur; cur = nxt;
      }
      // orientation: Newell normal of the loop (doubled edge mid points, integers) against inside -> outside
      const int n = (int)loop.size();
      long nrm[3] = {0, 0, 0}, d[3] = {0, 0, 0};
      for (int k = 0; k < n; ++k) {
        int p[3], q[3];
        for (int c = 0; c < 3; ++c) {
          p[c] = C.corner[C.edge[loop[k]][0]][c] + C.corner[C.edge[loop[k]][1]][c];
          q[c] = C.corner[C.edge[loop[(k + 1) % n]][0]][c] + C.corner[C.edge[loop[(k + 1) % n]][1]][c];
        }
        nrm[0] += p[1] * q[2] - p[2] * q[1]; nrm[1] += p[2] * q[0] - p[0] * q[2]; nrm[2] += p[0] * q[1] - p[1] * q[0];
        const int a = C.edge[loop[k]][0], b = C.edge[loop[k]][1];
        for (int c = 0; c < 3; ++c) d[c] += (C.corner[b][c] - C.corner[a][c]) * (inside[a] ? 1 : -1);
      }
      if (nrm[0] * d[0] + nrm[1] * d[1] + nrm[2] * d[2] < 0) {
        std::vector<int> r{loop[0]};
        for (int k = n - 1; k >= 1; --k) r.push_back(loop[k]);
        loop = r;
      }
      // first triangulation none of whose interior diagonals lies in a cube face
      bool found = false;
      for (const auto& tri : all_triangulations(loop)) {
        bool ok = true;
        for (const auto& t : tri)
          for (int j = 0; j < 3 && ok; ++j) {
            const int a = t[j], b = t[(j + 1) % 3];
            bool adjacent = false;
            for (int k = 0; k < n; ++k)
              adjacent |= (loop[k] == a && loop[(k + 1) % n] == b) || (loop[k] == b && loop[(k + 1) % n] == a);
            if (!adjacent && (C.edge_faces[a] & C.edge_faces[b])) ok = false;
          }
        if (!ok) continue;
        for (const auto& t : tri) {
          if (ntri >= MC_MAX_TRI) { T.ok = false; break; }
          for (int j = 0; j < 3; ++j) T.table[(cs * MC_MAX_TRI + ntri) * 3 + j] = (int8_t)t[j];
          ++ntri;
        }
        found = true;
        break;
      }
      if (!found) T.ok = false;
    }
    T.count[cs] = ntri;
  }
  return T;
}

static const McTables& host_tables() {
  static const McTables T = derive_tables();
  return T;
}

static int upload_tables() {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return NGM_E_HIP;
  static bool done[64] = {};
  if (dev < 0 || dev >= 64) return NGM_E_UNSUPPORTED;
  if (done[dev]) return NGM_OK;
  const McTables& T = host_tables();
  if (!T.ok) return NGM_E_UNSUPPORTED;
  if (hipMemcpyToSymbol(HIP_SYMBOL(d_tri_table), T.table, sizeof(T.table)) != hipSuccess) return NGM_E_HIP;
  if (hipMemcpyToSymbol(HIP_SYMBOL(d_tri_count), T.count, sizeof(T.count)) != hipSuccess) return NGM_E_HIP;
  done[dev] = true;
  return NGM_OK;
}

// ---- device -------------------------------------------------------------------------------------------------
struct McArgs {
  const float* vol;
  int nx, ny, nz;
  float iso;
  int32_t* vflag;   // (3 * N + 1): edge (grid point p, axis a) crossed; after the scan: vertex offset; [3N] = total
  int32_t* tcnt;    // (cells + 1): triangles of the cell; after the scan: face offset; [cells] = total
  float* verts;     // (V, 3) grid-index coordinates
  int64_t* faces;   // (T, 3)
  int64_t max_verts, max_faces;
  int64_t* counts;  // device [2]
};

__global__ void __launch_bounds__(256) k_mc_classify(McArgs a) {
  const int64_t N = (int64_t)a.nx * a.ny * a.nz;
  const int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (p >= N) {
    if (p == N) { a.vflag[3 * N] = 0; a.tcnt[(int64_t)(a.nx - 1) * (a.ny - 1) * (a.nz - 1)] = 0; }
    return;
  }
  const int z = (int)(p % a.nz), y = (int)((p / a.nz) % a.ny), x = (int)(p / ((int64_t)a.nz * a.ny));
  const int64_t sx = (int64_t)a.ny * a.nz, sy = a.nz;
  const bool in0 = a.vol[p] > a.iso;
  const bool hx = x + 1 < a.nx, hy = y + 1 < a.ny, hz = z + 1 < a.nz;
  const bool inx = hx && a.vol[p + sx] > a.iso, iny = hy && a.vol[p + sy] > a.iso, inz = hz && a.vol[p + 1] > a.iso;
  a.vflag[3 * p + 0] = hx && (inx != in0);
  a.vflag[3 * p + 1] = hy && (iny != in0);
  a.vflag[3 * p + 2] = hz && (inz != in0);
  if (hx && hy && hz) {
    int cs = (int)in0 | ((int)inx << 1) | ((int)iny << 2) | ((int)inz << 4);
    cs |= (int)(a.vol[p + sx + sy] > a.iso) << 3;
    cs |= (int)(a.vol[p + sx + 1] > a.iso) << 5;
    cs |= (int)(a.vol[p + sy + 1] > a.iso) << 6;
    cs |= (int)(a.vol[p + sx + sy + 1] > a.iso) << 7;
    a.tcnt[((int64_t)x * (a.ny - 1) + y) * (a.nz - 1) + z] = d_tri_count[cs];
  }
}

__global__ void k_mc_totals(McArgs a) {
  const int64_t N = (int64_t)a.nx * a.ny * a.nz;
  a.counts[0] = a.vflag[3 * N];
  a.counts[1] = a.tcnt[(int64_t)(a.nx - 1) * (a.ny - 1) * (a.nz - 1)];
}

__global__ void __launch_bounds__(256) k_mc_emit(McArgs a) {
  const int64_t N = (int64_t)a.nx * a.ny * a.nz;
  const int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (p >= N) return;
  const int z = (int)(p % a.nz), y = (int)((p / a.nz) % a.ny), x = (int)(p / ((int64_t)a.nz * a.ny));
  const int64_t sx = (int64_t)a.ny * a.nz, sy = a.nz;
  const int64_t stride[3] = {sx, sy, 1};
  const float v0 = a.vol[p];
  const bool in0 = v0 > a.iso;
  const float base[3] = {(float)x, (float)y, (float)z};
  const int lim[3] = {a.nx, a.ny, a.nz}, at[3] = {x, y, z};
#pragma unroll
  for (int ax = 0; ax < 3; ++ax) {
    if (at[ax] + 1 >= lim[ax]) continue;
    const float v1 = a.vol[p + stride[ax]];
    if ((v1 > a.iso) == in0) continue;
    const int64_t vi = a.vflag[3 * p + ax];
    if (vi >= a.max_verts) continue;
    const float t = (a.iso - v0) / (v1 - v0);
    float o[3] = {base[0], base[1], base[2]};
    o[ax] = base[ax] + t;
    a.verts[3 * vi + 0] = o[0]; a.verts[3 * vi + 1] = o[1]; a.verts[3 * vi + 2] = o[2];
  }
  if (x + 1 < a.nx && y + 1 < a.ny && z + 1 < a.nz) {
    const int64_t cell = ((int64_t)x * (a.ny - 1) + y) * (a.nz - 1) + z;
    const int64_t t0 = a.tcnt[cell];
    const int n = (int)(a.tcnt[cell + 1] - t0);
    if (n == 0) return;
    int cs = (int)in0;
#pragma unroll
    for (int i = 1; i < 8; ++i)
      cs |= (int)(a.vol[p + (i & 1) * sx + ((i >> 1) & 1) * sy + ((i >> 2) & 1)] > a.iso) << i;
    for (int k = 0; k < n; ++k) {
      if (t0 + k >= a.max_faces) break;
      for (int j = 0; j < 3; ++j) {
        const int e = d_tri_table[(cs * MC_MAX_TRI + k) * 3 + j];
        const int ax = e >> 2;                       // 4 edges per axis, ordered by the lower corner
        // lower corner of edge e: the (e & 3)-th corner whose bit `ax` is clear
        const int lo = e & 3;
        const int b0 = (ax == 0) ? 1 : 0, b1 = (ax == 2) ? 1 : 2;      // the two other axes, ascending
        const int corner = ((lo & 1) << b0) | ((lo >> 1) << b1);
        const int64_t q = p + (corner & 1) * sx + ((corner >> 1) & 1) * sy + ((corner >> 2) & 1);
        a.faces[3 * (t0 + k) + j] = a.vflag[3 * q + ax];
      }
    }
  }
}

}  // namespace

// ---- exclusive prefix sum of int32, in place (vertex / face offsets).  Hand-written reduce-then-scan, three launches:
//   k_scan_sums    one workgroup per 4096-element chunk: the chunk's sum
//   k_scan_chunks  ONE workgroup: exclusive scan of the chunk sums (a 201^3 block has 5 950 of them)
//   k_scan_apply   one workgroup per chunk: 16 consecutive elements per thread (four 16-byte loads), thread-local scan, wave scan
//                  by lane shifts, the four waves' totals through LDS, + the chunk's offset; written back over the input
// 12 bytes of traffic per element (read twice, written once) -- HBM-bound integer work, no atomics, fixed order.
constexpr int SCAN_T = 256, SCAN_V = 16, SCAN_CHUNK = SCAN_T * SCAN_V;

__device__ __forceinline__ int4 scan_load4(const int32_t* __restrict__ d, int64_t i, int64_t n) {
  if (i + 3 < n) return *reinterpret_cast<const int4*>(d + i);
  int4 v = make_int4(0, 0, 0, 0);
  if (i < n) v.x = d[i];
  if (i + 1 < n) v.y = d[i + 1];
  if (i + 2 < n) v.z = d[i + 2];
  return v;
}
__device__ __forceinline__ int block_sum_256(int v, int* sh) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
  __syncthreads();
  return sh[0] + sh[1] + sh[2] + sh[3];
}
__global__ __launch_bounds__(SCAN_T) void k_scan_sums(const int32_t* __restrict__ d, int64_t n, int32_t* __restrict__ csum) {
  __shared__ int sh[4];
  const int64_t base = (int64_t)blockIdx.x * SCAN_CHUNK;
  int s = 0;
#pragma unroll
  for (int j = 0; j < SCAN_V / 4; ++j) {              // striped: a wave reads 1 KB contiguous per instruction
    const int4 v = scan_load4(d, base + ((int64_t)j * SCAN_T + threadIdx.x) * 4, n);
    s += v.x + v.y + v.z + v.w;
  }
  const int t = block_sum_256(s, sh);
  if (threadIdx.x == 0) csum[blockIdx.x] = t;
}
__global__ __launch_bounds__(1024) void k_scan_chunks(int32_t* __restrict__ csum, int nc) {
  __shared__ int wsum[16];
  __shared__ int carry_s;
  if (threadIdx.x == 0) carry_s = 0;
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int b0 = 0; b0 < nc; b0 += 1024) {
    const int i = b0 + threadIdx.x;
    const int v = i < nc ? csum[i] : 0;
    int inc = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const int u = __shfl_up(inc, o, 64); if (lane >= o) inc += u; }
    if (lane == 63) wsum[wave] = inc;
    __syncthreads();
    int off = carry_s;
    for (int w = 0; w < wave; ++w) off += wsum[w];
    if (i < nc) csum[i] = off + inc - v;
    __syncthreads();
    if (threadIdx.x == 1023) carry_s = off + inc;
    __syncthreads();
  }
}
__global__ __launch_bounds__(SCAN_T) void k_scan_apply(int32_t* __restrict__ d, int64_t n, const int32_t* __restrict__ coff) {
  __shared__ int wsum[4];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t i0 = (int64_t)blockIdx.x * SCAN_CHUNK + (int64_t)threadIdx.x * SCAN_V;
  int4 v[SCAN_V / 4];
#pragma unroll
  for (int j = 0; j < SCAN_V / 4; ++j) v[j] = scan_load4(d, i0 + 4 * j, n);
  int run = 0;
#pragma unroll
  for (int j = 0; j < SCAN_V / 4; ++j) {              // thread-local exclusive scan
    const int4 x = v[j];
    v[j].x = run; run += x.x;
    v[j].y = run; run += x.y;
    v[j].z = run; run += x.z;
    v[j].w = run; run += x.w;
  }
  int inc = run;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) { const int u = __shfl_up(inc, o, 64); if (lane >= o) inc += u; }
  if (lane == 63) wsum[wave] = inc;
  __syncthreads();
  int off = coff[blockIdx.x] + inc - run;
  for (int w = 0; w < wave; ++w) off += wsum[w];
#pragma unroll
  for (int j = 0; j < SCAN_V / 4; ++j) {
    const int64_t i = i0 + 4 * j;
    const int4 o4 = make_int4(v[j].x + off, v[j].y + off, v[j].z + off, v[j].w + off);
    if (i + 3 < n) *reinterpret_cast<int4*>(d + i) = o4;
    else {
      if (i < n) d[i] = o4.x;
      if (i + 1 < n) d[i + 1] = o4.y;
      if (i + 2 < n) d[i + 2] = o4.z;
    }
  }
}
static void exclusive_scan_inplace(int32_t* d, int64_t n, int32_t* chunk_sums, hipStream_t st) {
  const int nc = (int)((n + SCAN_CHUNK - 1) / SCAN_CHUNK);
  hipLaunchKernelGGL(k_scan_sums, dim3(nc), dim3(SCAN_T), 0, st, d, n, chunk_sums);
  hipLaunchKernelGGL(k_scan_chunks, dim3(1), dim3(1024), 0, st, chunk_sums, nc);
  hipLaunchKernelGGL(k_scan_apply, dim3(nc), dim3(SCAN_T), 0, st, d, n, chunk_sums);
}

// workspace: vflag (3N+1) + tcnt (cells+1) int32 + the scans' chunk sums
static int64_t mc_scan_temp_bytes(int64_t n) { return 4 * ((n + SCAN_CHUNK - 1) / SCAN_CHUNK) + 256; }
static inline int64_t up256(int64_t x) { return (x + 255) / 256 * 256; }

int64_t ngm_mc_workspace_bytes(int nx, int ny, int nz) {
  const int64_t N = (int64_t)nx * ny * nz, cells = (int64_t)(nx - 1) * (ny - 1) * (nz - 1);
  return 256 + up256(4 * (3 * N + 1)) + up256(4 * (cells + 1)) + up256(mc_scan_temp_bytes(3 * N + 1));
}

static int mc_setup(McArgs& a, void** temp, int64_t* temp_bytes, const float* vol, int nx, int ny, int nz, float iso,
                    void* workspace, int64_t workspace_bytes) {
  if (!vol || nx < 2 || ny < 2 || nz < 2 || !workspace) return NGM_E_INVALID;
  const int64_t N = (int64_t)nx * ny * nz, cells = (int64_t)(nx - 1) * (ny - 1) * (nz - 1);
  if (3 * N + 1 > 0x7fffffff) return NGM_E_UNSUPPORTED;
  if (workspace_bytes < ngm_mc_workspace_bytes(nx, ny, nz)) return NGM_E_WORKSPACE;
  char* w = reinterpret_cast<char*>(up256((int64_t)workspace));
  memset(&a, 0, sizeof(a));
  a.vol = vol; a.nx = nx; a.ny = ny; a.nz = nz; a.iso = iso;
  a.vflag = reinterpret_cast<int32_t*>(w); w += up256(4 * (3 * N + 1));
  a.tcnt = reinterpret_cast<int32_t*>(w); w += up256(4 * (cells + 1));
  *temp = w; *temp_bytes = mc_scan_temp_bytes(3 * N + 1);
  return NGM_OK;
}

// pass 1: classify + scans; counts (device int64[2]) = number of vertices, number of faces.  Leaves the offsets in
// the workspace for ngm_launch_mc_emit.
int ngm_launch_mc_count(const float* vol, int nx, int ny, int nz, float iso, int64_t* counts, void* workspace,
                        int64_t workspace_bytes, hipStream_t st) {
  McArgs a; void* temp; int64_t tb;
  int e = mc_setup(a, &temp, &tb, vol, nx, ny, nz, iso, workspace, workspace_bytes);
  if (e) return e;
  if (!counts) return NGM_E_INVALID;
  e = upload_tables();
  if (e) return e;
  a.counts = counts;
  const int64_t N = (int64_t)nx * ny * nz, cells = (int64_t)(nx - 1) * (ny - 1) * (nz - 1);
  hipLaunchKernelGGL(k_mc_classify, dim3((unsigned)((N + 1 + 255) / 256)), dim3(256), 0, st, a);
  (void)tb;
  exclusive_scan_inplace(a.vflag, 3 * N + 1, reinterpret_cast<int32_t*>(temp), st);
  exclusive_scan_inplace(a.tcnt, cells + 1, reinterpret_cast<int32_t*>(temp), st);
  hipLaunchKernelGGL(k_mc_totals, dim3(1), dim3(1), 0, st, a);
  return NGM_OK;
}

// pass 2: emit (same volume / isolevel / workspace as pass 1)
int ngm_launch_mc_emit(const float* vol, int nx, int ny, int nz, float iso, float* verts, int64_t max_verts,
                       int64_t* faces, int64_t max_faces, void* workspace, int64_t workspace_bytes, hipStream_t st) {
  McArgs a; void* temp; int64_t tb;
  int e = mc_setup(a, &temp, &tb, vol, nx, ny, nz, iso, workspace, workspace_bytes);
  if (e) return e;
  if ((max_verts > 0 && !verts) || (max_faces > 0 && !faces)) return NGM_E_INVALID;
  a.verts = verts; a.faces = faces; a.max_verts = max_verts; a.max_faces = max_faces;
  const int64_t N = (int64_t)nx * ny * nz;
  hipLaunchKernelGGL(k_mc_emit, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, st, a);
  return NGM_OK;
}

// the derived table, for the tests (host data, no device needed): tri_table[256 * 15], tri_count[256]
int ngm_mc_copy_tables(int8_t* tri_table, int32_t* tri_count) {
  const McTables& T = host_tables();
  if (!T.ok) return NGM_E_UNSUPPORTED;
  if (tri_table) memcpy(tri_table, T.table, sizeof(T.table));
  if (tri_count) memcpy(tri_count, T.count, sizeof(T.count));
  return NGM_OK;
}
