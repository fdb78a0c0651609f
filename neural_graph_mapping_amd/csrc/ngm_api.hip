// C ABI of the gfx950 hot path (declared in include/ngm_hip.h): argument validation, launch planning,
// workspace carving.  No allocation, no synchronisation; everything is enqueued on the caller's stream.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../../include/ngm_hip.h"
#include "ngm_launch.h"

int ngm_launch_sampler_weighted(const ngm_render_cfg* rc, const ngm_rays* rays, int S, int B, const float* boundaries,
                                const float* weights, float* points_cam, float* distances, float* dirs, hipStream_t st);
int ngm_launch_sampler(const ngm_render_cfg* rc, const ngm_rays* rays, int S, float* points_cam, float* distances,
                       float* dirs, float* points_world, hipStream_t st);
int ngm_launch_loss_values(const ngm_render_cfg* rc, const float* sums, float* out, hipStream_t st);
int ngm_launch_loss_reduce(const float* partials, int nblocks, float* sums, uint64_t* counter, hipStream_t st);
int ngm_launch_neus_sd_grad(const float* d_isd_rays, int F, int R, const float* neus_sd, int64_t sd_stride,
                            const int64_t* field_index, float* d_sd, hipStream_t st);
int ngm_launch_read_stash(const float4* sa, const float2* sb, int64_t n, float* geoms, float* dists, hipStream_t st);
int ngm_launch_adam(float* param, float* m, float* v, int64_t stride, const float* grad, int64_t gstride,
                    const int64_t* field_index, int F, int64_t numel, int64_t step, float lr, float beta1, float beta2,
                    float eps, float wd, hipStream_t st);
int ngm_launch_adam_multi(const ngm_adam_tensor* tensors, int n, const int64_t* field_index, int F, int64_t step,
                          const int64_t* step_dev, float lr, float beta1, float beta2, float eps, float wd,
                          int64_t* advance_step, uint64_t* advance_offset, hipStream_t st);
int ngm_launch_step_advance(int64_t* step_dev, uint64_t* off_dev, hipStream_t st);
int64_t ngm_mc_workspace_bytes(int nx, int ny, int nz);
int ngm_launch_mc_count(const float* vol, int nx, int ny, int nz, float iso, int64_t* counts, void* workspace,
                        int64_t workspace_bytes, hipStream_t st);
int ngm_launch_mc_emit(const float* vol, int nx, int ny, int nz, float iso, float* verts, int64_t max_verts,
                       int64_t* faces, int64_t max_faces, void* workspace, int64_t workspace_bytes, hipStream_t st);
int ngm_mc_copy_tables(int8_t* tri_table, int32_t* tri_count);
int ngm_launch_knn(const ngm_field_cfg* fc, const ngm_params* pr, int num_fields, int64_t P, const float* points,
                   const float* pos, const float* quat, int K, float distance_factor, float outside_value, float mask_radius,
                   float* out, void* workspace, int64_t workspace_bytes, hipStream_t st);
int64_t ngm_knn_workspace_bytes(int num_fields, int64_t P, int K);
int64_t ngm_knn_render_workspace_bytes(int num_fields, int ray_block, int S, int K);
int ngm_launch_render_eval_knn(const ngm_field_cfg* fc, const ngm_render_cfg* rc, const ngm_params* pr, int num_fields,
                               const float* pos, const float* quat, const ngm_rays* rays, int K, float distance_factor,
                               float outside_value, float mask_radius, int ray_block, const ngm_prediction* pred,
                               void* workspace, int64_t workspace_bytes, hipStream_t st);

#include <mutex>
#include <vector>

static thread_local char g_err[512] = "";
static int prep_lattice_grad(const ngm_field_cfg* fc, const ngm_grads* grads, int F, FieldBwdArgs& a, hipStream_t st);

// ---- profiling hooks ----------------------------------------------------------------------------
namespace {
struct EvPair { hipEvent_t a, b; };
std::mutex g_prof_mu;
bool g_prof_on = false;
std::vector<EvPair> g_prof_pending[NGM_K_COUNT];
std::vector<EvPair> g_prof_free;
double g_prof_ms[NGM_K_COUNT] = {0};
int64_t g_prof_n[NGM_K_COUNT] = {0};
thread_local EvPair g_cur;
}  // namespace

NgmProfScope::NgmProfScope(int kernel_id, hipStream_t stream) : id(kernel_id), st(stream), on(false) {
  std::lock_guard<std::mutex> lk(g_prof_mu);
  if (!g_prof_on) return;
  on = true;
  if (g_prof_free.empty()) {
    EvPair p;
    (void)hipEventCreate(&p.a);
    (void)hipEventCreate(&p.b);
    g_cur = p;
  } else {
    g_cur = g_prof_free.back();
    g_prof_free.pop_back();
  }
  (void)hipEventRecord(g_cur.a, st);
}
NgmProfScope::~NgmProfScope() {
  if (!on) return;
  (void)hipEventRecord(g_cur.b, st);
  std::lock_guard<std::mutex> lk(g_prof_mu);
  g_prof_pending[id].push_back(g_cur);
}
static void prof_drain() {
  for (int k = 0; k < NGM_K_COUNT; ++k) {
    for (auto& p : g_prof_pending[k]) {
      float ms = 0.f;
      (void)hipEventSynchronize(p.b);
      if (hipEventElapsedTime(&ms, p.a, p.b) == hipSuccess) { g_prof_ms[k] += ms; g_prof_n[k] += 1; }
      g_prof_free.push_back(p);
    }
    g_prof_pending[k].clear();
  }
}

static int fail(int code, const char* msg) {
  snprintf(g_err, sizeof(g_err), "%s", msg);
  return code;
}
// prefer the 8-wave / 16-sample-tile backward; fall back to the 4-wave / 32-sample-tile kernel
#define NGM_FWD_DEBUG_WORDS (16 + 8 * 64)   // 16 summary slots + 8 waves x 64 timeline entries
static unsigned long long* g_debug_cycles = nullptr;
int g_ngm_last_matmul[3] = {-1, -1, -1};   // ngm_launch.h
int g_ngm_last_fwd_one_tile = 0;
static int g_last_bwd_variant = -1;   // 0: 32-sample tiles, 1: 16-sample tiles (recompute), 2: 16-sample tiles + activation stash, 3: bf16-split tiles + stash, 5: hash encoding + 1x32 MLP on the bf16 split
static int g_no_fused_comp = 0;       // ngm_debug_disable_fused_comp
static int g_last_stash_mode = -1;    // FieldBwdArgs::act_half of the last MLP backward that read an activation stash
static int g_last_comp_fused = 0;     // the last training backward did the compositing backward inside k_field_bwd_b3 (no k_stash_bwd launch)
// true when launch_bwd_any's first candidate is k_field_bwd_b3 (no experiment switch in the way)
static bool bwd_b3_is_default() {
  static const bool off = getenv("NGM_BWD32") != nullptr || getenv("NGM_NO_BWD_B3") != nullptr;
  return !off;
}
// Which targets the LAST forward on a workspace wrote its per-ray loss seeds for (host-side bookkeeping by pointer identity:
// the fused compositing backward trusts off_rayseed only when the forward that filled this workspace ran with the same
// targets; a forward without targets, or with other targets, leaves the backward on k_stash_bwd, which derives the seeds
// from targets + prediction itself).
#include <mutex>
#include <unordered_map>
static std::mutex g_seed_mu;
static std::unordered_map<const void*, const void*> g_seed_targets;       // workspace -> targets.rgbds of the forward that wrote the seeds
static std::unordered_map<const void*, int> g_ws_act_layers;              // workspace -> hidden layers the last training forward stashed (0: all)
static void note_forward_stash(const void* ws, int act_layers) {
  std::lock_guard<std::mutex> lk(g_seed_mu);
  if (g_ws_act_layers.size() >= 4096) g_ws_act_layers.clear();
  g_ws_act_layers[ws] = act_layers;
}
// -1: no record (another process wrote the workspace, or the record was dropped): trust the backward's own predicate
static int forward_stash_layers(const void* ws) {
  std::lock_guard<std::mutex> lk(g_seed_mu);
  auto it = g_ws_act_layers.find(ws);
  return it == g_ws_act_layers.end() ? -1 : it->second;
}
static void note_forward_seeds(const void* ws, const void* rgbds) {
  std::lock_guard<std::mutex> lk(g_seed_mu);
  if (!rgbds) { g_seed_targets.erase(ws); return; }
  // bounded: a caller that allocates a fresh workspace per forward (the autograd ops do) would otherwise grow the map by one
  // entry per distinct address for the life of the process.  Dropping old entries is safe: a backward that finds none runs the
  // separate k_stash_bwd instead of the fused seeds (same results, one launch more).
  if (g_seed_targets.size() >= 4096) g_seed_targets.clear();
  g_seed_targets[ws] = rgbds;
}
static bool forward_wrote_seeds_for(const void* ws, const void* rgbds) {
  std::lock_guard<std::mutex> lk(g_seed_mu);
  auto it = g_seed_targets.find(ws);
  return it != g_seed_targets.end() && it->second == rgbds;
}

static int launch_bwd_any(FieldBwdArgs& a, int blocks, hipStream_t st) {
  static const bool force32 = getenv("NGM_BWD32") != nullptr;
  static const bool timing = getenv("NGM_PHASE_TIMING") != nullptr;
  a.debug_cycles = nullptr;
  if (timing) {
    if (!g_debug_cycles) { (void)hipMalloc(&g_debug_cycles, NGM_FWD_DEBUG_WORDS * 8); (void)hipMemset(g_debug_cycles, 0, NGM_FWD_DEBUG_WORDS * 8); }
    a.debug_cycles = g_debug_cycles;
  }
  // order of preference: stashed activations (no forward recompute) -> 16-sample-tile recompute ->
  // 32-sample-tile recompute
  static const bool no_b3 = getenv("NGM_NO_BWD_B3") != nullptr;
  int e = NGM_E_UNSUPPORTED;
  g_last_stash_mode = a.act ? a.act_half : -1;
  if (e == NGM_E_UNSUPPORTED && !force32 && !no_b3 && a.act) { e = ngm_launch_field_bwd_b3(a, blocks, st); g_last_bwd_variant = 3; }
  if (a.act_half && e == NGM_E_UNSUPPORTED) {                   // no other kernel reads a half stash: never fall through
    snprintf(g_err, sizeof(g_err), "render_bwd: the forward stashed one hidden layer (split path) but k_field_bwd_b3 does not take this problem");
    return NGM_E_INVALID;
  }
  if (e == NGM_E_UNSUPPORTED && !force32 && !no_b3 && a.act) { e = ngm_launch_hash_mlp_bwd(a, blocks, st); g_last_bwd_variant = 5; }
  g_last_comp_fused = (a.fused_comp && e == 0) ? 1 : 0;
  if (a.fused_comp && e) return e;                              // no other kernel composites: never fall through
  if (e == NGM_E_UNSUPPORTED && !force32 && a.act) { e = ngm_launch_field_bwd16s(a, blocks, st); g_last_bwd_variant = 2; }
  if (e == NGM_E_UNSUPPORTED && !force32) { e = ngm_launch_field_bwd16(a, blocks, st); g_last_bwd_variant = 1; }
  if (e == NGM_E_UNSUPPORTED) { e = ngm_launch_field_bwd(a, blocks, st); g_last_bwd_variant = 0; }
  return e;
}
static int check_launch(const char* what) {
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    snprintf(g_err, sizeof(g_err), "%s: %s", what, hipGetErrorString(e));
    return NGM_E_HIP;
  }
  return NGM_OK;
}
static int num_cus() {
  static int cached = 0;
  if (cached) return cached;
  int dev = 0;
  hipDeviceProp_t prop;
  if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0)
    cached = prop.multiProcessorCount;
  else
    cached = 256;  // MI355X
  (void)hipGetLastError();
  return cached;
}
static inline int64_t align_up(int64_t v, int64_t a) { return (v + a - 1) / a * a; }
// floats of LDS taken by one field's weights (mirrors FieldLds<MI,MH,L>::TOTAL in ngm_field.h)
static int64_t field_lds_floats(const ngm_field_cfg* fc) {
  const int64_t MI = (fc->dim_enc + 31) / 32, MH = (fc->dim_hidden + 31) / 32;
  const int64_t MC = MH + (fc->skip_mode == NGM_SKIP_CONCAT ? MI : 0);      // input tiles of layers >= 1 and of the output layer
  int64_t t = MI * 32 * 4;
  for (int l = 0; l < fc->num_layers; ++l) t += MH * (l == 0 ? MI : MC) * 16 * 2 * 33 + MH * 32;
  return t + MC * 32 * 4 + 8;
}

static int check_field_cfg(const ngm_field_cfg* fc) {
  if (!fc) return fail(NGM_E_INVALID, "field cfg is NULL");
  if (fc->dim_out != 4) return fail(NGM_E_UNSUPPORTED, "dim_out must be 4 (r,g,b,geometry)");
  if (fc->num_layers < 1 || fc->num_layers > NGM_MAX_LAYERS) return fail(NGM_E_UNSUPPORTED, "num_layers out of range");
  if (fc->encoding == NGM_ENC_PERMUTO) {
    if (fc->nr_feat_per_level != 2 || fc->nr_levels < 1 || fc->nr_levels > 16 || fc->dim_enc != 2 * fc->nr_levels)
      return fail(NGM_E_UNSUPPORTED, "permutohedral encoding: nr_feat_per_level must be 2, nr_levels <= 16, no concat_points");
    if (fc->log2_hashmap_size < 4 || fc->log2_hashmap_size > 24) return fail(NGM_E_UNSUPPORTED, "permutohedral: log2_hashmap_size out of range");
    if (!(fc->level_scale[0] > 0.f)) return fail(NGM_E_INVALID, "permutohedral: level_scale not filled (ngm_permuto_fill_scales)");
  }
  if (fc->encoding == NGM_ENC_NERF && fc->dim_enc != 6 * fc->num_octaves) return fail(NGM_E_INVALID, "nerf: dim_enc != 6*octaves");
  if (fc->encoding == NGM_ENC_TRIPLANE) {
    if (fc->tri_resolution < 2 || fc->tri_resolution > 4096 || fc->tri_mode < NGM_TRI_SUM || fc->tri_mode > NGM_TRI_CONCAT ||
        (fc->tri_mode == NGM_TRI_CONCAT && fc->dim_enc % 3))
      return fail(NGM_E_INVALID, "triplane: resolution / mode / dim_enc inconsistent");
  }
  if (fc->encoding == NGM_ENC_NONE && fc->dim_enc != 3) return fail(NGM_E_INVALID, "no encoding: dim_enc must be 3");
  if (fc->dim_enc < 1 || fc->dim_enc > 64 || fc->dim_hidden < 1 || fc->dim_hidden > 64)
    return fail(NGM_E_UNSUPPORTED, "dim_enc / dim_hidden must be <= 64");
  if (((fc->dim_enc + 31) / 32) != ((fc->dim_hidden + 31) / 32) && !(fc->dim_enc <= 32 && fc->dim_hidden <= 32))
    return fail(NGM_E_UNSUPPORTED, "dim_enc and dim_hidden must pad to the same multiple of 32");
  if (fc->skip_mode != NGM_SKIP_NO && fc->skip_mode != NGM_SKIP_ADD && fc->skip_mode != NGM_SKIP_CONCAT)
    return fail(NGM_E_UNSUPPORTED, "skip_mode: no / add / concat");
  // skip connections are encoding-agnostic like models.py:159-169 (every encoding x {no, add, concat}); "add" needs room for
  // the encoding in the hidden units (the reference adds out[..., :D] += encoding: D <= H)
  if (fc->skip_mode == NGM_SKIP_ADD && fc->dim_hidden < fc->dim_enc)
    return fail(NGM_E_UNSUPPORTED, "skip_mode add: needs dim_hidden >= dim_enc");
  if (fc->activation_stash != NGM_STASH_FULL && fc->activation_stash != NGM_STASH_HALF)
    return fail(NGM_E_INVALID, "activation_stash: NGM_STASH_FULL / NGM_STASH_HALF (ABI 10: is the struct the caller built 276 bytes?)");
  if (fc->hash_grad_atomics != NGM_HASH_ATOMICS_EXACT && fc->hash_grad_atomics != NGM_HASH_ATOMICS_FLOAT)
    return fail(NGM_E_INVALID, "hash_grad_atomics: NGM_HASH_ATOMICS_EXACT / NGM_HASH_ATOMICS_FLOAT");
  return NGM_OK;
}
static int check_params(const ngm_field_cfg* fc, const ngm_params* pr) {
  if (!pr) return fail(NGM_E_INVALID, "params is NULL");
  if (fc->encoding == NGM_ENC_FOURIER && !pr->enc_w) return fail(NGM_E_INVALID, "fourier encoding needs enc_w");
  if (fc->encoding == NGM_ENC_PERMUTO && (!pr->lattice || !pr->shift)) return fail(NGM_E_INVALID, "permutohedral encoding needs lattice + shift");
  if (fc->encoding == NGM_ENC_TRIPLANE && (!pr->planes || pr->dtype != NGM_DT_F32)) return fail(NGM_E_INVALID, "triplane encoding needs fp32 planes");
  for (int l = 0; l <= fc->num_layers; ++l)
    if (!pr->w[l] || !pr->b[l]) return fail(NGM_E_INVALID, "missing layer weight/bias pointer");
  if (pr->dtype != NGM_DT_F32 && pr->dtype != NGM_DT_BF16 && pr->dtype != NGM_DT_F16) return fail(NGM_E_INVALID, "params.dtype");
  return NGM_OK;
}

extern "C" {

int ngm_abi_version(void) { return NGM_ABI_VERSION; }

int ngm_permuto_fill_scales(ngm_field_cfg* cfg) {
  if (!cfg || cfg->nr_levels < 1 || cfg->nr_levels > 16 || !(cfg->coarsest_scale > 0) || !(cfg->finest_scale > 0))
    return fail(NGM_E_INVALID, "ngm_permuto_fill_scales: bad configuration");
  const int L = cfg->nr_levels;
  const double la = log10((double)cfg->coarsest_scale), lb = log10((double)cfg->finest_scale);
  for (int l = 0; l < L; ++l) {
    double sig = (L == 1) ? cfg->coarsest_scale : pow(10.0, la + (lb - la) * (double)l / (double)(L - 1));
    if (l == 0) sig = cfg->coarsest_scale;
    if (l == L - 1 && L > 1) sig = cfg->finest_scale;
    for (int i = 0; i < 3; ++i) cfg->level_scale[3 * l + i] = (float)((1.0 / sqrt((double)((i + 1) * (i + 2)))) / sig);
  }
  return NGM_OK;
}
const char* ngm_last_error(void) { return g_err; }

int ngm_profile_enable(int32_t on) {
  std::lock_guard<std::mutex> lk(g_prof_mu);
  g_prof_on = on != 0;
  return NGM_OK;
}
int ngm_profile_reset(void) {
  std::lock_guard<std::mutex> lk(g_prof_mu);
  prof_drain();
  for (int k = 0; k < NGM_K_COUNT; ++k) { g_prof_ms[k] = 0; g_prof_n[k] = 0; }
  return NGM_OK;
}
int ngm_profile_read(int32_t kernel_id, double* total_ms, int64_t* launches) {
  if (kernel_id < 0 || kernel_id >= NGM_K_COUNT) return fail(NGM_E_INVALID, "bad kernel id");
  std::lock_guard<std::mutex> lk(g_prof_mu);
  prof_drain();
  if (total_ms) *total_ms = g_prof_ms[kernel_id];
  if (launches) *launches = g_prof_n[kernel_id];
  return NGM_OK;
}

// ------------------------------------------------------------------------------------------------
// training-target sampler
// ------------------------------------------------------------------------------------------------
static int check_keyframes(const ngm_keyframes* kf) {
  if (!kf || !kf->c2ws || !kf->rgbd || !kf->frame_to_store) return fail(NGM_E_INVALID, "target sampler: NULL keyframe argument");
  if (kf->num_frames < 1 || kf->height < 1 || kf->width < 1) return fail(NGM_E_INVALID, "target sampler: empty keyframe set");
  return NGM_OK;
}
int ngm_target_visibility(const ngm_keyframes* kf, int32_t F, const float* field_pos, int32_t num_offsets, const float* offsets,
                          float radius, uint8_t* kf_mask, float* bbox, void* stream) {
  int e = check_keyframes(kf);
  if (e) return e;
  if (F < 0 || num_offsets < 1 || !field_pos || !offsets || !kf_mask || !bbox) return fail(NGM_E_INVALID, "ngm_target_visibility: bad argument");
  if (F == 0) return NGM_OK;
  ngm_launch_target_visibility(*kf, F, field_pos, num_offsets, offsets, radius, kf_mask, bbox, (hipStream_t)stream);
  return check_launch("ngm_target_visibility");
}
int ngm_target_rays(const ngm_keyframes* kf, int32_t F, int32_t R, const float* field_pos, float radius, const float* bbox,
                    const int64_t* frame_cids, const float* u_xy, const ngm_target_out* out, void* stream) {
  int e = check_keyframes(kf);
  if (e) return e;
  if (F < 0 || R < 1 || !field_pos || !bbox || !frame_cids || !u_xy || !out || !out->ijs || !out->near || !out->far || !out->gt ||
      !out->rgbds || !out->rgb_mask || !out->depth_mask || !out->term_probs || !out->term_mask)
    return fail(NGM_E_INVALID, "ngm_target_rays: bad argument");
  if (F == 0) return NGM_OK;
  ngm_launch_target_rays(*kf, F, R, field_pos, radius, bbox, frame_cids, u_xy, *out, (hipStream_t)stream);
  return check_launch("ngm_target_rays");
}

int ngm_target_sv_intersect(int32_t F, int64_t N, const float* field_pos_cam, const float* points_cam, float radius, uint8_t* hit,
                            void* stream) {
  if (F < 0 || N < 0 || radius < 0.f || (F > 0 && N > 0 && (!field_pos_cam || !points_cam || !hit)))
    return fail(NGM_E_INVALID, "ngm_target_sv_intersect: bad argument");
  if (F == 0 || N == 0) return NGM_OK;
  if (F > 65535) return fail(NGM_E_UNSUPPORTED, "ngm_target_sv_intersect: more than 65535 candidate fields");
  ngm_launch_target_sv_intersect(F, N, field_pos_cam, points_cam, radius, hit, (hipStream_t)stream);
  return check_launch("ngm_target_sv_intersect");
}
int ngm_target_sv_rays(int32_t F, int32_t R, const float* field_pos_cam, float radius, const int64_t* pts_ijs, const int64_t* segments,
                       const float* image, int32_t height, int32_t width, float fx, float fy, float cx, float cy,
                       const ngm_target_out* out, void* stream) {
  if (F < 0 || R < 1 || height < 1 || width < 1 || !field_pos_cam || !pts_ijs || !segments || !image || !out || !out->ijs || !out->near ||
      !out->far || !out->gt || !out->rgbds || !out->rgb_mask || !out->depth_mask || !out->term_probs || !out->term_mask)
    return fail(NGM_E_INVALID, "ngm_target_sv_rays: bad argument");
  if (F == 0) return NGM_OK;
  ngm_launch_target_sv_rays(F, R, field_pos_cam, radius, pts_ijs, segments, image, height, width, fx, fy, cx, cy, *out, (hipStream_t)stream);
  return check_launch("ngm_target_sv_rays");
}

int ngm_debug_last_bwd_variant(void) { return g_last_bwd_variant; }
int ngm_debug_last_matmul(int which) { return (which >= 0 && which < 3) ? g_ngm_last_matmul[which] : -1; }
int ngm_debug_last_fwd_one_tile(void) { return g_ngm_last_fwd_one_tile; }
int ngm_debug_last_comp_fused(void) { return g_last_comp_fused; }
int ngm_debug_disable_fused_comp(int on) { const int old = g_no_fused_comp; g_no_fused_comp = on ? 1 : 0; return old; }

static unsigned long long* g_debug_cycles_fwd = nullptr;
int ngm_debug_fwd_phase_cycles(unsigned long long* out528) {
  if (!g_debug_cycles_fwd || !out528) return NGM_E_INVALID;
  (void)hipDeviceSynchronize();
  return hipMemcpy(out528, g_debug_cycles_fwd, NGM_FWD_DEBUG_WORDS * 8, hipMemcpyDeviceToHost) == hipSuccess ? NGM_OK : NGM_E_HIP;
}
int ngm_debug_phase_cycles(unsigned long long* out528) {
  if (!g_debug_cycles || !out528) return NGM_E_INVALID;
  (void)hipDeviceSynchronize();
  return hipMemcpy(out528, g_debug_cycles, NGM_FWD_DEBUG_WORDS * 8, hipMemcpyDeviceToHost) == hipSuccess ? NGM_OK : NGM_E_HIP;
}

int ngm_device_info(int* ncu, char* name, int name_len) {
  int dev = 0;
  hipDeviceProp_t prop;
  if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) {
    (void)hipGetLastError();
    return fail(NGM_E_HIP, "no HIP device");
  }
  if (ncu) *ncu = prop.multiProcessorCount;
  if (name && name_len > 0) snprintf(name, name_len, "%s (%s)", prop.name, prop.gcnArchName);
  return NGM_OK;
}

// ------------------------------------------------------------------------------------------------
int ngm_sample_rays(const ngm_render_cfg* cfg, const ngm_rays* rays, float* points_cam, float* distances, float* dirs,
                    void* stream) {
  if (!cfg || !rays || !rays->ijs) return fail(NGM_E_INVALID, "ngm_sample_rays: NULL argument");
  const int S = cfg->num_samples_coarse + (rays->gt ? cfg->num_samples_guided : 0);
  if (S < 1) return fail(NGM_E_INVALID, "no samples");
  ngm_launch_sampler(cfg, rays, S, points_cam, distances, dirs, nullptr, (hipStream_t)stream);
  return check_launch("ngm_sample_rays");
}

int ngm_sample_rays_world(const ngm_render_cfg* cfg, const ngm_rays* rays, float* points_cam, float* points_world,
                          float* distances, float* dirs, void* stream) {
  if (!cfg || !rays || !rays->ijs) return fail(NGM_E_INVALID, "ngm_sample_rays_world: NULL argument");
  if (points_world && !rays->c2ws) return fail(NGM_E_INVALID, "ngm_sample_rays_world: c2ws required");
  const int S = cfg->num_samples_coarse + (rays->gt ? cfg->num_samples_guided : 0);
  if (S < 1) return fail(NGM_E_INVALID, "no samples");
  ngm_launch_sampler(cfg, rays, S, points_cam, distances, dirs, points_world, (hipStream_t)stream);
  return check_launch("ngm_sample_rays_world");
}

int ngm_sample_rays_weighted(const ngm_render_cfg* cfg, const ngm_rays* rays, int32_t num_bins, const float* boundaries,
                             const float* weights, float* points_cam, float* distances, float* dirs, void* stream) {
  if (!cfg || !rays || !rays->ijs || !boundaries || !weights) return fail(NGM_E_INVALID, "ngm_sample_rays_weighted: NULL argument");
  // camera.py:260-261: "Either both or none of weights and boundaries must be None" -- both are required here; one draw array
  // without the other would mix the reference's two torch.rand streams with the Philox stream
  if ((rays->u_coarse == nullptr) != (rays->u_guided == nullptr))
    return fail(NGM_E_INVALID, "ngm_sample_rays_weighted: u_coarse (bin draws) and u_guided (offset draws) must both be given or both be NULL");
  const int S = cfg->num_samples_coarse;
  if (S < 1 || num_bins < 1) return fail(NGM_E_INVALID, "ngm_sample_rays_weighted: no samples / no bins");
  ngm_launch_sampler_weighted(cfg, rays, S, num_bins, boundaries, weights, points_cam, distances, dirs, (hipStream_t)stream);
  return check_launch("ngm_sample_rays_weighted");
}

// ------------------------------------------------------------------------------------------------
int ngm_field_eval_fwd(const ngm_field_cfg* fcfg, const ngm_params* params, int32_t F, int64_t P, const float* points,
                       const float* field_pos, const float* field_quat, float* out, void* stream) {
  int rc = check_field_cfg(fcfg);
  if (rc) return rc;
  rc = check_params(fcfg, params);
  if (rc) return rc;
  if (!points || !out || F < 1 || P < 0) return fail(NGM_E_INVALID, "ngm_field_eval_fwd: bad argument");
  if ((field_pos == nullptr) != (field_quat == nullptr)) return fail(NGM_E_INVALID, "pos/quat must both be given");
  if (P == 0) return NGM_OK;
  PointsFwdArgs a;
  memset(&a, 0, sizeof(a));
  a.fc = *fcfg; a.pr = *params; a.F = F; a.P = P; a.points = points; a.pos = field_pos; a.quat = field_quat; a.out = out;
  const int ncu = num_cus();
  int64_t bpf = (ncu + F - 1) / F;                         // workgroups per field
  int64_t per = align_up((P + bpf - 1) / bpf, NGM_BLOCK);
  bpf = (P + per - 1) / per;
  a.per_block = per;
  rc = ngm_launch_points_fwd(a, (int)(bpf * F), (hipStream_t)stream);
  if (rc) return fail(rc, "ngm_field_eval_fwd: no kernel for this (D,H,L)");
  return check_launch("ngm_field_eval_fwd");
}

int ngm_encode_fwd(const ngm_field_cfg* fcfg, const ngm_params* params, int32_t F, int64_t P, const float* points,
                   const float* field_pos, const float* field_quat, float* out, void* stream) {
  int rc = check_field_cfg(fcfg);
  if (rc) return rc;
  rc = check_params(fcfg, params);
  if (rc) return rc;
  if (!points || !out || F < 1 || P < 0) return fail(NGM_E_INVALID, "ngm_encode_fwd: bad argument");
  if ((field_pos == nullptr) != (field_quat == nullptr)) return fail(NGM_E_INVALID, "pos/quat must both be given");
  if (P == 0) return NGM_OK;
  rc = ngm_launch_encode_points(*fcfg, *params, F, P, points, field_pos, field_quat, out, (hipStream_t)stream);
  if (rc) return fail(rc, "ngm_encode_fwd: encoding not available as a standalone stage (triplane)");
  return check_launch("ngm_encode_fwd");
}

static int64_t hash_scratch_bytes(const ngm_field_cfg* fc, int F, int64_t P);
static void carve_hash_scratch(const ngm_field_cfg* fc, int F, int64_t P, char* base, FieldBwdArgs& a);

int64_t ngm_encode_bwd_workspace(const ngm_field_cfg* fcfg, int32_t F, int64_t P) {
  if (!fcfg || F < 1 || P < 0) return NGM_E_INVALID;
  if (fcfg->encoding == NGM_ENC_PERMUTO) return hash_scratch_bytes(fcfg, F, P) + 512;
  if (fcfg->encoding == NGM_ENC_FOURIER) return ngm_encode_bwd_fourier_scratch(F, P) + 512;
  return 256;
}

int ngm_encode_bwd(const ngm_field_cfg* fcfg, const ngm_params* params, int32_t F, int64_t P, const float* points,
                   const float* field_pos, const float* field_quat, const float* d_enc, const ngm_grads* grads, void* workspace,
                   int64_t workspace_bytes, void* stream) {
  int rc = check_field_cfg(fcfg);
  if (rc) return rc;
  rc = check_params(fcfg, params);
  if (rc) return rc;
  if (!points || !d_enc || !grads || F < 1 || P < 0) return fail(NGM_E_INVALID, "ngm_encode_bwd: bad argument");
  if ((field_pos == nullptr) != (field_quat == nullptr)) return fail(NGM_E_INVALID, "pos/quat must both be given");
  if (fcfg->encoding == NGM_ENC_TRIPLANE)
    return fail(NGM_E_UNSUPPORTED, "ngm_encode_bwd: the triplane encoding has no standalone backward stage (ngm_field_eval_bwd)");
  if (fcfg->encoding != NGM_ENC_PERMUTO && fcfg->encoding != NGM_ENC_FOURIER) return NGM_OK;     // no parameters
  if (!workspace || workspace_bytes < ngm_encode_bwd_workspace(fcfg, F, P)) return fail(NGM_E_WORKSPACE, "ngm_encode_bwd: workspace too small");
  char* base = reinterpret_cast<char*>(align_up((int64_t)workspace, 256));
  hipStream_t st = (hipStream_t)stream;
  if (fcfg->encoding == NGM_ENC_FOURIER) {
    if (!grads->enc_w) return fail(NGM_E_INVALID, "ngm_encode_bwd: grads->enc_w is NULL");
    if (P == 0) {
      const int64_t n = (int64_t)(fcfg->dim_enc - (fcfg->raw_coords ? 3 : 0)) * 3;
      for (int f = 0; f < F; ++f) (void)hipMemsetAsync(grads->enc_w + f * grads->enc_w_stride, 0, 4 * (size_t)n, st);
      return check_launch("ngm_encode_bwd");
    }
    rc = ngm_launch_encode_bwd_fourier(*fcfg, *params, F, P, points, field_pos, field_quat, d_enc, grads->enc_w, grads->enc_w_stride,
                                       reinterpret_cast<float*>(base), st);
    if (rc) return fail(rc, "ngm_encode_bwd: launch failed");
    return check_launch("ngm_encode_bwd");
  }
  if (!grads->lattice) return fail(NGM_E_INVALID, "ngm_encode_bwd: grads->lattice is NULL");
  FieldBwdArgs a;
  memset(&a, 0, sizeof(a));
  a.fc = *fcfg; a.pr = *params; a.F = F; a.P = P;
  carve_hash_scratch(fcfg, F, P, base, a);
  a.lattice_grad = grads->lattice; a.lattice_grad_stride = grads->lattice_stride;
  if (P == 0) {
    const int64_t n = (int64_t)fcfg->nr_levels * ((int64_t)1 << fcfg->log2_hashmap_size) * 2;
    for (int f = 0; f < F; ++f) (void)hipMemsetAsync(grads->lattice + f * grads->lattice_stride, 0, 4 * (size_t)n, st);
    return check_launch("ngm_encode_bwd");
  }
  rc = ngm_launch_encode_bwd_prep_hash(*fcfg, *params, F, P, points, field_pos, field_quat, d_enc, a.hash_dE, a.hash_xyz, st);
  if (!rc) rc = ngm_launch_hash_grad(a, st);
  if (rc) return fail(rc, "ngm_encode_bwd: table size not supported by the table-gradient kernel");
  return check_launch("ngm_encode_bwd");
}

// unit: samples a workgroup's range is a multiple of -- whole 32-sample tiles for each of its waves (4 waves: 128; the
// hash network's 8-wave backward: 256)
static void plan_bwd(int F, int64_t P, int64_t* per_block, int* bpf, int64_t unit = 32 * NGM_WAVES_PER_BLOCK) {
  const int ncu = num_cus();
  int64_t b = (ncu + F - 1) / F;
  int64_t per = align_up((P + b - 1) / b, unit);
  if (per < unit) per = unit;
  *per_block = per;
  *bpf = (int)((P + per - 1) / per);
}
// permutohedral backward scratch: dL/dE per sample and level (8 B) + scaled local position (16 B)
static int64_t tri_numel(const ngm_field_cfg* fc) {       // floats of one field's planes (3, C, res, res)
  const int64_t C = fc->tri_mode == NGM_TRI_CONCAT ? fc->dim_enc / 3 : fc->dim_enc;
  return 3 * C * fc->tri_resolution * fc->tri_resolution;
}
static int64_t hash_scratch_bytes(const ngm_field_cfg* fc, int F, int64_t P) {
  if (fc->encoding == NGM_ENC_TRIPLANE) return align_up((int64_t)F * tri_numel(fc) * 8, 256);   // Q23.40 accumulators
  if (fc->encoding != NGM_ENC_PERMUTO) return 0;
  const int64_t part = (int64_t)F * fc->nr_levels * 8 * 2 * ((int64_t)1 << fc->log2_hashmap_size) * 4;
  return align_up((int64_t)fc->nr_levels * F * P * 8, 256) + align_up((int64_t)F * P * 16, 256) + align_up(part, 256);
}
static void carve_hash_scratch(const ngm_field_cfg* fc, int F, int64_t P, char* base, FieldBwdArgs& a) {
  a.hash_dE = nullptr; a.hash_xyz = nullptr; a.hash_part = nullptr; a.tri_acc = nullptr; a.tri_numel = 0;
  if (fc->encoding == NGM_ENC_TRIPLANE) { a.tri_acc = reinterpret_cast<long long*>(base); a.tri_numel = tri_numel(fc); return; }
  if (fc->encoding != NGM_ENC_PERMUTO) return;
  a.hash_dE = reinterpret_cast<float2*>(base);
  a.hash_xyz = reinterpret_cast<float4*>(base + align_up((int64_t)fc->nr_levels * F * P * 8, 256));
  a.hash_part = reinterpret_cast<float*>(base + align_up((int64_t)fc->nr_levels * F * P * 8, 256) + align_up((int64_t)F * P * 16, 256));
}
static int64_t param_pad(const ngm_field_cfg* fc) {
  int64_t e, w[NGM_MAX_LAYERS + 1], b[NGM_MAX_LAYERS + 1];
  return align_up(ngm_param_offsets(fc, &e, w, b), 64);
}

int64_t ngm_field_eval_bwd_workspace(const ngm_field_cfg* fcfg, int32_t F, int64_t P) {
  if (check_field_cfg(fcfg)) return NGM_E_INVALID;
  int64_t per; int bpf;
  plan_bwd(F, P, &per, &bpf);
  return align_up((int64_t)F * bpf * param_pad(fcfg) * 4 + 256, 256) + hash_scratch_bytes(fcfg, F, P) + 256;
}

int ngm_field_eval_bwd(const ngm_field_cfg* fcfg, const ngm_params* params, int32_t F, int64_t P, const float* points,
                       const float* field_pos, const float* field_quat, const float* d_out, const ngm_grads* grads,
                       void* workspace, int64_t workspace_bytes, void* stream) {
  int rc = check_field_cfg(fcfg);
  if (rc) return rc;
  rc = check_params(fcfg, params);
  if (rc) return rc;
  if (!points || !d_out || !grads || F < 1 || P < 1) return fail(NGM_E_INVALID, "ngm_field_eval_bwd: bad argument");
  if (workspace_bytes < ngm_field_eval_bwd_workspace(fcfg, F, P) || !workspace) return fail(NGM_E_WORKSPACE, "workspace too small");
  FieldBwdArgs a;
  memset(&a, 0, sizeof(a));
  a.fc = *fcfg; a.pr = *params; a.F = F; a.P = P; a.points = points; a.pos = field_pos; a.quat = field_quat;
  a.d_out = reinterpret_cast<const float4*>(d_out);
  plan_bwd(F, P, &a.per_block, &a.blocks_per_field);
  a.p_pad = param_pad(fcfg);
  a.partials = reinterpret_cast<float*>(align_up((int64_t)workspace, 256));
  carve_hash_scratch(fcfg, F, P, reinterpret_cast<char*>(a.partials) + align_up((int64_t)F * a.blocks_per_field * a.p_pad * 4 + 256, 256), a);
  rc = prep_lattice_grad(fcfg, grads, F, a, (hipStream_t)stream);
  if (rc) return rc;
  rc = launch_bwd_any(a, a.blocks_per_field * F, (hipStream_t)stream);
  if (rc) return fail(rc, "ngm_field_eval_bwd: no kernel for this (D,H,L)");
  rc = check_launch("ngm_field_eval_bwd");
  if (rc) return rc;
  if (fcfg->encoding == NGM_ENC_TRIPLANE) { ngm_launch_tri_finish(a, (hipStream_t)stream); rc = check_launch("ngm_tri_finish"); if (rc) return rc; }
  if (fcfg->encoding == NGM_ENC_PERMUTO) {
    rc = ngm_launch_hash_grad(a, (hipStream_t)stream);
    if (rc) return fail(rc, "permutohedral backward: hash table too large for the LDS-staged scatter");
    rc = check_launch("ngm_hash_grad");
    if (rc) return rc;
  }
  GradReduceArgs g;
  memset(&g.adam, 0, sizeof(g.adam));
  g.fc = *fcfg; g.gr = *grads; g.F = F; g.blocks_per_field = a.blocks_per_field; g.partials = a.partials; g.p_pad = a.p_pad;
  ngm_launch_grad_reduce(g, (hipStream_t)stream);
  return check_launch("ngm_grad_reduce");
}

// ---- training forward / backward of the point evaluation with an activation stash (ABI 11) ---------------------------------
// The reference's unchanged _optimization_iteration reaches NeuralFieldSet.forward(use_vmap=True) under autograd; its backward
// used to be k_field_bwd16 alone (fp32 MFMA, every hidden layer recomputed: 0.45 of the fp32 MFMA peak).  With the stash the
// forward writes what the fused training step's forward writes (ngm_field.h ActStash, 256 B per sample and hidden layer) and
// the backward is the fused step's kernel, k_field_bwd_b3, in point mode.
static int act_stash_kind(const ngm_field_cfg* fc);
static bool field_eval_stash_applies(const ngm_field_cfg* fc, int32_t F, int64_t P) {
  if (act_stash_kind(fc) != 1 || !bwd_b3_is_default()) return false;
  FieldBwdArgs probe;
  memset(&probe, 0, sizeof(probe));
  probe.fc = *fc; probe.F = F; probe.P = P;
  probe.act = reinterpret_cast<const float*>(1); probe.points = reinterpret_cast<const float*>(1);
  return ngm_field_bwd_b3_applies(probe);
}
static int64_t field_eval_stash_stride(int32_t F, int64_t P) { return align_up((int64_t)F * P, 32) * 64 + 2048; }   // floats per layer
int64_t ngm_field_eval_stash_bytes(const ngm_field_cfg* fcfg, int32_t F, int64_t P) {
  if (check_field_cfg(fcfg) || F < 1 || P < 0) return NGM_E_INVALID;
  if (P == 0 || !field_eval_stash_applies(fcfg, F, P)) return 0;
  return align_up(fcfg->num_layers * field_eval_stash_stride(F, P) * 4 + 64, 256) + 256;
}
int ngm_field_eval_fwd_train(const ngm_field_cfg* fcfg, const ngm_params* params, int32_t F, int64_t P, const float* points,
                             const float* field_pos, const float* field_quat, float* out, void* stash, int64_t stash_bytes,
                             void* stream) {
  int rc = check_field_cfg(fcfg);
  if (rc) return rc;
  rc = check_params(fcfg, params);
  if (rc) return rc;
  if (!points || !out || F < 1 || P < 0) return fail(NGM_E_INVALID, "ngm_field_eval_fwd_train: bad argument");
  if ((field_pos == nullptr) != (field_quat == nullptr)) return fail(NGM_E_INVALID, "pos/quat must both be given");
  if (P == 0) return NGM_OK;
  const int64_t need = ngm_field_eval_stash_bytes(fcfg, F, P);
  if (need <= 0) return fail(NGM_E_UNSUPPORTED, "ngm_field_eval_fwd_train: no stash-reading backward for this configuration (ngm_field_eval_stash_bytes == 0): use ngm_field_eval_fwd / ngm_field_eval_bwd");
  if (!stash || stash_bytes < need) return fail(NGM_E_WORKSPACE, "ngm_field_eval_fwd_train: stash too small (ngm_field_eval_stash_bytes)");
  PointsFwdArgs a;
  memset(&a, 0, sizeof(a));
  a.fc = *fcfg; a.pr = *params; a.F = F; a.P = P; a.points = points; a.pos = field_pos; a.quat = field_quat; a.out = out;
  a.act = reinterpret_cast<float*>(align_up((int64_t)stash, 256)); a.act_layer_stride = field_eval_stash_stride(F, P);
  const int ncu = num_cus();
  int64_t bpf = (ncu + F - 1) / F;
  int64_t per = align_up((P + bpf - 1) / bpf, NGM_BLOCK);
  bpf = (P + per - 1) / per;
  a.per_block = per;
  rc = ngm_launch_points_fwd(a, (int)(bpf * F), (hipStream_t)stream);
  if (rc) return fail(rc, "ngm_field_eval_fwd_train: no kernel for this (D,H,L)");
  return check_launch("ngm_field_eval_fwd_train");
}
int ngm_field_eval_bwd_stash(const ngm_field_cfg* fcfg, const ngm_params* params, int32_t F, int64_t P, const float* points,
                             const float* field_pos, const float* field_quat, const float* d_out, const ngm_grads* grads,
                             const void* stash, int64_t stash_bytes, void* workspace, int64_t workspace_bytes, void* stream) {
  int rc = check_field_cfg(fcfg);
  if (rc) return rc;
  rc = check_params(fcfg, params);
  if (rc) return rc;
  if (!points || !d_out || !grads || F < 1 || P < 1) return fail(NGM_E_INVALID, "ngm_field_eval_bwd_stash: bad argument");
  if ((field_pos == nullptr) != (field_quat == nullptr)) return fail(NGM_E_INVALID, "pos/quat must both be given");
  const int64_t need = ngm_field_eval_stash_bytes(fcfg, F, P);
  if (need <= 0) return fail(NGM_E_UNSUPPORTED, "ngm_field_eval_bwd_stash: no stash-reading backward for this configuration");
  if (!stash || stash_bytes < need) return fail(NGM_E_WORKSPACE, "ngm_field_eval_bwd_stash: stash too small");
  if (workspace_bytes < ngm_field_eval_bwd_workspace(fcfg, F, P) || !workspace) return fail(NGM_E_WORKSPACE, "workspace too small");
  FieldBwdArgs a;
  memset(&a, 0, sizeof(a));
  a.fc = *fcfg; a.pr = *params; a.F = F; a.P = P; a.points = points; a.pos = field_pos; a.quat = field_quat;
  a.d_out = reinterpret_cast<const float4*>(d_out);
  plan_bwd(F, P, &a.per_block, &a.blocks_per_field);
  a.p_pad = param_pad(fcfg);
  a.partials = reinterpret_cast<float*>(align_up((int64_t)workspace, 256));
  a.act = reinterpret_cast<const float*>(align_up((int64_t)stash, 256)); a.act_layer_stride = field_eval_stash_stride(F, P);
  a.debug_cycles = nullptr;
  rc = ngm_launch_field_bwd_b3(a, a.blocks_per_field * F, (hipStream_t)stream);
  g_last_bwd_variant = 3; g_last_stash_mode = 0; g_last_comp_fused = 0;
  if (rc) return fail(rc, "ngm_field_eval_bwd_stash: k_field_bwd_b3 does not take this problem");
  rc = check_launch("ngm_field_eval_bwd_stash");
  if (rc) return rc;
  GradReduceArgs g;
  memset(&g.adam, 0, sizeof(g.adam));
  g.fc = *fcfg; g.gr = *grads; g.F = F; g.blocks_per_field = a.blocks_per_field; g.partials = a.partials; g.p_pad = a.p_pad;
  ngm_launch_grad_reduce(g, (hipStream_t)stream);
  return check_launch("ngm_grad_reduce");
}

// ------------------------------------------------------------------------------------------------
int ngm_composite_fwd(const ngm_render_cfg* cfg, int64_t N, int32_t S, const float* colors, const float* geoms,
                      const float* dists, const float* depths, const float* neus_isds, float* C, float* D, float* Cvar,
                      float* Dvar, float* term, float* weights, void* stream) {
  if (!cfg || !colors || !geoms || !dists || !depths || N < 0) return fail(NGM_E_INVALID, "ngm_composite_fwd: bad argument");
  if (N == 0) return NGM_OK;
  CompositeArgs a;
  memset(&a, 0, sizeof(a));
  a.rc = *cfg; a.N = N; a.S = S; a.colors = colors; a.geoms = geoms; a.dists = dists; a.depths = depths; a.isds = neus_isds;
  a.C = C; a.D = D; a.Cv = Cvar; a.Dv = Dvar; a.term = term; a.weights = weights;
  const int rc = ngm_launch_composite_fwd(a, (hipStream_t)stream);
  if (rc) return fail(rc, "ngm_composite_fwd: unsupported S (max 1024)");
  return check_launch("ngm_composite_fwd");
}

int ngm_composite_fwd_packed(const ngm_render_cfg* cfg, int64_t N, int32_t S, const float* field_out4, const float* dists,
                             const float* points_cam, float* rgbd, float* Cvar, float* Dvar, float* term, void* stream) {
  if (!cfg || !field_out4 || !dists || !points_cam || N < 0) return fail(NGM_E_INVALID, "ngm_composite_fwd_packed: bad argument");
  if (N == 0) return NGM_OK;
  CompositeArgs a;
  memset(&a, 0, sizeof(a));
  a.rc = *cfg; a.N = N; a.S = S; a.out4 = reinterpret_cast<const float4*>(field_out4); a.pcam = points_cam; a.dists = dists;
  a.rgbd = rgbd; a.Cv = Cvar; a.Dv = Dvar; a.term = term;
  const int rc = ngm_launch_composite_fwd(a, (hipStream_t)stream);
  if (rc) return fail(rc, "ngm_composite_fwd_packed: unsupported S (max 1024)");
  return check_launch("ngm_composite_fwd_packed");
}

int ngm_composite_bwd(const ngm_render_cfg* cfg, int64_t N, int32_t S, const float* colors, const float* geoms,
                      const float* dists, const float* depths, const float* neus_isds, const float* dC, const float* dD,
                      const float* dterm, float* d_colors, float* d_geoms, float* d_neus_isds, void* stream) {
  if (!cfg || !colors || !geoms || !dists || !depths || N < 0) return fail(NGM_E_INVALID, "ngm_composite_bwd: bad argument");
  if (N == 0) return NGM_OK;
  CompositeArgs a;
  memset(&a, 0, sizeof(a));
  a.rc = *cfg; a.N = N; a.S = S; a.colors = colors; a.geoms = geoms; a.dists = dists; a.depths = depths; a.isds = neus_isds;
  a.dC = dC; a.dD = dD; a.dterm = dterm; a.d_colors = d_colors; a.d_geoms = d_geoms; a.d_isds = d_neus_isds;
  const int rc = ngm_launch_composite_bwd(a, (hipStream_t)stream);
  if (rc) return fail(rc, "ngm_composite_bwd: S > 1024");
  return check_launch("ngm_composite_bwd");
}

// ------------------------------------------------------------------------------------------------
// fused render / train step
// ------------------------------------------------------------------------------------------------
struct RenderPlan {
  int S, rays_per_block, blocks_fwd, waves_fwd, maxs, b3;
  int64_t per_block_bwd; int blocks_per_field_bwd;
  int64_t p_pad;
  int64_t off_rayseed;             // (F*R, 8) per-ray loss derivatives without the normalisers (fused compositing backward)
  int64_t off_raytab, off_stashA, off_stashB, off_losspart, off_gradpart, off_hash, off_act, act_layer_stride, total;
  int stash_mode;                  // 0: every hidden layer's output; 1: layer 0's only (half stash, k_field_bwd_b3<HS>)
  int64_t off_dout, off_disd;      // neus: separate per-sample gradient buffer, per-ray d loss / d isd
};
// The training forward stashes the hidden activations (64 floats per sample and layer) when the backward
// has a kernel that consumes them: 49..64-wide hidden layers, 1-2 layers, non-hash encoding.  The
// backward then skips its forward recompute.  NGM_NO_ACT_STASH=1 turns it off (recompute; saves
// 256 B * L per sample of workspace).
// 0: none; 1: post-ReLU hidden activations of 64-wide layers (k_field_bwd16s); 2: the hash encoding itself
// (<= 32 features; k_field_bwd16 then skips the simplex search and the table gathers, the hidden layers are recomputed)
static int act_stash_kind(const ngm_field_cfg* fc) {
  static const bool off = getenv("NGM_NO_ACT_STASH") != nullptr;
  if (off) return 0;
  const int th = (fc->dim_hidden + 15) / 16, ti = (fc->dim_enc + 15) / 16;
  if (fc->encoding == NGM_ENC_PERMUTO) return (ti == 2 && fc->dim_hidden <= 32 && fc->skip_mode == NGM_SKIP_NO) ? 2 : 0;   // the stash's consumers (k_hash_mlp_bwd, k_field_bwd16) are skip-less
  if (fc->encoding == NGM_ENC_TRIPLANE) return 0;      // 32-sample-tile backward (recompute: it needs the taps anyway)
  // with a skip connection the stashed activation no longer tells the ReLU mask
  return (fc->skip_mode == NGM_SKIP_NO && th == 4 && ti == 4 && fc->num_layers >= 1 && fc->num_layers <= 2) ? 1 : 0;
}

#ifndef NGM_STASH_DEFAULT
#define NGM_STASH_DEFAULT 0
#endif
// Half stash (round 5): with two hidden layers on the split path the forward stashes layer 0's output only and
// k_field_bwd_b3<.., HS> recomputes the output layer's input from it on the matrix pipe: 256 instead of 512 bytes of stash per
// sample each way.  Decided from the field configuration and the samples per field alone, so that ngm_render_fwd and
// ngm_render_bwd* (which see the same fcfg and rays) agree; the forward's choice is also recorded per workspace.
// NGM_FULL_STASH=1: both layers, as before (A/B, and what every other backward kernel reads).
// NGM_STASH=full | half | planes overrides the default (A/B on one box); NGM_FULL_STASH=1 = NGM_STASH=full.
// ABI 10: the mode is ngm_field_cfg.activation_stash; the override below is a developer A/B switch only (tools/)
static int g_stash_override = -1;      // ngm_debug_stash_mode
static int stash_pref(const ngm_field_cfg* fc) {
  if (g_stash_override >= 0) return g_stash_override;
  static const int env = [] {
    const char* e = getenv("NGM_STASH");
    if (getenv("NGM_FULL_STASH")) return 0;
    if (!e) return -1;
    return !strcmp(e, "full") ? 0 : !strcmp(e, "half") ? 1 : -1;
  }();
  if (env >= 0) return env;
  return fc->activation_stash == NGM_STASH_HALF ? 1 : NGM_STASH_DEFAULT;
}
static bool half_stash_applies(const ngm_field_cfg* fc, int64_t P) {
  if (stash_pref(fc) == 0 || !bwd_b3_is_default() || fc->num_layers != 2 || act_stash_kind(fc) != 1) return false;
  FieldBwdArgs probe;
  memset(&probe, 0, sizeof(probe));
  probe.fc = *fc; probe.P = P; probe.act = reinterpret_cast<const float*>(1);
  return ngm_field_bwd_b3_applies(probe);
}

static RenderPlan plan_render(const ngm_field_cfg* fc, const ngm_render_cfg* rc, int F, int R, bool guided, bool train) {
  RenderPlan p;
  memset(&p, 0, sizeof(p));
  p.S = rc->num_samples_coarse + (guided ? rc->num_samples_guided : 0);
  const int ncu = num_cus();
  // forward: one workgroup per CU-slot; 8 waves (2 per SIMD: latency hiding) unless the per-wave LDS
  // sample planes would not fit in 160 KiB, then 4 waves
  // the bf16 weight planes of the split path (ngm_matmul_mode): 3 planes x 2 bytes per hidden weight
  const int64_t MIp = (fc->dim_enc + 31) / 32, MHp = (fc->dim_hidden + 31) / 32;
  const int64_t b3_bytes = 3 * 2 * 1024 * MHp * (MIp + (fc->num_layers - 1) * MHp);
  const bool b3_compiled = MIp == 2 && MHp == 2 && fc->num_layers <= 2 && fc->skip_mode == NGM_SKIP_NO &&
                           (fc->encoding == NGM_ENC_FOURIER || fc->encoding == NGM_ENC_NONE);
  auto shape = [&](int waves, int64_t extra, int64_t* lds_out, int64_t maxs_cap = 1024) {
    int ch = (ncu + F - 1) / F;
    const int max_ch = (R + waves - 1) / waves;
    if (ch > max_ch) ch = max_ch;
    if (ch < 1) ch = 1;
    int rpb = (R + ch - 1) / ch;
    rpb = (int)align_up(rpb, waves);
    const int rpw = rpb / waves;
    int64_t maxs = align_up((int64_t)(rpw < 32 ? rpw : 32) * p.S, 64);
    if (maxs > maxs_cap) maxs = maxs_cap;                // samples a wave buffers per ray batch (whole rays)
    if (maxs < align_up(p.S, 64)) maxs = align_up(p.S, 64);
    *lds_out = 4 * (field_lds_floats(fc) + (int64_t)waves * (32 * 28 + 5 * maxs)) + extra;
    p.rays_per_block = rpb; p.waves_fwd = waves; p.maxs = (int)maxs;
    p.blocks_fwd = F * ((R + rpb - 1) / rpb);
  };
  const int64_t LDS_MAX = 160 * 1024;
  int64_t lds = 0;
  shape(8, 0, &lds);                                    // exact-fp32 plan first: 8 waves unless its LDS does not fit
  if (lds > LDS_MAX || R < 8) shape(4, 0, &lds);
  // a grid of 8-wave workgroups that leaves CUs idle (a rank with one or two active fields): 4 waves per
  // workgroup on twice as many CUs -- one wave per SIMD runs a step in about half the time (F = 1: 29 -> 25.5 us, split path)
  const bool few = R >= 8 && (int64_t)F * ((R + 7) / 8) < ncu;
  p.b3 = 0;
  // The split path's weight planes (48 KB for two 64-wide layers) compete with the per-wave sample planes for LDS: a
  // batch of many samples per ray (8192 x 256: 1024 samples buffered per wave) leaves no room at 8 waves.  Smaller ray
  // batches per wave do (one 256-sample ray at a time: 8.7 KB per wave), at no measurable cost -- tried in that order.
  auto fit_b3 = [&]() {
    for (int64_t cap : {(int64_t)1024, (int64_t)512, (int64_t)256}) {
      shape(8, b3_bytes, &lds, cap);
      if (lds <= LDS_MAX && R >= 8) return true;
    }
    shape(4, b3_bytes, &lds);
    return lds <= LDS_MAX;
  };
  if (fc->matmul_mode == NGM_MATMUL_BF16X3) {           // explicit: whatever shape makes it fit, else the launcher fails loudly
    p.b3 = 1;
    (void)fit_b3();
  } else if (fc->matmul_mode == NGM_MATMUL_AUTO && b3_compiled && rc->geometry_mode != NGM_GEO_NEUS) {
    const int64_t lds_f32 = lds;
    const RenderPlan keep = p;
    bool ok = false;
    if (few) { shape(4, b3_bytes, &lds); ok = lds <= LDS_MAX; }
    if (!ok) ok = fit_b3() && p.waves_fwd == 8;                       // auto: the planes next to an 8-wave plan, else exact-fp32 MFMA
    if (ok) p.b3 = 1;
    else { p = keep; lds = lds_f32; }
  }
  p.p_pad = param_pad(fc);
  int64_t o = 0;
  if (train) {
    const int64_t NR = (int64_t)F * R, NS = NR * p.S;
    p.off_raytab = o; o = align_up(o + NR * 8 * 4, 256);
    p.off_rayseed = o; o = align_up(o + NR * 8 * 4, 256);
    p.off_stashA = o; o = align_up(o + NS * 16, 256);
    p.off_stashB = o; o = align_up(o + NS * 8, 256);
    p.off_losspart = o; o = align_up(o + (int64_t)p.blocks_fwd * NGM_NUM_LOSS_SUMS * 4, 256);
    if (rc->geometry_mode == NGM_GEO_NEUS) {
      p.off_dout = o; o = align_up(o + NS * 16, 256);
      p.off_disd = o; o = align_up(o + NR * 4, 256);
    }
    plan_bwd(F, (int64_t)R * p.S, &p.per_block_bwd, &p.blocks_per_field_bwd, fc->encoding == NGM_ENC_PERMUTO ? 256 : 128);
    p.off_gradpart = o; o = align_up(o + (int64_t)F * p.blocks_per_field_bwd * p.p_pad * 4, 256);
    p.off_hash = o; o = align_up(o + hash_scratch_bytes(fc, F, (int64_t)R * p.S), 256);
    const int kind = act_stash_kind(fc);
    if (kind == 1) {
      p.act_layer_stride = align_up(NS, 32) * 64 + 2048;      // floats: whole 32-sample tiles (+1: a field may start mid-tile)
      const bool half = half_stash_applies(fc, (int64_t)R * p.S);                              // half stash: layer 0's output only
      p.stash_mode = half ? 1 : 0;
      p.off_act = o; o = align_up(o + (half ? 1 : fc->num_layers) * p.act_layer_stride * 4 + 64, 256);
    } else if (kind == 2) {
      p.act_layer_stride = align_up(NS, 32) * 32 + 1024;      // one "layer": the 32-feature encoding
      p.off_act = o; o = align_up(o + p.act_layer_stride * 4 + 64, 256);
    }
  }
  p.total = o + 256;
  return p;
}

int64_t ngm_render_workspace(const ngm_field_cfg* fcfg, const ngm_render_cfg* rcfg, int32_t F, int32_t R, int32_t train) {
  if (check_field_cfg(fcfg) || !rcfg || F < 1 || R < 1) return NGM_E_INVALID;
  // sized for the guided case (S_c + S_g), the larger of the two
  return plan_render(fcfg, rcfg, F, R, true, train != 0).total;
}

// permutohedral: validate the gradient table (it is fully overwritten by k_hash_reduce)
static int prep_lattice_grad(const ngm_field_cfg* fc, const ngm_grads* grads, int F, FieldBwdArgs& a, hipStream_t st) {
  a.lattice_grad = nullptr; a.lattice_grad_stride = 0;
  a.planes_grad = nullptr; a.planes_grad_stride = 0;
  if (fc->encoding == NGM_ENC_TRIPLANE) {
    if (!grads->planes || grads->planes_stride < tri_numel(fc)) return fail(NGM_E_INVALID, "triplane: grads.planes missing / stride too small");
    if (!a.tri_acc) return fail(NGM_E_WORKSPACE, "triplane: no accumulator scratch");
    a.planes_grad = grads->planes; a.planes_grad_stride = grads->planes_stride;
    (void)hipMemsetAsync(a.tri_acc, 0, (size_t)F * tri_numel(fc) * 8, st);
    return NGM_OK;
  }
  if (fc->encoding != NGM_ENC_PERMUTO) return NGM_OK;
  if (!grads->lattice) return fail(NGM_E_INVALID, "permutohedral: grads.lattice is NULL");
  const int64_t per = (int64_t)fc->nr_levels * ((int64_t)1 << fc->log2_hashmap_size) * 2;
  if (grads->lattice_stride < per) return fail(NGM_E_INVALID, "permutohedral: grads.lattice_stride too small");
  (void)st;   // k_hash_reduce overwrites every table entry: no zero-fill
  a.lattice_grad = grads->lattice; a.lattice_grad_stride = grads->lattice_stride;
  return NGM_OK;
}

static int check_render(const ngm_field_cfg* fc, const ngm_render_cfg* rc, const ngm_params* pr, const ngm_rays* rays) {
  int e = check_field_cfg(fc);
  if (e) return e;
  e = check_params(fc, pr);
  if (e) return e;
  if (!rc || !rays || !rays->ijs || !rays->c2ws || !rays->field_pos || !rays->field_quat)
    return fail(NGM_E_INVALID, "render: NULL ray argument");
  if (rays->F < 1 || rays->R < 1) return fail(NGM_E_INVALID, "render: empty batch");
  if (rc->geometry_mode == NGM_GEO_NEUS && !pr->neus_sd)
    return fail(NGM_E_INVALID, "fused render: the neus geometry mode needs params.neus_sd (the per-field _neus_sd, rm.py:641-644)");
  const int S = rc->num_samples_coarse + (rays->gt ? rc->num_samples_guided : 0);
  if (S < 1 || S > 1024) return fail(NGM_E_UNSUPPORTED, "fused render: samples per ray must be in [1,1024]");
  return NGM_OK;
}

int ngm_render_fwd(const ngm_field_cfg* fcfg, const ngm_render_cfg* rcfg, const ngm_params* params, const ngm_rays* rays,
                   const ngm_targets* targets, const ngm_prediction* pred, float* loss_sums, void* workspace,
                   int64_t workspace_bytes, void* stream) {
  int e = check_render(fcfg, rcfg, params, rays);
  if (e) return e;
  if (!pred) return fail(NGM_E_INVALID, "render_fwd: pred is NULL");
  const bool has_tg = targets != nullptr;
  const bool save = workspace != nullptr;   // keep the per-sample stash for a later backward
  if (has_tg && (!targets->rgbds || !targets->depth_mask)) return fail(NGM_E_INVALID, "render_fwd: incomplete targets");
  if (has_tg && targets->term_mask && !targets->term_probs) return fail(NGM_E_INVALID, "render_fwd: term_mask without term_probs");
  if (has_tg && !save) return fail(NGM_E_WORKSPACE, "render_fwd: targets need a workspace");
  if (save && (!pred->rgbds || !pred->term_probs)) return fail(NGM_E_INVALID, "render_fwd(save): pred.rgbds/term_probs required");
  const RenderPlan p = plan_render(fcfg, rcfg, rays->F, rays->R, rays->gt != nullptr, save);
  if (save && workspace_bytes < p.total) return fail(NGM_E_WORKSPACE, "render_fwd: workspace too small");
  char* ws = reinterpret_cast<char*>(align_up((int64_t)workspace, 256));
  RenderFwdArgs a;
  memset(&a, 0, sizeof(a));
  a.fc = *fcfg; a.pr = *params; a.rc = *rcfg; a.rays = *rays; a.pred = *pred;
  a.fc.matmul_mode = p.b3 ? NGM_MATMUL_BF16X3 : NGM_MATMUL_F32;       // AUTO resolved by the plan (LDS budget of this batch shape)
  if (rcfg->geometry_mode == NGM_GEO_NEUS) {                          // compiled with exact-fp32 MFMA
    if (fcfg->matmul_mode == NGM_MATMUL_BF16X3) return fail(NGM_E_UNSUPPORTED, "render_fwd: neus runs the fp32-MFMA kernel");
    a.fc.matmul_mode = NGM_MATMUL_F32;
    a.neus_sd = params->neus_sd; a.neus_sd_stride = params->neus_sd_stride;
  }
  a.has_targets = has_tg ? 1 : 0;
  if (has_tg) a.tg = *targets;
  a.S = p.S; a.rays_per_block = p.rays_per_block; a.waves_per_block = p.waves_fwd; a.maxs = p.maxs;
  if (save) {
    a.raytab = reinterpret_cast<float*>(ws + p.off_raytab);
    if (has_tg) a.rayseed = reinterpret_cast<float*>(ws + p.off_rayseed);
    a.stashA = reinterpret_cast<float4*>(ws + p.off_stashA);
    a.stashB = reinterpret_cast<float2*>(ws + p.off_stashB);
    a.loss_partials = reinterpret_cast<float*>(ws + p.off_losspart);
    if (p.act_layer_stride) { a.act = reinterpret_cast<float*>(ws + p.off_act); a.act_layer_stride = p.act_layer_stride; }
    a.act_layers = (a.act && p.stash_mode) ? 1 : 0;
    note_forward_stash(workspace, p.stash_mode);
    static const bool timing = getenv("NGM_PHASE_TIMING") != nullptr;
    if (timing) {
      if (!g_debug_cycles_fwd) { (void)hipMalloc(&g_debug_cycles_fwd, NGM_FWD_DEBUG_WORDS * 8); (void)hipMemset(g_debug_cycles_fwd, 0, NGM_FWD_DEBUG_WORDS * 8); }
      a.debug_cycles = g_debug_cycles_fwd;
    }
  }
  e = ngm_launch_render_fwd(a, p.blocks_fwd, (hipStream_t)stream);
  if (e) return fail(e, "render_fwd: no kernel for this (D,H,L)");
  if (save) note_forward_seeds(workspace, a.rayseed ? targets->rgbds : nullptr);
  e = check_launch("ngm_render_fwd");
  if (e) return e;
  if (has_tg && loss_sums) {      // loss_sums == NULL: deferred -- ngm_render_bwd* (loss_sums == NULL) reduces the partials itself
    ngm_launch_loss_reduce(a.loss_partials, p.blocks_fwd, loss_sums,
                           (rays->philox_offset_autoinc && rays->philox_offset_dev) ? const_cast<uint64_t*>(rays->philox_offset_dev) : nullptr,
                           (hipStream_t)stream);
    e = check_launch("ngm_loss_reduce");
  }
  return e;
}

static int render_bwd_common(const ngm_field_cfg* fcfg, const ngm_render_cfg* rcfg, const ngm_params* params,
                             const ngm_rays* rays, StashBwdArgs& sb, const ngm_grads* grads, void* workspace,
                             int64_t workspace_bytes, hipStream_t st, const GradAdam* adam = nullptr,
                             const GradAdam* lattice_adam = nullptr, bool* lattice_adam_applied = nullptr) {
  const RenderPlan p = plan_render(fcfg, rcfg, rays->F, rays->R, rays->gt != nullptr, true);
  if (!workspace || workspace_bytes < p.total) return fail(NGM_E_WORKSPACE, "render_bwd: workspace too small");
  char* ws = reinterpret_cast<char*>(align_up((int64_t)workspace, 256));
  sb.rc = *rcfg; sb.F = rays->F; sb.R = rays->R; sb.S = p.S;
  if (!rays->gt) sb.rc.w_freespace = sb.rc.w_tsdf = 0.f;       // no gt: the reference forms no free-space / TSDF terms (rm.py:624, 632)
  sb.stashA = reinterpret_cast<float4*>(ws + p.off_stashA);
  sb.stashB = reinterpret_cast<const float2*>(ws + p.off_stashB);
  sb.raytab = reinterpret_cast<const float*>(ws + p.off_raytab);
  const bool neus = rcfg->geometry_mode == NGM_GEO_NEUS;
  if (neus) {
    sb.neus_sd = params->neus_sd; sb.neus_sd_stride = params->neus_sd_stride; sb.field_index = params->field_index;
    sb.d_out = reinterpret_cast<float4*>(ws + p.off_dout);
    sb.d_isd_rays = reinterpret_cast<float*>(ws + p.off_disd);
  }
  if (sb.seed_mode == 0 && !sb.loss_sums) {      // deferred loss reduction: the forward left its partials in the workspace
    sb.loss_partials = reinterpret_cast<const float*>(ws + p.off_losspart);
    sb.n_partials = p.blocks_fwd;
    sb.counter = (rays->philox_offset_autoinc && rays->philox_offset_dev)
                     ? reinterpret_cast<unsigned long long*>(const_cast<uint64_t*>(rays->philox_offset_dev)) : nullptr;
  }
  FieldBwdArgs a;
  memset(&a, 0, sizeof(a));
  a.fc = *fcfg; a.pr = *params; a.F = rays->F; a.P = (int64_t)rays->R * p.S; a.S = p.S;
  carve_hash_scratch(fcfg, rays->F, a.P, ws + p.off_hash, a);
  a.per_block = p.per_block_bwd; a.blocks_per_field = p.blocks_per_field_bwd;
  a.raytab = sb.raytab; a.stashB = sb.stashB; a.d_out = neus ? sb.d_out : sb.stashA;
  a.partials = reinterpret_cast<float*>(ws + p.off_gradpart); a.p_pad = p.p_pad;
  if (p.act_layer_stride) { a.act = reinterpret_cast<const float*>(ws + p.off_act); a.act_layer_stride = p.act_layer_stride; }
  {
    const int rec = forward_stash_layers(workspace);       // what the forward on THIS workspace wrote (-1: no record: the plan's mode)
    a.act_half = a.act ? (rec < 0 ? p.stash_mode : rec) : 0;
  }
  // Compositing backward inside the MLP backward (k_field_bwd_b3<FC>, k_hash_mlp_bwd<FC>): loss seeds, pointwise geometry
  // modes, and a kernel that implements it about to be chosen.  Otherwise k_stash_bwd runs first and leaves
  // dL/d(raw outputs) in place of the forward's stash.  NGM_NO_FUSED_COMP=1: never.
  static const bool no_fuse_env = getenv("NGM_NO_FUSED_COMP") != nullptr;
  const bool no_fuse = no_fuse_env || g_no_fused_comp;
  const bool pointwise = rcfg->geometry_mode != NGM_GEO_NEUS && rcfg->geometry_mode != NGM_GEO_DENSITY;
  // the variance-weighted loss modes (gradients through the rendered variances) are k_stash_bwd's
  const bool nll_loss = rcfg->photometric_mode == NGM_PHOTO_GAUSSIAN_NLL || rcfg->depth_mode != NGM_DEPTH_HUBER;
  if (nll_loss && sb.seed_mode == 0 && (!sb.pred.color_vars || !sb.pred.depth_vars))
    return fail(NGM_E_INVALID, "render_bwd: the *_nll loss modes need pred.color_vars and pred.depth_vars");
  bool fuse = !no_fuse && !nll_loss && sb.seed_mode == 0 && pointwise && bwd_b3_is_default() && a.P < (1 << 24) &&
              forward_wrote_seeds_for(workspace, sb.tg.rgbds) && (ngm_field_bwd_b3_applies(a) || ngm_hash_mlp_bwd_applies(a));
  int e = 0;
  const FieldBwdArgs a_plain = a;            // for the fall-back below: the launch records before the fused fields are set
  if (fuse) {
    a.fused_comp = 1; a.rc = sb.rc;
    a.rayseed = reinterpret_cast<const float*>(ws + p.off_rayseed);
    a.loss_sums = sb.loss_sums; a.loss_partials = sb.loss_partials; a.n_partials = sb.n_partials;
    a.sums_out = sb.sums_out; a.loss_out = sb.loss_out; a.counter = sb.counter;
    // (hash encoding: k_hash_mlp_bwd writes the positions k_hash_grad needs into a.hash_xyz itself)
  } else {
    sb.xyz_out = a.hash_xyz;                 // positions for the table-gradient kernel (hash encoding)
    a.hash_xyz_ready = a.hash_xyz != nullptr;
    e = ngm_launch_stash_bwd(sb, st);
    if (e) return fail(e, "render_bwd: unsupported geometry mode");
    e = check_launch("ngm_stash_bwd");
    if (e) return e;
  }
  if (neus && grads->neus_sd)
    ngm_launch_neus_sd_grad(sb.d_isd_rays, rays->F, rays->R, params->neus_sd, params->neus_sd_stride, params->field_index,
                            grads->neus_sd, st);
  e = prep_lattice_grad(fcfg, grads, rays->F, a, st);
  if (e) return e;
  e = launch_bwd_any(a, a.blocks_per_field * a.F, st);
  if (e == NGM_E_UNSUPPORTED && fuse) {
    // the fused kernel declined after all (its own LDS / shape checks): composite in k_stash_bwd, then any MLP backward
    fuse = false;
    a = a_plain;
    sb.xyz_out = a.hash_xyz;
    a.hash_xyz_ready = a.hash_xyz != nullptr;
    e = ngm_launch_stash_bwd(sb, st);
    if (e) return fail(e, "render_bwd: unsupported geometry mode");
    e = check_launch("ngm_stash_bwd");
    if (e) return e;
    e = prep_lattice_grad(fcfg, grads, rays->F, a, st);
    if (e) return e;
    e = launch_bwd_any(a, a.blocks_per_field * a.F, st);
  }
  if (e) return fail(e, "render_bwd: no kernel for this (D,H,L)");
  e = check_launch("ngm_field_bwd");
  if (e) return e;
  if (fcfg->encoding == NGM_ENC_TRIPLANE) { ngm_launch_tri_finish(a, st); e = check_launch("ngm_tri_finish"); if (e) return e; }
  GradReduceArgs g;
  memset(&g.adam, 0, sizeof(g.adam));
  if (adam) g.adam = *adam;
  g.fc = *fcfg; g.gr = *grads; g.F = rays->F; g.blocks_per_field = a.blocks_per_field; g.partials = a.partials; g.p_pad = a.p_pad;
  bool mlp_reduced = false;
  if (fcfg->encoding == NGM_ENC_PERMUTO) {
    if (lattice_adam) a.lattice_adam = *lattice_adam;
    // the MLP's reduction + Adam rides along in the table-gradient launch (both depend on the MLP backward only)
    e = ngm_launch_hash_grad(a, st, lattice_adam_applied, &g, &mlp_reduced);
    if (e == NGM_E_INVALID) return fail(e, "render_bwd_adam: adam tensors do not match the parameter segments");
    if (e) return fail(e, "permutohedral backward: hash table too large for the LDS-staged scatter");
    e = check_launch("ngm_hash_grad");
    if (e) return e;
  }
  if (mlp_reduced) return NGM_OK;
  if (ngm_launch_grad_reduce(g, st)) return fail(NGM_E_INVALID, "render_bwd_adam: adam tensors do not match the parameter segments");
  return check_launch("ngm_grad_reduce");
}

int ngm_render_bwd(const ngm_field_cfg* fcfg, const ngm_render_cfg* rcfg, const ngm_params* params, const ngm_rays* rays,
                   const ngm_targets* targets, const ngm_prediction* pred, const float* loss_sums, const ngm_grads* grads,
                   float* loss_out, void* workspace, int64_t workspace_bytes, void* stream) {
  int e = check_render(fcfg, rcfg, params, rays);
  if (e) return e;
  if (!targets || !targets->rgbds || !targets->depth_mask || !pred || !pred->rgbds || !pred->term_probs || !grads)
    return fail(NGM_E_INVALID, "render_bwd: NULL argument");
  StashBwdArgs sb;
  memset(&sb, 0, sizeof(sb));
  sb.seed_mode = 0; sb.tg = *targets; sb.pred = *pred; sb.loss_sums = loss_sums;
  sb.loss_out = loss_out;      // written by the compositing-backward kernel (no separate launch)
  return render_bwd_common(fcfg, rcfg, params, rays, sb, grads, workspace, workspace_bytes, (hipStream_t)stream);
}

int ngm_render_bwd_adam(const ngm_field_cfg* fcfg, const ngm_render_cfg* rcfg, const ngm_params* params, const ngm_rays* rays,
                        const ngm_targets* targets, const ngm_prediction* pred, const float* loss_sums, const ngm_grads* grads,
                        const ngm_adam_tensor* mlp_tensors, int32_t num_mlp_tensors, const ngm_adam_tensor* lattice_tensor,
                        const int64_t* field_index, int64_t step, int64_t* step_dev, float lr, float beta1, float beta2,
                        float eps, float weight_decay, float* loss_out, void* workspace, int64_t workspace_bytes, void* stream) {
  int e = check_render(fcfg, rcfg, params, rays);
  if (e) return e;
  if (!targets || !targets->rgbds || !targets->depth_mask || !pred || !pred->rgbds || !pred->term_probs || !grads ||
      !mlp_tensors || num_mlp_tensors < 1 || (step < 1 && !step_dev))
    return fail(NGM_E_INVALID, "render_bwd_adam: bad argument");
  if ((fcfg->encoding == NGM_ENC_PERMUTO) != (lattice_tensor != nullptr))
    return fail(NGM_E_INVALID, "render_bwd_adam: lattice_tensor goes with the permutohedral encoding");
  StashBwdArgs sb;
  memset(&sb, 0, sizeof(sb));
  sb.seed_mode = 0; sb.tg = *targets; sb.pred = *pred; sb.loss_sums = loss_sums;
  sb.loss_out = loss_out;
  GradAdam ad;
  ad.tensors = mlp_tensors; ad.num = num_mlp_tensors; ad.field_index = field_index; ad.step = step; ad.step_dev = step_dev;
  ad.lr = lr; ad.beta1 = beta1; ad.beta2 = beta2; ad.eps = eps; ad.wd = weight_decay;
  if (lattice_tensor && (!lattice_tensor->param || !lattice_tensor->exp_avg || !lattice_tensor->exp_avg_sq || !lattice_tensor->grad))
    return fail(NGM_E_INVALID, "render_bwd_adam: NULL lattice tensor");
  GradAdam lad = ad;
  lad.tensors = lattice_tensor; lad.num = lattice_tensor ? 1 : 0;
  bool lattice_done = false;
  e = render_bwd_common(fcfg, rcfg, params, rays, sb, grads, workspace, workspace_bytes, (hipStream_t)stream, &ad,
                        lattice_tensor ? &lad : nullptr, &lattice_done);
  if (e) return e;
  if (lattice_tensor && !lattice_done) {   // unaligned tables: k_hash_reduce left the update to a plain Adam launch
    ngm_launch_adam_multi(lattice_tensor, 1, field_index, rays->F, step, step_dev, lr, beta1, beta2, eps, weight_decay,
                          nullptr, nullptr, (hipStream_t)stream);
    e = check_launch("ngm_adam_sparse_multi");
  }
  return e;
}

int ngm_render_bwd_seeded(const ngm_field_cfg* fcfg, const ngm_render_cfg* rcfg, const ngm_params* params,
                          const ngm_rays* rays, const float* d_rgbds, const float* d_term, const float* d_geom_samples,
                          const ngm_grads* grads, void* workspace, int64_t workspace_bytes, void* stream) {
  int e = check_render(fcfg, rcfg, params, rays);
  if (e) return e;
  if (!d_rgbds || !grads) return fail(NGM_E_INVALID, "render_bwd_seeded: NULL argument");
  StashBwdArgs sb;
  memset(&sb, 0, sizeof(sb));
  sb.seed_mode = 1; sb.d_rgbds = d_rgbds; sb.d_term = d_term; sb.d_geom_samples = d_geom_samples;
  return render_bwd_common(fcfg, rcfg, params, rays, sb, grads, workspace, workspace_bytes, (hipStream_t)stream);
}

int ngm_render_bwd_seeded_vars(const ngm_field_cfg* fcfg, const ngm_render_cfg* rcfg, const ngm_params* params,
                               const ngm_rays* rays, const ngm_prediction* pred, const float* d_rgbds, const float* d_color_vars,
                               const float* d_depth_vars, const float* d_term, const float* d_geom_samples,
                               const ngm_grads* grads, void* workspace, int64_t workspace_bytes, void* stream) {
  int e = check_render(fcfg, rcfg, params, rays);
  if (e) return e;
  if (!d_rgbds || !grads) return fail(NGM_E_INVALID, "render_bwd_seeded_vars: NULL argument");
  if ((d_color_vars || d_depth_vars) && (!pred || !pred->rgbds || !pred->term_probs))
    return fail(NGM_E_INVALID, "render_bwd_seeded_vars: seeds on the variances need the forward's pred.rgbds and pred.term_probs");
  StashBwdArgs sb;
  memset(&sb, 0, sizeof(sb));
  sb.seed_mode = 1; sb.d_rgbds = d_rgbds; sb.d_term = d_term; sb.d_geom_samples = d_geom_samples;
  sb.d_cvars = d_color_vars; sb.d_dvars = d_depth_vars;
  if (pred) sb.pred = *pred;
  return render_bwd_common(fcfg, rcfg, params, rays, sb, grads, workspace, workspace_bytes, (hipStream_t)stream);
}

int ngm_render_read_samples(const ngm_field_cfg* fcfg, const ngm_render_cfg* rcfg, int32_t F, int32_t R, int32_t S,
                            const void* workspace, float* geoms, float* dists, void* stream) {
  if (check_field_cfg(fcfg) || !rcfg || !workspace) return fail(NGM_E_INVALID, "render_read_samples: bad argument");
  // S tells which plan the forward ran with (it plans from rays->gt): the stash offsets and the copy size depend on it
  const bool guided = S != rcfg->num_samples_coarse;
  if (S != rcfg->num_samples_coarse + (guided ? rcfg->num_samples_guided : 0))
    return fail(NGM_E_INVALID, "render_read_samples: S is neither S_c nor S_c + S_g of this render cfg");
  const RenderPlan p = plan_render(fcfg, rcfg, F, R, guided, true);
  const char* ws = reinterpret_cast<const char*>(align_up((int64_t)workspace, 256));
  ngm_launch_read_stash(reinterpret_cast<const float4*>(ws + p.off_stashA), reinterpret_cast<const float2*>(ws + p.off_stashB),
                        (int64_t)F * R * p.S, geoms, dists, (hipStream_t)stream);
  return check_launch("ngm_render_read_samples");
}

// ------------------------------------------------------------------------------------------------
int ngm_adam_sparse(float* param, float* exp_avg, float* exp_avg_sq, int64_t stride, const float* grad,
                    int64_t grad_stride, const int64_t* field_index, int32_t F, int64_t numel_per_field, int64_t step,
                    float lr, float beta1, float beta2, float eps, float weight_decay, void* stream) {
  if (!param || !exp_avg || !exp_avg_sq || !grad || F < 1 || numel_per_field < 1 || step < 1)
    return fail(NGM_E_INVALID, "ngm_adam_sparse: bad argument");
  ngm_launch_adam(param, exp_avg, exp_avg_sq, stride, grad, grad_stride, field_index, F, numel_per_field, step, lr, beta1,
                  beta2, eps, weight_decay, (hipStream_t)stream);
  return check_launch("ngm_adam_sparse");
}

int ngm_adam_sparse_multi(const ngm_adam_tensor* tensors, int32_t num_tensors, const int64_t* field_index, int32_t F,
                          int64_t step, int64_t* step_dev, float lr, float beta1, float beta2, float eps,
                          float weight_decay, int32_t advance_step_dev, uint64_t* advance_philox_offset_dev, void* stream) {
  if (!tensors || num_tensors < 1 || num_tensors > 2 * (NGM_MAX_LAYERS + 1) + 2 || F < 1 || (step < 1 && !step_dev))
    return fail(NGM_E_INVALID, "ngm_adam_sparse_multi: bad argument");
  for (int i = 0; i < num_tensors; ++i)
    if (!tensors[i].param || !tensors[i].exp_avg || !tensors[i].exp_avg_sq || !tensors[i].grad || tensors[i].numel < 1)
      return fail(NGM_E_INVALID, "ngm_adam_sparse_multi: NULL tensor");
  ngm_launch_adam_multi(tensors, num_tensors, field_index, F, step, step_dev, lr, beta1, beta2, eps, weight_decay,
                        (advance_step_dev && step_dev) ? step_dev : nullptr, advance_philox_offset_dev, (hipStream_t)stream);
  return check_launch("ngm_adam_sparse_multi");
}

int ngm_step_advance(int64_t* step_dev, uint64_t* philox_offset_dev, void* stream) {
  ngm_launch_step_advance(step_dev, philox_offset_dev, (hipStream_t)stream);
  return check_launch("ngm_step_advance");
}

int64_t ngm_field_eval_knn_workspace(int32_t num_fields, int64_t P, int32_t num_knn) {
  const int K = num_knn < num_fields ? num_knn : num_fields;
  if (num_fields < 1 || P < 0 || K < 1) return NGM_E_INVALID;
  return ngm_knn_workspace_bytes(num_fields, P, K);
}

int ngm_field_eval_knn(const ngm_field_cfg* fcfg, const ngm_params* params, int32_t num_fields, int64_t P,
                       const float* points, const float* field_pos, const float* field_quat, int32_t num_knn,
                       float distance_factor, float outside_value, float mask_radius, float* out, void* workspace,
                       int64_t workspace_bytes, void* stream) {
  int e = check_field_cfg(fcfg);
  if (e) return e;
  e = check_params(fcfg, params);
  if (e) return e;
  if (!points || !field_pos || !field_quat || !out || num_fields < 1 || P < 0) return fail(NGM_E_INVALID, "ngm_field_eval_knn: bad argument");
  if (P == 0) return NGM_OK;
  const int K = num_knn < num_fields ? num_knn : num_fields;
  if (K < 1 || K > 16) return fail(NGM_E_UNSUPPORTED, "ngm_field_eval_knn: K must be in [1,16]");
  e = ngm_launch_knn(fcfg, params, num_fields, P, points, field_pos, field_quat, K, distance_factor, outside_value,
                     mask_radius > 0.f ? mask_radius : fcfg->field_radius, out, workspace, workspace_bytes, (hipStream_t)stream);
  if (e == NGM_E_WORKSPACE) return fail(e, "ngm_field_eval_knn: workspace too small");
  if (e) return fail(e, "ngm_field_eval_knn: not available for this configuration");
  return check_launch("ngm_field_eval_knn");
}

int64_t ngm_render_eval_knn_workspace(const ngm_render_cfg* rcfg, int32_t num_fields, int32_t ray_block, int32_t num_knn) {
  const int K = num_knn < num_fields ? num_knn : num_fields;
  if (!rcfg || num_fields < 1 || ray_block < 1 || K < 1 || rcfg->num_samples_coarse < 1) return NGM_E_INVALID;
  return ngm_knn_render_workspace_bytes(num_fields, ray_block, rcfg->num_samples_coarse, K);
}

int ngm_render_eval_knn(const ngm_field_cfg* fcfg, const ngm_render_cfg* rcfg, const ngm_params* params, int32_t num_fields,
                        const float* field_pos, const float* field_quat, const ngm_rays* rays, int32_t num_knn,
                        float distance_factor, float outside_value, float mask_radius, int32_t ray_block,
                        const ngm_prediction* pred, void* workspace, int64_t workspace_bytes, void* stream) {
  int e = check_field_cfg(fcfg);
  if (e) return e;
  e = check_params(fcfg, params);
  if (e) return e;
  if (!rcfg || !rays || !pred || !field_pos || !field_quat || num_fields < 1 || ray_block < 1 || !rays->ijs || !rays->c2ws ||
      rays->F < 0 || rays->R < 0)
    return fail(NGM_E_INVALID, "ngm_render_eval_knn: bad argument");
  if ((int64_t)rays->F * rays->R == 0) return NGM_OK;
  const int K = num_knn < num_fields ? num_knn : num_fields;
  if (K < 1 || K > 8) return fail(NGM_E_UNSUPPORTED, "ngm_render_eval_knn: K must be in [1,8] (the blend inside the quadrature is compiled for up to 8 neighbours; K = 9..16: the staged entry points, ngm_sample_rays_world -> ngm_field_eval_knn -> ngm_composite_fwd_packed)");
  e = ngm_launch_render_eval_knn(fcfg, rcfg, params, num_fields, field_pos, field_quat, rays, K, distance_factor, outside_value,
                                 mask_radius > 0.f ? mask_radius : fcfg->field_radius, ray_block, pred, workspace,
                                 workspace_bytes, (hipStream_t)stream);
  if (e == NGM_E_WORKSPACE) return fail(e, "ngm_render_eval_knn: workspace too small");
  if (e) return fail(e, "ngm_render_eval_knn: not available for this configuration (samples per ray <= 1024, ray_block * samples * K < 2^31)");
  return check_launch("ngm_render_eval_knn");
}

// ------------------------------------------------------------------------------------------------
// mesh extraction
// ------------------------------------------------------------------------------------------------
int64_t ngm_marching_cubes_workspace(int32_t nx, int32_t ny, int32_t nz) {
  if (nx < 2 || ny < 2 || nz < 2 || 3 * (int64_t)nx * ny * nz + 1 > 0x7fffffff) return NGM_E_INVALID;
  return ngm_mc_workspace_bytes(nx, ny, nz);
}
static int mc_fail(int e, const char* what) {
  if (e == NGM_E_WORKSPACE) return fail(e, "marching cubes: workspace too small");
  if (e == NGM_E_UNSUPPORTED) return fail(e, "marching cubes: grid too large (3 * nx * ny * nz must fit 31 bits)");
  if (e == NGM_E_HIP) return fail(e, "marching cubes: HIP error");
  if (e) return fail(e, what);
  return check_launch(what);
}
int ngm_marching_cubes_count(const float* volume, int32_t nx, int32_t ny, int32_t nz, float isolevel, int64_t* counts,
                             void* workspace, int64_t workspace_bytes, void* stream) {
  return mc_fail(ngm_launch_mc_count(volume, nx, ny, nz, isolevel, counts, workspace, workspace_bytes, (hipStream_t)stream),
                 "ngm_marching_cubes_count: bad argument");
}
int ngm_marching_cubes_emit(const float* volume, int32_t nx, int32_t ny, int32_t nz, float isolevel, float* verts,
                            int64_t max_verts, int64_t* faces, int64_t max_faces, void* workspace, int64_t workspace_bytes,
                            void* stream) {
  return mc_fail(ngm_launch_mc_emit(volume, nx, ny, nz, isolevel, verts, max_verts, faces, max_faces, workspace,
                                    workspace_bytes, (hipStream_t)stream), "ngm_marching_cubes_emit: bad argument");
}
int ngm_marching_cubes_tables(int8_t* tri_table, int32_t* tri_count) {
  const int e = ngm_mc_copy_tables(tri_table, tri_count);
  return e ? fail(e, "marching cubes: table derivation failed") : NGM_OK;
}

// ------------------------------------------------------------------------------------------------
// one-shot loss exchange between the ranks of a node (ngm_peer.hip)
// ------------------------------------------------------------------------------------------------
static int hip_fail(hipError_t e, const char* what) {
  snprintf(g_err, sizeof(g_err), "%s: %s", what, hipGetErrorString(e));
  (void)hipGetLastError();
  return NGM_E_HIP;
}
int64_t ngm_peer_mailbox_bytes(void) { return 2 * NGM_MAX_PEERS * 16 * 8; }
int ngm_peer_alloc(int64_t bytes, void** ptr) {
  if (!ptr || bytes <= 0) return fail(NGM_E_INVALID, "ngm_peer_alloc: bad argument");
  const int e = ngm_peer_alloc_impl(bytes, ptr);
  return e ? hip_fail((hipError_t)e, "ngm_peer_alloc") : NGM_OK;
}
int ngm_peer_free(void* ptr) {
  const hipError_t e = hipFree(ptr);
  return e == hipSuccess ? NGM_OK : hip_fail(e, "ngm_peer_free");
}
int ngm_ipc_export(void* ptr, unsigned char handle[64]) {
  static_assert(sizeof(hipIpcMemHandle_t) == 64, "hipIpcMemHandle_t is 64 bytes");
  if (!ptr || !handle) return fail(NGM_E_INVALID, "ngm_ipc_export: NULL");
  hipIpcMemHandle_t h;
  const hipError_t e = hipIpcGetMemHandle(&h, ptr);
  if (e != hipSuccess) return hip_fail(e, "hipIpcGetMemHandle");
  memcpy(handle, &h, 64);
  return NGM_OK;
}
int ngm_ipc_open(const unsigned char handle[64], void** ptr) {
  if (!ptr || !handle) return fail(NGM_E_INVALID, "ngm_ipc_open: NULL");
  hipIpcMemHandle_t h;
  memcpy(&h, handle, 64);
  const hipError_t e = hipIpcOpenMemHandle(ptr, h, hipIpcMemLazyEnablePeerAccess);
  return e == hipSuccess ? NGM_OK : hip_fail(e, "hipIpcOpenMemHandle");
}
int ngm_ipc_close(void* ptr) {
  const hipError_t e = hipIpcCloseMemHandle(ptr);
  return e == hipSuccess ? NGM_OK : hip_fail(e, "hipIpcCloseMemHandle");
}
int ngm_debug_last_stash_mode(void) { return g_last_stash_mode; }
int ngm_debug_stash_mode(int mode) {
  const int prev = g_stash_override;
  if (mode >= 0 && mode <= 1) g_stash_override = mode;
  else if (mode == -2) g_stash_override = -1;
  return prev;
}

double ngm_peer_set_timeout(double seconds) { return ngm_peer_set_timeout_impl(seconds); }

int ngm_loss_exchange(const ngm_peer_exchange* px, float* loss_sums, void* stream) {
  if (!px || !loss_sums || !px->seq || !px->status) return fail(NGM_E_INVALID, "ngm_loss_exchange: NULL");
  if (px->world < 1 || px->world > NGM_MAX_PEERS || px->rank < 0 || px->rank >= px->world)
    return fail(NGM_E_INVALID, "ngm_loss_exchange: 1 <= world <= 8, 0 <= rank < world");
  for (int p = 0; p < px->world; ++p)
    if (!px->mailbox[p]) return fail(NGM_E_INVALID, "ngm_loss_exchange: mailbox of a rank not mapped");
  ngm_launch_loss_exchange(*px, loss_sums, (hipStream_t)stream);
  return check_launch("ngm_loss_exchange");
}

}  // extern "C"
