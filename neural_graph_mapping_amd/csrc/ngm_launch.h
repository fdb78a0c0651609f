// Kernel argument records + host launcher prototypes (internal; the public surface is include/ngm_hip.h).
#pragma once
#include <hip/hip_runtime.h>

#include "../../include/ngm_hip.h"

#ifndef NGM_BLOCK
#define NGM_WAVE 64
#define NGM_BLOCK 256
#define NGM_WAVES_PER_BLOCK 4
#endif

struct PointsFwdArgs {
  ngm_field_cfg fc;
  ngm_params pr;
  int F;
  int64_t P;
  int64_t per_block;      // samples per workgroup (multiple of 256)
  const float* points;    // (F,P,3)
  const float* pos;       // (F,3) or NULL
  const float* quat;      // (F,4) or NULL
  float* out;             // (F,P,4)
  // training forward (ngm_field_eval_fwd_train): the hidden activations of every sample, in the tiled stash layout of the fused
  // training step (ngm_field.h ActStash; global sample index = f * P + p), read by k_field_bwd_b3 in point mode
  float* act;             // NULL: nothing is stashed
  int64_t act_layer_stride;   // floats
};

struct RenderFwdArgs {
  ngm_field_cfg fc;
  ngm_params pr;
  ngm_render_cfg rc;
  ngm_rays rays;
  ngm_targets tg;
  ngm_prediction pred;
  int has_targets;
  int S;                  // samples per ray actually drawn (S_c + S_g when gt given)
  int rays_per_block;
  int waves_per_block;    // 8 (two waves per SIMD) when the per-wave LDS planes fit, else 4
  int maxs;               // samples per wave batch (size of the per-wave LDS sample planes)
  float* raytab;          // (F*R, 8): o_local(3), d_local(3), dz_cam, gt     (train) or NULL
  float4* stashA;         // (F*R*S): colour(3), geometry                      (train) or NULL
  float2* stashB;         // (F*R*S): t, T_exclusive
  float* loss_partials;   // (blocks, 16)
  float* rayseed;         // (F*R, 8) train + targets: per-ray loss derivatives before the global normalisers -- d photometric /
                          // d colour (3), d depth loss / d depth, d termination loss / d termination probability, 3 x unused --
                          // read by the fused compositing backward (k_field_bwd_b3, FieldBwdArgs::fused_comp)
  float* act;             // hidden-activation stash [L][F*R*S][64] (train, 64-wide hidden layers) or NULL
  int64_t act_layer_stride;  // floats
  int act_layers;         // how many of the hidden layers' outputs are stashed (half stash: 1 of 2; 0 = all)
  const float* neus_sd;   // neus: "_neus_sd" (N,) rows like the other parameters (field_index), else NULL
  int64_t neus_sd_stride;
  unsigned long long* debug_cycles;   // optional (NGM_PHASE_TIMING): per-phase shader-clock cycles of one wave
};

// backward of the field MLP for flat samples of each field
// optional: apply sparse Adam to the reduced gradient in the same kernel (one adam tensor per gradient segment,
// in the segment order enc_w (Fourier only), w_0, b_0, ..., w_L, b_L)
struct GradAdam {
  const ngm_adam_tensor* tensors;    // host array, num == number of segments; NULL: reduction only
  int num;
  const int64_t* field_index;
  int64_t step; const int64_t* step_dev;
  float lr, beta1, beta2, eps, wd;
};
struct FieldBwdArgs {
  ngm_field_cfg fc;
  ngm_params pr;
  int F;
  int64_t P;              // samples per field
  int64_t per_block;      // samples per workgroup (multiple of 128)
  int blocks_per_field;
  // sample source: explicit points (points != NULL) or rays (raytab + stashB.t)
  const float* points;    // (F,P,3) world/local
  const float* pos;
  const float* quat;
  const float* raytab;    // (F*R, 8)
  const float2* stashB;   // (F*R*S)
  int S;                  // samples per ray (ray mode)
  const float4* d_out;    // (F,P) float4: dL/d(r,g,b,geometry) raw MLP outputs
  float* partials;        // (F*blocks_per_field, P_pad) per-workgroup gradient partial sums
  int64_t p_pad;          // padded parameter count per field (floats)
  float* lattice_grad;    // permutohedral: (F, L, T, 2) gradient table, stride between fields
  int64_t lattice_grad_stride;
  float2* hash_dE;        // permutohedral: [L][F*P] dL/d(level features) per sample (level-major: coalesced)
  float4* hash_xyz;       // permutohedral: [F*P] scaled field-local sample positions
  float* hash_part;       // permutohedral: [F][L][8][2T] partial gradient tables
  int hash_xyz_ready;     // ray mode: k_stash_bwd already wrote hash_xyz (one fmaf triple per sample where ray entry and distance
                          // are in registers anyway), the MLP backward must not
  long long* tri_acc;     // triplane: [F][3 C res res] Q23.40 gradient accumulators (zeroed by the API before the launch)
  int64_t tri_numel;      // 3 C res res
  float* planes_grad;     // triplane: caller's (F, 3, C, res, res) gradient, stride between fields
  int64_t planes_grad_stride;
  const float* act;       // hidden-activation stash written by the forward (ray mode) or NULL = recompute
  int64_t act_layer_stride;
  int act_half;           // two hidden layers, split path: the forward stashed layer 0's output ONLY (k_field_bwd_b3<.., HS> recomputes
                          // the output layer's input); no other kernel reads such a stash
  unsigned long long* debug_cycles;   // optional (NGM_PHASE_TIMING): per-phase s_memtime cycles of wave 0 / block 0
  GradAdam lattice_adam;              // optional (tensors != NULL, num == 1): k_hash_reduce applies Adam to the hash tables
  // Fused compositing backward (k_field_bwd_b3<FC> and k_hash_mlp_bwd<FC>, pointwise geometry modes, loss seeds written by the
  // forward): d_out then holds the forward's (colour, geometry) stash untouched, and the kernel does what k_stash_bwd does --
  // per-ray reverse scan, loss derivatives, the loss bookkeeping of its block 0 -- on the tile it is about to back-propagate.
  // Every wave walks a contiguous range of tiles back to front; per_block is a plain multiple of 128 (k_hash_mlp_bwd: 256),
  // the ray a range cuts is handled by comp_suffix_beyond (ngm_bwd16.h).
  int fused_comp;
  ngm_render_cfg rc;
  const float* rayseed;               // (F*R, 8) from the forward
  const float* loss_sums;             // global (all-reduced) sums, or NULL -> loss_partials
  const float* loss_partials; int n_partials;
  float* sums_out; float* loss_out; unsigned long long* counter;   // as StashBwdArgs
};
bool ngm_field_bwd_b3_applies(const FieldBwdArgs& a);   // would ngm_launch_field_bwd_b3 take this problem
bool ngm_hash_mlp_bwd_applies(const FieldBwdArgs& a);   // would ngm_launch_hash_mlp_bwd (given positions, or fused_comp)
struct GradReduceArgs;
// mlp_reduce (optional): the MLP's gradient reduction (+ Adam) to run as extra workgroups of the same launch; *mlp_reduced tells
// the caller that it did (no ngm_launch_grad_reduce needed then)
int ngm_launch_hash_grad(const FieldBwdArgs& a, hipStream_t st, bool* adam_applied = nullptr, const GradReduceArgs* mlp_reduce = nullptr,
                         bool* mlp_reduced = nullptr);

struct GradReduceArgs {
  ngm_field_cfg fc;
  ngm_grads gr;
  int F;
  int blocks_per_field;
  const float* partials;
  int64_t p_pad;
  GradAdam adam;
};

// composite (quadrature) standalone + stash variants
struct CompositeArgs {
  ngm_render_cfg rc;
  int64_t N;              // rays
  int S;
  // separate-tensor source (standalone API)
  const float* colors; const float* geoms; const float* dists; const float* depths; const float* isds;
  // packed source (eval path): (N,S,4) field outputs + camera-frame points; rgbd output (N,4)
  const float4* out4; const float* pcam; float* rgbd;
  // packed source fused with the kNN blend (ngm_render_eval_knn): the pair records instead of out4, the rays' camera-frame
  // directions (N,3) instead of pcam (depth = -direction.z * distance)
  const int* pair_field; const float* pair_w; const float4* pair_out; int pair_K; float outside_value; const float* ray_dir;
  // outputs fwd
  float* C; float* D; float* Cv; float* Dv; float* term; float* weights;
  // bwd seeds / outputs
  const float* dC; const float* dD; const float* dterm;
  float* d_colors; float* d_geoms; float* d_isds;
};

// backward of compositing on the saved per-sample stash of the fused forward; overwrites stashA with
// dL/d(raw MLP outputs)
struct StashBwdArgs {
  ngm_render_cfg rc;
  int F, R, S;
  float4* stashA;
  const float2* stashB;
  const float* raytab;
  // seed mode 0: losses (targets + global sums), 1: explicit seeds
  int seed_mode;
  ngm_targets tg;
  ngm_prediction pred;
  const float* loss_sums;     // global (all-reduced) sums
  const float* d_rgbds;       // (F,R,4)
  const float* d_term;        // (F,R)
  const float* d_geom_samples;// (F,R,S) or NULL
  const float* d_cvars;       // seed mode 1, optional: dL/d(color_vars) (F,R,3) and dL/d(depth_vars) (F,R); `pred` then carries the
  const float* d_dvars;       // forward's rgbds / term_probs (the means and the weight sum the variances are taken around)
  float* loss_out;            // (8) loss scalars from loss_sums (seed mode 0) or NULL
  // deferred loss reduction (single GPU: nothing happens between forward and backward): every workgroup sums the
  // forward's per-workgroup partials itself, in the fixed order of k_loss_reduce -> one launch less per step
  // neus (rm.py:641-644, 753-758): per-field inverse standard deviations, a separate gradient buffer (the kernel reads
  // its neighbours' saved geometry, so it cannot overwrite stashA in place) and per-ray d loss / d isd
  const float* neus_sd; int64_t neus_sd_stride; const int64_t* field_index;
  float4* d_out;              // (F*R*S) dL/d(raw MLP outputs) when not written over stashA (neus), else NULL
  float* d_isd_rays;          // (F*R), zeroed by the launcher
  const float* loss_partials; // (n_partials, 16) or NULL -> loss_sums holds the (all-reduced) sums
  int n_partials;
  float* sums_out;            // (16) optional copy of the reduced sums
  unsigned long long* counter;// optional iteration counter to advance (what k_loss_reduce does otherwise)
  float4* xyz_out;            // (F*R*S) optional: scaled field-local sample positions for k_hash_grad (permutohedral encoding)
};

// which arithmetic the last launch of each forward-type kernel resolved to (ngm_debug_last_matmul): 0 = the fused render
// forward, 1 = the point evaluation, 2 = the kNN evaluation; values NGM_MATMUL_F32 / NGM_MATMUL_BF16X3, -1 = none yet
extern int g_ngm_last_matmul[3];
extern int g_ngm_last_fwd_one_tile;      // the last fused render forward ran the one-tile wave step instance (k_render_fwd<.., HALF>)
int ngm_launch_points_fwd(const PointsFwdArgs& a, int blocks, hipStream_t st);
int64_t ngm_encode_bwd_fourier_scratch(int F, int64_t P);
int ngm_launch_encode_bwd_fourier(const ngm_field_cfg& fc, const ngm_params& pr, int F, int64_t P, const float* points,
                                  const float* pos, const float* quat, const float* d_enc, float* grad, int64_t grad_stride,
                                  float* scratch, hipStream_t st);
int ngm_launch_encode_bwd_prep_hash(const ngm_field_cfg& fc, const ngm_params& pr, int F, int64_t P, const float* points,
                                    const float* pos, const float* quat, const float* d_enc, float2* dE, float4* xyz,
                                    hipStream_t st);
int ngm_launch_encode_points(const ngm_field_cfg& fc, const ngm_params& pr, int F, int64_t P, const float* points, const float* pos,
                             const float* quat, float* out, hipStream_t st);   // the positional encoding alone -> (F, P, dim_enc)
int ngm_launch_render_fwd(const RenderFwdArgs& a, int blocks, hipStream_t st);
int ngm_launch_field_bwd(const FieldBwdArgs& a, int blocks, hipStream_t st);
int ngm_launch_tri_finish(const FieldBwdArgs& a, hipStream_t st);
int ngm_launch_field_bwd_b3(const FieldBwdArgs& a, int blocks, hipStream_t st);   // 32-sample tiles on the bf16 matrix pipe (three-way split), activation stash
int ngm_launch_hash_mlp_bwd(const FieldBwdArgs& a, int blocks, hipStream_t st);          // hash encoding + 1 x 32 MLP (the reference's default network), bf16 split, encoding stash
int ngm_launch_field_bwd16s(const FieldBwdArgs& a, int blocks, hipStream_t st);  // 16-sample tiles, activations from the forward's stash
int ngm_launch_field_bwd16(const FieldBwdArgs& a, int blocks, hipStream_t st);   // 16-sample tiles, 8 waves; NGM_E_UNSUPPORTED -> fall back
int ngm_launch_grad_reduce(const GradReduceArgs& a, hipStream_t st);
int ngm_launch_composite_fwd(const CompositeArgs& a, hipStream_t st);
int ngm_launch_composite_bwd(const CompositeArgs& a, hipStream_t st);
int ngm_launch_stash_bwd(const StashBwdArgs& a, hipStream_t st);
int ngm_launch_loss_exchange(const ngm_peer_exchange& px, float* sums, hipStream_t st);   // ngm_peer.hip
double ngm_peer_set_timeout_impl(double seconds);
int ngm_peer_alloc_impl(int64_t bytes, void** out);

int ngm_launch_target_visibility(const ngm_keyframes& kf, int F, const float* field_pos, int num_offsets, const float* offsets,
                                 float radius, uint8_t* kf_mask, float* bbox, hipStream_t st);
int ngm_launch_target_sv_intersect(int F, int64_t N, const float* pos_c, const float* points, float radius, uint8_t* hit, hipStream_t st);
int ngm_launch_target_sv_rays(int F, int R, const float* pos_c, float radius, const int64_t* pts_ijs, const int64_t* segments,
                              const float* image, int height, int width, float fx, float fy, float cx, float cy, const ngm_target_out& o,
                              hipStream_t st);
int ngm_launch_target_rays(const ngm_keyframes& kf, int F, int R, const float* field_pos, float radius, const float* bbox,
                           const int64_t* frame_cids, const float* u_xy, const ngm_target_out& o, hipStream_t st);

// flat parameter vector layout of one field: [enc_w][w0][b0]...[wL][bL]
__host__ __device__ static inline int64_t ngm_param_offsets(const ngm_field_cfg* fc, int64_t* enc_off, int64_t* w_off, int64_t* b_off) {
  int64_t o = 0;
  *enc_off = 0;
  if (fc->encoding == NGM_ENC_FOURIER) o += (int64_t)(fc->raw_coords ? fc->dim_enc - 3 : fc->dim_enc) * 3;
  for (int l = 0; l <= fc->num_layers; ++l) {
    const int din = (l == 0) ? fc->dim_enc : fc->dim_hidden + (fc->skip_mode == NGM_SKIP_CONCAT ? fc->dim_enc : 0);
    const int dout = (l == fc->num_layers) ? fc->dim_out : fc->dim_hidden;
    w_off[l] = o; o += (int64_t)din * dout;
    b_off[l] = o; o += dout;
  }
  return o;
}

// ---- measurement hooks: HIP events on the launch stream around selected kernels ----------------
struct NgmProfScope {
  int id; hipStream_t st; bool on;
  NgmProfScope(int kernel_id, hipStream_t stream);
  ~NgmProfScope();
};
