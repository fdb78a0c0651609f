/*
 * ngm_hip.h -- C ABI of the MI355X-native (gfx950) render/train hot path of Neural Graph Mapping.
 *
 * The reference (KTH-RPL/neural_graph_mapping) is pure Python and has no FFI; its boundary is a
 * Python method contract.  Each entry point below replaces the reference interface cited next to
 * it (paths relative to /root/reference/src/neural_graph_mapping, rm.py = run_mapping.py).  The
 * reference-side binding (a ctypes stub a maintainer would add) is shown in INTEGRATION.md.
 *
 * Conventions
 *  - plain C, opaque device pointers + explicit sizes; no torch types, no allocation inside:
 *    the caller owns and pre-allocates every output and the workspace;
 *  - all floating point tensors are contiguous row-major fp32, indices int64 (as the reference);
 *  - every call is asynchronous on `stream` (a hipStream_t passed as void*; NULL = default stream);
 *  - return value: 0 = ok, negative = NGM_E_* (ngm_last_error() holds a message); never throws.
 *  - F = active fields in the batch, R = rays per field, S = S_c + S_g samples per ray,
 *    P = points per field, D = encoding width, H = hidden width, L = hidden layers.
 */
#ifndef NGM_HIP_H
#define NGM_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define NGM_ABI_VERSION 11 /* 11: ngm_field_eval_stash_bytes / ngm_field_eval_fwd_train / ngm_field_eval_bwd_stash (training forward of the point evaluation writes the activation stash, its backward is the fused step's kernel); 10: ngm_field_cfg.activation_stash / .hash_grad_atomics (per configuration, no process-wide switch), empty loss selections report NaN like the reference; 9: ngm_sample_rays_weighted; 8: ngm_peer_set_timeout (a time-out now also poisons the sums with NaN); 7: ngm_encode_bwd; 6: ngm_render_eval_knn; 5: ngm_encode_fwd, ngm_render_bwd_seeded_vars, *_nll loss modes (+ loss-sum slot 10), peer status bits */
#define NGM_MAX_LAYERS 4 /* hidden layers; +1 output layer */
#define NGM_NUM_LOSS_SUMS 16

enum ngm_status {
  NGM_OK = 0,
  NGM_E_INVALID = -1,     /* bad argument / inconsistent sizes          */
  NGM_E_UNSUPPORTED = -2, /* configuration has no compiled kernel       */
  NGM_E_WORKSPACE = -3,   /* workspace too small                        */
  NGM_E_HIP = -4          /* HIP runtime error (launch / device)        */
};

enum ngm_encoding { NGM_ENC_NONE = 0, NGM_ENC_FOURIER = 1, NGM_ENC_NERF = 2, NGM_ENC_PERMUTO = 3, NGM_ENC_TRIPLANE = 4 };
enum ngm_triplane_mode { NGM_TRI_SUM = 0, NGM_TRI_PRODUCT = 1, NGM_TRI_CONCAT = 2 };   /* positional_encodings.py:152-161 */
enum ngm_skip_mode { NGM_SKIP_NO = 0, NGM_SKIP_ADD = 1, NGM_SKIP_CONCAT = 2 }; /* models.py:159-180; rezero: the reference's constructor raises */
/* Arithmetic of the hidden layers' matrix products: forward kernels (fused render, point evaluation, kNN evaluation)
 * and the fused training step's MLP backward.
 * NGM_MATMUL_F32: exact-fp32 MFMA.  NGM_MATMUL_BF16X3: every fp32 operand is split exactly into three bf16
 * (hi + mid + lo) and the six leading cross products are summed in fp32 on the bf16 matrix pipe -- fp32-level accuracy
 * (dropped terms < 2^-23 relative), not narrower arithmetic; bitwise deterministic; compiled for 33..64-wide layers,
 * <= 2 hidden layers, Fourier / no encoding, skip_mode no.  As an explicit request ngm_render_fwd returns
 * NGM_E_UNSUPPORTED where it is not compiled or its weight planes (24 KB of LDS per layer) do not fit -- never a silent
 * fallback.  NGM_MATMUL_AUTO: the split wherever it is compiled and fits, exact-fp32 MFMA otherwise (resolved per batch
 * shape by the same plan in forward and backward).  The fused step's MLP backward follows the mode on its own terms:
 * F32 -> fp32 MFMA; BF16X3 / AUTO -> the split kernel where it is compiled (49..64-wide layers, 1-2 hidden layers,
 * Fourier / NeRF / no encoding, skip_mode no, activation stash present), fp32 MFMA otherwise. */
enum ngm_matmul_mode { NGM_MATMUL_F32 = 0, NGM_MATMUL_BF16X3 = 1, NGM_MATMUL_AUTO = 2 };
enum ngm_param_dtype { NGM_DT_F32 = 0, NGM_DT_BF16 = 1, NGM_DT_F16 = 2 };   /* storage type of the weights (ngm_params.dtype) */
enum ngm_scale_mode { NGM_SCALE_NO = 0, NGM_SCALE_UNIT_BALL = 1, NGM_SCALE_UNIT_CUBE = 2 };
enum ngm_geometry_mode { NGM_GEO_NRGBD = 0, NGM_GEO_OCCUPANCY = 1, NGM_GEO_DENSITY = 2, NGM_GEO_NEUS = 3 };

/* slots of the loss partial-sum vector (rm.py:1769-1872); counts are stored as floats */
enum ngm_loss_slot {
  NGM_LS_PHOTO_SUM = 0, /* sum |rgb - rgb*| over masked rays, 3 channels     */
  NGM_LS_PHOTO_CNT = 1, /* number of masked rays (mean divides by 3*cnt)      */
  NGM_LS_DEPTH_SUM = 2, /* sum huber(depth - depth*)                          */
  NGM_LS_DEPTH_CNT = 3,
  NGM_LS_FS_SUM = 4,    /* sum (g*tau - tau)^2 over free-space samples        */
  NGM_LS_FS_CNT = 5,
  NGM_LS_TSDF_SUM = 6,  /* sum (g*tau - (gt - t))^2 over truncation samples   */
  NGM_LS_TSDF_CNT = 7,
  NGM_LS_TERM_SUM = 8,  /* sum (term - term*)^2 over term_mask rays           */
  NGM_LS_TERM_CNT = 9,
  NGM_LS_PHOTO_L1_SUM = 10 /* photometric gaussian_nll only: sum |rgb - rgb*| for its L1 branch (losses.py:34-35) */
};

/* NeuralField architecture: models.py:69-128, positional_encodings.py:167-195,222-243 */
typedef struct ngm_field_cfg {
  int32_t encoding;     /* ngm_encoding */
  int32_t dim_enc;      /* D: encoding output width (Fourier: dim_out; NeRF: 6*octaves)      */
  int32_t raw_coords;   /* Fourier: 1 -> cat(x, sin(Wx)), W is (D-3,3); 0 -> sin(Wx), (D,3)  */
  int32_t num_octaves;  /* NeRF */
  int32_t start_octave; /* NeRF */
  int32_t num_layers;   /* L hidden Linear+ReLU layers (skip_mode "no")                     */
  int32_t dim_hidden;   /* H (= D when dim_mlp_out is null, models.py:99-100)               */
  int32_t dim_out;      /* 4: r,g,b,geometry                                                */
  int32_t scale_mode;   /* ngm_scale_mode, models.py:278-285                                */
  float field_radius;
  /* permutohedral hash encoding (positional_encodings.py:19-66, config/neural_graph_map.yaml:6-14).
   * PARITY UNPINNED: the arithmetic of the reference lives in an un-vendored CUDA package; the kernels
   * implement the published lattice algorithm as restated in oracle/ngm_oracle.py:encode_permuto. */
  int32_t nr_levels;          /* L <= 16; dim_enc = L * nr_feat_per_level                          */
  int32_t nr_feat_per_level;  /* 2                                                                 */
  int32_t log2_hashmap_size;  /* table entries per level T = 2^log2                                */
  float coarsest_scale, finest_scale; /* sigma_l = geomspace(coarsest, finest, L)                  */
  float level_scale[16 * 3];  /* per level, per axis i: 1 / (sqrt((i+1)(i+2)) * sigma_l); fill with    */
                              /* ngm_permuto_fill_scales() (or from numpy.geomspace, as _capi.py does)  */
  int32_t skip_mode;          /* ngm_skip_mode: "add" adds the encoding to the first D units after every       */
                              /* hidden layer (models.py:162-169); needs dim_hidden >= dim_enc.  "concat"       */
                              /* appends it (models.py:159-161): layers 1..L (incl. the output layer) then have */
                              /* H + D inputs, "_linears.{i}.weight" is (out_i, H + D) for i >= 1               */
  int32_t matmul_mode;        /* ngm_matmul_mode of the hidden layers' matrix products (see the enum)             */
  /* triplane encoding (positional_encodings.py:69-161): three (C, res, res) feature planes per field, bilinear lookup
   * (grid_sample, align_corners, border padding) of the (x,y), (x,z), (y,z) projections of a point in [-1,1]^3,
   * combined per ngm_triplane_mode; dim_enc = C (sum, product) or 3 C (concat) */
  int32_t tri_resolution;
  int32_t tri_mode;
  /* ABI 10: two choices that used to be process-wide switches, now part of the configuration (two renderers of one process
   * may differ; ngm_render_workspace sizes the workspace for the configuration it is handed) */
  int32_t activation_stash;   /* ngm_activation_stash: what the training forward of a two-hidden-layer network on the split
                               * path stashes for the backward                                                            */
  int32_t hash_grad_atomics;  /* ngm_hash_grad_atomics: accumulation of the hash-table gradient (k_hash_grad)               */
} ngm_field_cfg;
/* NGM_STASH_FULL: both hidden layers' outputs (512 B per sample; fastest).  NGM_STASH_HALF: layer 0's output only (256 B
 * per sample: half the stash memory and HBM traffic); k_field_bwd_b3<HS> recomputes the output layer's input on the matrix
 * pipe.  Same results at the same tolerances, each bitwise reproducible. */
enum ngm_activation_stash { NGM_STASH_FULL = 0, NGM_STASH_HALF = 1 };
/* NGM_HASH_ATOMICS_EXACT (default): Q23.40 fixed-point integer LDS atomics -- order-independent, the table gradient is
 * bitwise reproducible.  NGM_HASH_ATOMICS_FLOAT (opt-in): fp32 LDS atomics, what the reference's CUDA package does with its
 * global float atomics -- the same sums up to the rounding of the order the adds happen to land in (NOT reproducible run
 * to run), half the LDS per workgroup, no fixed-point conversion. */
enum ngm_hash_grad_atomics { NGM_HASH_ATOMICS_EXACT = 0, NGM_HASH_ATOMICS_FLOAT = 1 };

/* fills cfg->level_scale from nr_levels / coarsest_scale / finest_scale (double precision) */
int ngm_permuto_fill_scales(ngm_field_cfg* cfg);

/* Stacked per-field parameters, models.py:245-276 (`all_fields_params` / `vmap_fields_params`).
 * One device pointer per tensor plus the element stride between consecutive fields, so both a
 * dict of separate (N,...) tensors and views into one arena are accepted.  `field_index`
 * (optional, int64[F]) selects rows: batch field f uses row field_index[f] (NULL: row f). */
typedef struct ngm_params {
  const float* enc_w; /* "_encoding._linear.weight" (N, D-3|D, 3); NULL unless Fourier */
  int64_t enc_w_stride;
  const float* w[NGM_MAX_LAYERS + 1]; /* "_linears.{i}.weight" (N, out_i, in_i) */
  int64_t w_stride[NGM_MAX_LAYERS + 1];
  const float* b[NGM_MAX_LAYERS + 1]; /* "_linears.{i}.bias" (N, out_i) */
  int64_t b_stride[NGM_MAX_LAYERS + 1];
  const int64_t* field_index;
  const float* lattice; /* "_encoding.lattice_values" (N, L, T, 2); NULL unless permutohedral */
  int64_t lattice_stride;
  const float* shift;   /* "_encoding.random_shift_per_level" (N, L, 3); always fp32 (constants) */
  int64_t shift_stride;
  int32_t dtype;        /* ngm_param_dtype: STORAGE type of enc_w, w[], b[] and lattice.  NGM_DT_BF16 / NGM_DT_F16: the
                         * pointers above address 16-bit elements (strides stay element counts); the kernels widen
                         * them to fp32 when they stage a field's weights into LDS / gather the hash table, and
                         * compute in fp32 exactly as for fp32 storage (BASELINE configs 1 and 4: bf16 / fp16 weights;
                         * the reference itself is fp32 only).  Gradients and Adam moments stay fp32.            */
  int32_t reserved_;
  const float* planes;  /* "_encoding.plane_coef" (N, 3, C, res, res), fp32; NULL unless triplane */
  int64_t planes_stride;
  const float* neus_sd; /* "_neus_sd" (N,), fp32: per-field standard deviation of the neus geometry mode (rm.py:641-644); */
  int64_t neus_sd_stride; /* required by the fused render entry points in that mode, ignored otherwise                      */
} ngm_params;

/* Gradient outputs, same layout rules as ngm_params (row f of each tensor = batch field f). */
typedef struct ngm_grads {
  float* enc_w;
  int64_t enc_w_stride;
  float* w[NGM_MAX_LAYERS + 1];
  int64_t w_stride[NGM_MAX_LAYERS + 1];
  float* b[NGM_MAX_LAYERS + 1];
  int64_t b_stride[NGM_MAX_LAYERS + 1];
  float* lattice;       /* (F, L, T, 2): fully overwritten by the backward entry points                */
  int64_t lattice_stride;
  float* neus_sd;       /* (F,) d loss / d "_neus_sd" (neus geometry mode, fused render backward) or NULL */
  float* planes;        /* (F, 3, C, res, res): fully overwritten by the backward entry points (triplane) */
  int64_t planes_stride;
} ngm_grads;

/* Loss modes of losses.py built into the fused kernels: all of them.
 * losses.py:26-36.  GAUSSIAN_NLL: 0.5 e^2 / var + log sqrt(var) over the rendered colour variances (rm.py:781-785, no epsilon),
 * replaced by the L1 loss whenever its global mean exceeds 2 (the reference's data-dependent switch). */
typedef enum ngm_photometric_mode { NGM_PHOTO_L1 = 0, NGM_PHOTO_L2 = 1, NGM_PHOTO_GAUSSIAN_NLL = 2 } ngm_photometric_mode;
/* losses.py:60-75.  GAUSSIAN_NLL: 0.5 e^2 / (var + 1e-15) + log sqrt(var + 1e-15); LAPLACIAN_NLL: |e| / sqrt(0.5 var + 1e-6)
 * + 0.5 log(2 var + 1e-6), var = the rendered depth variance (rm.py:786-790).  The variance-weighted modes differentiate
 * through the variances; they run the compositing backward as a launch of its own (k_stash_bwd). */
typedef enum ngm_depth_mode { NGM_DEPTH_HUBER = 0, NGM_DEPTH_GAUSSIAN_NLL = 1, NGM_DEPTH_LAPLACIAN_NLL = 2 } ngm_depth_mode;

/* Renderer + loss constants: rm.py:116-220, config/neural_graph_map.yaml */
typedef struct ngm_render_cfg {
  int32_t geometry_mode;       /* ngm_geometry_mode, rm.py:746-762            */
  int32_t num_samples_coarse;  /* S_c                                         */
  int32_t num_samples_guided;  /* S_g (0: single stratum, eval style)         */
  int32_t overwrite_behind_camera; /* != 0: samples with z_cam > 0 get a constant geometry value and no
                                    * gradient (rm.py:614-622: -100 occupancy/density, 1.0 neus/nrgbd)  */
  float geometry_factor;       /* gamma                                       */
  float color_factor;
  float truncation_distance;   /* tau                                         */
  float range_depth_guided;    /* rho (rm.py:169-170)                         */
  float fx, fy, cx, cy;        /* intrinsics at pixel centre 0 (camera.py:188) */
  float w_termination, w_photometric, w_depth, w_freespace, w_tsdf; /* rm.py:129-135 */
  float huber_delta;           /* 0.05, losses.py:63                          */
  float term_threshold;        /* 0.8, rm.py:1787                             */
  int32_t photometric_mode;    /* ngm_photometric_mode (config key photometric_loss, losses.py:26-29); ABI 3 */
  int32_t depth_mode;          /* ngm_depth_mode (config key depth_loss, losses.py:60-75)                    */
} ngm_render_cfg;

/* One batch of rays: the reference's Target record (rm.py:43-58) in device memory. */
typedef struct ngm_rays {
  int32_t F, R;
  const int64_t* ijs;      /* (F,R,2) [row, col]                                            */
  const float* c2ws;       /* (F,R,4,4) when c2w_per_ray != 0 else (4,4) shared              */
  int32_t c2w_per_ray;
  int32_t philox_offset_autoinc; /* != 0 (training forward with targets only): *philox_offset_dev is incremented once
                                  * after every workgroup of the forward has read it (the loss-reduction kernel does
                                  * it), i.e. the counter counts iterations; ngm_adam_sparse_multi can read the same
                                  * counter as its device-side step.  The pointer is then written through.         */
  const float* near;       /* (F,R) or NULL -> near_const                                    */
  const float* far;        /* (F,R) or NULL -> far_const                                     */
  const float* gt;         /* (F,R) or NULL; 0.0 = no depth                                  */
  float near_const, far_const;
  const float* field_pos;  /* (F,3) field positions in the world frame                       */
  const float* field_quat; /* (F,4) real-first quaternions, applied like pytorch3d's quaternion_apply (raw        */
                           /* Hamilton products, models.py:338-339: a norm != 1 scales the local frame by |q|^2) */
  const float* u_coarse;   /* (F,R,S_c) torch.rand draws of camera.py:274, or NULL -> Philox */
  const float* u_guided;   /* (F,R,S_g) or NULL -> Philox                                    */
  const float* lin_coarse; /* (S_c+1) torch.linspace(0,1) table of camera.py:271 or NULL     */
  const float* lin_guided; /* (S_g+1) or NULL                                                */
  uint64_t philox_seed;    /* used when u_* is NULL                                          */
  uint64_t philox_offset;
  const int64_t* pose_index;          /* optional (F,): batch field f uses field_pos/field_quat row pose_index[f] */
  const uint64_t* philox_offset_dev;  /* optional device counter added to philox_offset (hipGraph replay:  */
                                      /* a captured step draws fresh jitter every replay)                   */
} ngm_rays;

/* Supervision of one batch (rm.py:43-58, 1769-1872); masks are uint8 0/1. */
typedef struct ngm_targets {
  const float* rgbds;        /* (F,R,4) r,g,b,depth                 */
  const uint8_t* depth_mask; /* (F,R)                               */
  const uint8_t* term_mask;  /* (F,R) or NULL (all false)           */
  const float* term_probs;   /* (F,R) or NULL                       */
} ngm_targets;

/* Per-ray outputs: the reference's Prediction (rm.py:59-69) minus the data-dependent vectors. */
typedef struct ngm_prediction {
  float* rgbds;      /* (F,R,4) */
  float* color_vars; /* (F,R,3) */
  float* depth_vars; /* (F,R)   */
  float* term_probs; /* (F,R)   */
} ngm_prediction;

/* ---- library ------------------------------------------------------------------------------ */
int ngm_abi_version(void);
const char* ngm_last_error(void);
/* number of compute units / device name of the current device (for launch sizing, bench) */
int ngm_device_info(int* num_cus, char* name, int name_len);

/* ---- K1: ray sampler ----------------------------------------------------------------------
 * Replaces Camera.ijs_to_directions + Camera.sample_ijs_uniform (camera.py:186-292), the
 * depth-guided merge + sort + gather (rm.py:521-545).  Outputs the sorted samples in the camera
 * frame: points_cam (F,R,S,3), distances (F,R,S), dirs (F,R,3) (any may be NULL). */
int ngm_sample_rays(const ngm_render_cfg* cfg, const ngm_rays* rays, float* points_cam,
                    float* distances, float* dirs, void* stream);
/* same, additionally the samples in the WORLD frame (utils.transform_points, utils.py:276-286 via
 * rm.py:547): points_world (F,R,S,3) = R(c2w) p_cam + t(c2w). */
int ngm_sample_rays_world(const ngm_render_cfg* cfg, const ngm_rays* rays, float* points_cam,
                          float* points_world, float* distances, float* dirs, void* stream);
/* Camera.sample_ijs_uniform with `weights` / `boundaries` (camera.py:227-244, 277-289; no caller inside run_mapping.py, part of
 * the Camera interface): weighted sampling from distance bins given per ray by sorted boundaries (F,R,num_bins+1) and bin
 * probabilities weights (F,R,num_bins): bin = searchsorted(cumsum(weights) + 1e-3, u_bin), distance = boundaries[bin] +
 * (boundaries[bin+1] - boundaries[bin]) * u_off; cfg->num_samples_coarse samples per ray, in DRAW order (the reference does
 * not sort this branch).  rays->u_coarse = the first torch.rand draw (bins), rays->u_guided = the second (offsets), both
 * (F,R,S), or both NULL = words 0 / 1 of Philox block (ray * S + sample, stream 0); rays->near / far / gt are not read.  The running sum is torch.cumsum's on the
 * CPU (sequential, fp64 accumulator, every prefix rounded to fp32); a draw beyond the last cumulative weight takes the last bin (the reference's gather is out
 * of range there).  Outputs as ngm_sample_rays (any may be NULL). */
int ngm_sample_rays_weighted(const ngm_render_cfg* cfg, const ngm_rays* rays, int32_t num_bins, const float* boundaries,
                             const float* weights, float* points_cam, float* distances, float* dirs, void* stream);

/* ---- K2+K3: NeuralFieldSet.forward(use_vmap=True) -----------------------------------------
 * models.py:329-345: world -> field-local transform, scaling, encoding, MLP; points (F,P,3)
 * (world frame when field_pos/field_quat are given, local otherwise), out (F,P,4). */
int ngm_field_eval_fwd(const ngm_field_cfg* fcfg, const ngm_params* params, int32_t F, int64_t P,
                       const float* points, const float* field_pos, const float* field_quat,
                       float* out, void* stream);
/* The positional encoding of the same points ALONE (standalone stage entry point for roofline accounting;
 * positional_encodings.py:19-66 permutohedral hash, :164-276 Fourier / NeRF octaves): out (F,P,dim_enc).  The fused kernels
 * never materialise this tensor.  Triplane: NGM_E_UNSUPPORTED. */
int ngm_encode_fwd(const ngm_field_cfg* fcfg, const ngm_params* params, int32_t F, int64_t P,
                   const float* points, const float* field_pos, const float* field_quat,
                   float* out, void* stream);
/* Backward of the encoding alone: d_enc (F,P,dim_enc) -> the encoding's parameter gradients, fully overwritten.  Fourier
 * (positional_encodings.py:197-212): grads->enc_w (F, dim_enc - 3, 3) = sum_p d_enc cos(W x) x; permutohedral hash
 * (:19-66): grads->lattice (F, L, T, 2) through the table-gradient kernel of the training step.  NeRF octaves / no encoding
 * have no parameters (returns NGM_OK, nothing written); the triplane encoding returns NGM_E_UNSUPPORTED (its gradient exists
 * inside ngm_field_eval_bwd only).  Other members of `grads` are ignored. */
int64_t ngm_encode_bwd_workspace(const ngm_field_cfg* fcfg, int32_t F, int64_t P);
int ngm_encode_bwd(const ngm_field_cfg* fcfg, const ngm_params* params, int32_t F, int64_t P, const float* points,
                   const float* field_pos, const float* field_quat, const float* d_enc, const ngm_grads* grads,
                   void* workspace, int64_t workspace_bytes, void* stream);
/* Backward of the above w.r.t. every parameter: d_out (F,P,4) -> grads.  workspace: see
 * ngm_field_eval_bwd_workspace(). */
int ngm_field_eval_bwd(const ngm_field_cfg* fcfg, const ngm_params* params, int32_t F, int64_t P,
                       const float* points, const float* field_pos, const float* field_quat,
                       const float* d_out, const ngm_grads* grads, void* workspace,
                       int64_t workspace_bytes, void* stream);
int64_t ngm_field_eval_bwd_workspace(const ngm_field_cfg* fcfg, int32_t F, int64_t P);
/* Training pair of the above (ABI 11; models.py:329-345 under autograd -- the path the reference's unchanged
 * _optimization_iteration takes, rm.py:1164-1177).  ngm_field_eval_fwd_train computes what ngm_field_eval_fwd computes (same
 * kernel, same bits) and also writes the hidden activations of every sample into the caller-owned `stash`
 * (ngm_field_eval_stash_bytes: 256 B per sample and hidden layer + one tile); ngm_field_eval_bwd_stash reads them back instead
 * of recomputing the hidden layers and runs the MLP backward of the fused training step (bf16-split matrix products, fp32
 * accumulate) on the explicit points.  ngm_field_eval_stash_bytes returns 0 when the configuration has no stash-reading
 * backward (anything but 33..64-wide encoding and hidden layers, 1-2 layers, Fourier / NeRF / no encoding, skip_mode no,
 * matmul_mode auto / bf16x3): the pair then returns NGM_E_UNSUPPORTED and ngm_field_eval_fwd / ngm_field_eval_bwd serve the
 * call.  `points`, poses and parameters passed to the backward must be those of the forward that wrote the stash.
 * workspace of the backward: ngm_field_eval_bwd_workspace(). */
int64_t ngm_field_eval_stash_bytes(const ngm_field_cfg* fcfg, int32_t F, int64_t P);
int ngm_field_eval_fwd_train(const ngm_field_cfg* fcfg, const ngm_params* params, int32_t F, int64_t P,
                             const float* points, const float* field_pos, const float* field_quat,
                             float* out, void* stash, int64_t stash_bytes, void* stream);
int ngm_field_eval_bwd_stash(const ngm_field_cfg* fcfg, const ngm_params* params, int32_t F, int64_t P,
                             const float* points, const float* field_pos, const float* field_quat,
                             const float* d_out, const ngm_grads* grads, const void* stash, int64_t stash_bytes,
                             void* workspace, int64_t workspace_bytes, void* stream);

/* ---- K4: volume renderer -------------------------------------------------------------------
 * NeuralGraphMap._quadrature (rm.py:709-799) on N rays x S samples: colors (N,S,3), geoms (N,S),
 * dists (N,S), depths (N,S), neus_isds (N) or NULL.  Outputs: C (N,3), D (N), Cvar (N,3),
 * Dvar (N), term (N), weights (N,S_eff) (any may be NULL). */
int ngm_composite_fwd(const ngm_render_cfg* cfg, int64_t N, int32_t S, const float* colors,
                      const float* geoms, const float* dists, const float* depths,
                      const float* neus_isds, float* C, float* D, float* Cvar, float* Dvar,
                      float* term, float* weights, void* stream);
/* Same quadrature fed directly by the (N,S,4) field outputs [r,g,b,geometry] (colour scaled by
 * cfg->color_factor, rm.py:610-612) and the camera-frame sample points (depth = -z): the eval path
 * render_image -> _render_ijs(use_vmap=False) -> _quadrature (rm.py:402-437, 586-595, 650-656). */
int ngm_composite_fwd_packed(const ngm_render_cfg* cfg, int64_t N, int32_t S, const float* field_out4,
                             const float* dists, const float* points_cam, float* rgbd, float* Cvar,
                             float* Dvar, float* term, void* stream);
/* Backward of C, D, term w.r.t. colors (N,S,3), geoms (N,S) and, in neus mode, the per-ray inverse
 * standard deviations (N) (d_neus_isds may be NULL).  All four geometry modes. */
int ngm_composite_bwd(const ngm_render_cfg* cfg, int64_t N, int32_t S, const float* colors,
                      const float* geoms, const float* dists, const float* depths,
                      const float* neus_isds, const float* dC, const float* dD,
                      const float* dterm, float* d_colors, float* d_geoms, float* d_neus_isds,
                      void* stream);

/* ---- fused render / train step -------------------------------------------------------------
 * ngm_render_fwd replaces NeuralGraphMap._render_ijs(use_vmap=True) (rm.py:439-666): sampler ->
 * world->local -> encoding -> MLP -> compositing in ONE kernel.  When `targets` is non-NULL it
 * also accumulates the loss partial sums of _compute_losses (rm.py:1769-1872) into
 * loss_sums[NGM_NUM_LOSS_SUMS] (device, overwritten) and fills the saved-for-backward part of
 * the workspace.  loss_sums == NULL with targets = DEFERRED reduction: the per-workgroup partial sums
 * stay in the workspace and the matching ngm_render_bwd / ngm_render_bwd_adam call (also with
 * loss_sums == NULL) sums them itself -- one launch less per step on a single GPU, where nothing
 * (no all-reduce) happens between forward and backward; the philox_offset_autoinc counter is then
 * advanced by the backward.  sample_stash (optional, (F,R,S,6): r,g,b,geometry,t,T) exposes the per-sample
 * values for callers that need the compacted free-space / TSDF vectors of the Prediction. */
int64_t ngm_render_workspace(const ngm_field_cfg* fcfg, const ngm_render_cfg* rcfg, int32_t F,
                             int32_t R, int32_t train);
int ngm_render_fwd(const ngm_field_cfg* fcfg, const ngm_render_cfg* rcfg,
                   const ngm_params* params, const ngm_rays* rays, const ngm_targets* targets,
                   const ngm_prediction* pred, float* loss_sums, void* workspace,
                   int64_t workspace_bytes, void* stream);
/* Backward of loss["combined"] (rm.py:1871) w.r.t. every field parameter.  loss_sums are the
 * GLOBAL sums/counts (after the caller's cross-GPU all-reduce of ngm_render_fwd's output);
 * workspace must be the one filled by the matching ngm_render_fwd call.  The workspace is
 * SINGLE-USE: every ngm_render_bwd* call overwrites the saved forward values with the per-sample
 * gradients in place, so a second backward needs a new ngm_render_fwd.  loss_out (optional,
 * device float[8]): combined, termination, photometric, depth, freespace, tsdf. */
int ngm_render_bwd(const ngm_field_cfg* fcfg, const ngm_render_cfg* rcfg,
                   const ngm_params* params, const ngm_rays* rays, const ngm_targets* targets,
                   const ngm_prediction* pred, const float* loss_sums, const ngm_grads* grads,
                   float* loss_out, void* workspace, int64_t workspace_bytes, void* stream);
/* Generic-seed variant used by the autograd wrapper of render_ijs: explicit dL/d(rgbds) (F,R,4),
 * dL/d(term_probs) (F,R) and optional per-sample dL/d(geometry) (F,R,S). */
int ngm_render_bwd_seeded(const ngm_field_cfg* fcfg, const ngm_render_cfg* rcfg,
                          const ngm_params* params, const ngm_rays* rays, const float* d_rgbds,
                          const float* d_term, const float* d_geom_samples,
                          const ngm_grads* grads, void* workspace, int64_t workspace_bytes,
                          void* stream);
/* The same with seeds on the rendered variances as well (rm.py:781-790: V = sum_k w_k (c_k - C)^2): dL/d(color_vars) (F,R,3)
 * and dL/d(depth_vars) (F,R), either may be NULL; `pred` = the forward's outputs (rgbds and term_probs are read: the means and
 * the weight sum the variances are taken around).  What the reference's autograd does for losses.py's *_nll modes. */
int ngm_render_bwd_seeded_vars(const ngm_field_cfg* fcfg, const ngm_render_cfg* rcfg,
                               const ngm_params* params, const ngm_rays* rays, const ngm_prediction* pred,
                               const float* d_rgbds, const float* d_color_vars, const float* d_depth_vars,
                               const float* d_term, const float* d_geom_samples, const ngm_grads* grads,
                               void* workspace, int64_t workspace_bytes, void* stream);
/* read access to the per-sample values saved by ngm_render_fwd(train): copies geometry (F,R,S)
 * and sorted distances (F,R,S) out of the workspace (for Prediction.freespace_geometry /
 * tsdf_residuals, rm.py:624-639).  S = the samples per ray the forward ran with: S_c + S_g when
 * its rays carried gt, S_c otherwise (anything else: NGM_E_INVALID); geoms / dists hold F*R*S floats. */
int ngm_render_read_samples(const ngm_field_cfg* fcfg, const ngm_render_cfg* rcfg, int32_t F,
                            int32_t R, int32_t S, const void* workspace, float* geoms, float* dists,
                            void* stream);

/* ---- sparse per-field Adam (rm.py:347-389, 668-707, 1183-1221) ---------------
 * torch.optim.Adam (L2-coupled weight decay) applied in place to rows field_index[f] of one
 * stacked tensor of `numel_per_field` elements: param/exp_avg/exp_avg_sq have `stride` elements
 * between fields, grad is (F, numel_per_field) with grad_stride.  `step` is the NEW shared step
 * count (one counter for all fields, rm.py:380-385). */
int ngm_adam_sparse(float* param, float* exp_avg, float* exp_avg_sq, int64_t stride,
                    const float* grad, int64_t grad_stride, const int64_t* field_index, int32_t F,
                    int64_t numel_per_field, int64_t step, float lr, float beta1, float beta2,
                    float eps, float weight_decay, void* stream);

/* All parameter tensors of a field set in ONE launch.  `step_dev` (optional device int64) overrides
 * `step` so that a captured hipGraph advances the bias correction on every replay.  End-of-iteration
 * bookkeeping can ride along: with advance_step_dev != 0 the kernel does ++*step_dev, and with a non-NULL
 * advance_philox_offset_dev ++*that, once every block has finished (same effect as ngm_step_advance, but
 * measured slower than the separate launch on MI355X: every block fences its stores first). */
typedef struct ngm_adam_tensor {
  float* param; float* exp_avg; float* exp_avg_sq; /* (N, numel) rows with `stride` elements between fields */
  const float* grad;                               /* (F, numel) rows with `grad_stride`                    */
  int64_t stride, grad_stride, numel;
  void* param_lp;                                  /* optional reduced-precision copy of `param` (same row layout,  */
  int32_t lp_dtype;                                /* 16-bit elements, ngm_param_dtype): refreshed with every update */
  int32_t reserved_;                               /* (fp32 master weights + the copy the kernels read)              */
} ngm_adam_tensor;
int ngm_adam_sparse_multi(const ngm_adam_tensor* tensors, int32_t num_tensors, const int64_t* field_index,
                          int32_t F, int64_t step, int64_t* step_dev, float lr, float beta1, float beta2,
                          float eps, float weight_decay, int32_t advance_step_dev,
                          uint64_t* advance_philox_offset_dev, void* stream);
/* ngm_render_bwd + the sparse Adam update of the active fields (rm.py:1183-1221) with the MLP tensors updated by the
 * gradient-reduction kernel itself (no Adam launch, no gradient round trip; the gradients are still written to `grads`).
 * mlp_tensors: one entry per gradient segment, in the order enc_w (Fourier encoding only), w_0, b_0, ..., w_L, b_L
 * (their `grad` members are ignored); lattice_tensor: the hash tables (permutohedral encoding only, NULL otherwise),
 * updated by the table-gradient reduction kernel.  step / step_dev as in ngm_adam_sparse_multi. */
int ngm_render_bwd_adam(const ngm_field_cfg* fcfg, const ngm_render_cfg* rcfg, const ngm_params* params,
                        const ngm_rays* rays, const ngm_targets* targets, const ngm_prediction* pred,
                        const float* loss_sums, const ngm_grads* grads, const ngm_adam_tensor* mlp_tensors,
                        int32_t num_mlp_tensors, const ngm_adam_tensor* lattice_tensor, const int64_t* field_index,
                        int64_t step, int64_t* step_dev, float lr, float beta1, float beta2, float eps,
                        float weight_decay, float* loss_values, void* workspace, int64_t workspace_bytes, void* stream);
/* ++*step_dev, ++*philox_offset_dev on the stream (either may be NULL): end-of-iteration bookkeeping
 * for graph-captured training loops. */
int ngm_step_advance(int64_t* step_dev, uint64_t* philox_offset_dev, void* stream);

/* ---- eval path: kNN-blended field evaluation (models.py:347-405) --------------------------
 * points (P,3) world; all N_f fields' poses; params cover all N_f fields (field_index optional,
 * maps field slot -> parameter row).  out (P,4).  K = min(num_knn, N_f) <= 16 (unrolled neighbour lists for K <= 8, one 16-slot instance above); N_f unbounded (the centres are binned into a
 * uniform grid in the workspace per call; exact K nearest, distance ties to the lower field index).
 * mask_radius: the `field_radius` ARGUMENT of NeuralFieldSet.forward (models.py:293, 368): a point is evaluated when its
 * nearest field centre is closer than this; the local coordinates are still scaled with fcfg->field_radius
 * (models.py:278-285, 378) -- _extract_mesh colours its vertices with radius + 0.1 (rm.py:2324-2336).  <= 0: fcfg->field_radius. */
int64_t ngm_field_eval_knn_workspace(int32_t num_fields, int64_t P, int32_t num_knn);
int ngm_field_eval_knn(const ngm_field_cfg* fcfg, const ngm_params* params, int32_t num_fields,
                       int64_t P, const float* points, const float* field_pos,
                       const float* field_quat, int32_t num_knn, float distance_factor,
                       float outside_value, float mask_radius, float* out, void* workspace, int64_t workspace_bytes,
                       void* stream);

/* ---- eval path, whole: render_image's loop over pixel blocks (rm.py:402-437) -------------------------------------------
 * = per block of `ray_block` rays: Camera.sample_ijs_uniform + transform_points (rm.py:513-547, eval-style: one stratum of
 * rcfg->num_samples_coarse samples, no depth guidance -- rays->gt / u_guided are ignored), NeuralFieldSet.forward(
 * use_vmap=False) (models.py:347-405) and _quadrature (rm.py:709-799) on the blended outputs.  The same arithmetic as
 * ngm_sample_rays_world -> ngm_field_eval_knn -> ngm_composite_fwd_packed per block, without their intermediates in memory:
 * the samples are drawn inside the neighbour assignment, the blend happens inside the quadrature, the grid over the field
 * centres is built once per call.  rays: rays->F * rays->R rays in all (field_pos / field_quat / pose_index unused); block b
 * draws its jitter from u_coarse, or from the Philox stream (philox_seed + b * ray_block, ray index within the block).
 * pred: (F*R, .) outputs, any may be NULL.  mask_radius as in ngm_field_eval_knn; K <= 8; S <= 1024. */
int64_t ngm_render_eval_knn_workspace(const ngm_render_cfg* rcfg, int32_t num_fields, int32_t ray_block, int32_t num_knn);
int ngm_render_eval_knn(const ngm_field_cfg* fcfg, const ngm_render_cfg* rcfg, const ngm_params* params,
                        int32_t num_fields, const float* field_pos, const float* field_quat, const ngm_rays* rays,
                        int32_t num_knn, float distance_factor, float outside_value, float mask_radius, int32_t ray_block,
                        const ngm_prediction* pred, void* workspace, int64_t workspace_bytes, void* stream);

/* ---- training-target sampler -------------------------------------------------------
 * Device part of NeuralGraphMap._sample_target_mv (rm.py:1259-1459).  The random draws stay with the caller
 * (torch.multinomial / randn / rand on its generator, as in the reference); the keyframe store is read in place. */
typedef struct ngm_keyframes {
  int32_t num_frames;            /* len(_c_c2w_tensor)                                              */
  int32_t height, width;         /* keyframe image size                                              */
  int32_t reserved0;
  const float* c2ws;             /* (num_frames,4,4) current keyframe poses, rm.py:1711-1713          */
  const float* rgbd;             /* (N_store,height,width,4) keyframe RGB-D store [r,g,b,depth]       */
  const int64_t* frame_to_store; /* (num_frames) _frame_cid_to_ncid, rm.py:1703                       */
  float fx, fy, cx, cy;          /* intrinsics at pixel centre 0 (camera.py:188)                      */
} ngm_keyframes;

/* outputs of ngm_target_rays = the reference's Target record (rm.py:43-58) for F fields x R rays */
typedef struct ngm_target_out {
  int64_t* ijs;        /* (F,R,2) [row, col]        */
  float* c2ws;         /* (F,R,4,4) or NULL         */
  float* near;         /* (F,R)                     */
  float* far;          /* (F,R)                     */
  float* gt;           /* (F,R) ray distance of the keyframe depth, 0 = missing */
  float* rgbds;        /* (F,R,4)                   */
  uint8_t* rgb_mask;   /* (F,R)                     */
  uint8_t* depth_mask; /* (F,R)                     */
  float* term_probs;   /* (F,R)                     */
  uint8_t* term_mask;  /* (F,R)                     */
} ngm_target_out;

/* rm.py:1321-1392: kf_mask (F,num_frames) u8 = keyframe sees the field; bbox (F,num_frames,4) =
 * (min_x, min_y, max_x, max_y) of the projected sphere samples, clamped to the image. */
int ngm_target_visibility(const ngm_keyframes* kf, int32_t F, const float* field_pos, int32_t num_offsets,
                          const float* offsets, float radius, uint8_t* kf_mask, float* bbox, void* stream);
/* rm.py:1394-1459: frame_cids (F,R) i64 = sampled keyframe per ray, u_xy (F,R,2) = pixel uniforms. */
int ngm_target_rays(const ngm_keyframes* kf, int32_t F, int32_t R, const float* field_pos, float radius,
                    const float* bbox, const int64_t* frame_cids, const float* u_xy,
                    const ngm_target_out* out, void* stream);

/* Device part of NeuralGraphMap._sample_target_sv (rm.py:1461-1583), the single-view variant (`update_mode: single_view`):
 * hit (F,N) u8 = the segment camera origin -> point n of the (subsampled) back-projected depth image passes through the
 * sphere of field f (geometry.py:67-105); field centres and points in the camera frame. */
int ngm_target_sv_intersect(int32_t F, int64_t N, const float* field_pos_cam, const float* points_cam, float radius,
                            uint8_t* hit, void* stream);
/* rm.py:1536-1561: segments (F,R) i64 = sampled point indices, pts_ijs (N,2) i64 their pixels, image (H,W,4) the RGB-D
 * frame; near / far are not clamped at 0 in this variant, rgb_mask = depth_mask = (gt < far), term_mask = 1, out->c2ws
 * is not written (one pose for all rays). */
int ngm_target_sv_rays(int32_t F, int32_t R, const float* field_pos_cam, float radius, const int64_t* pts_ijs,
                       const int64_t* segments, const float* image, int32_t height, int32_t width, float fx, float fy, float cx,
                       float cy, const ngm_target_out* out, void* stream);

/* ---- mesh extraction -------------------------------------------------------------------
 * Marching cubes on a dense grid of field values: replaces the pytorch3d.ops.marching_cubes call of
 * NeuralGraphMap._extract_mesh (rm.py:2255-2298; the grid itself is filled by ngm_field_eval_knn, rm.py:2255-2261).
 * volume (nx,ny,nz) fp32 row-major (z fastest), "inside" = value > isolevel (the caller negates nrgbd / neus volumes,
 * rm.py:2277-2289).  Two passes over the same volume / workspace:
 *   _count: classifies edges and cells, scans; counts (device int64[2]) = number of vertices, number of faces;
 *   _emit : verts (V,3) fp32 in GRID-INDEX coordinates (x in [0,nx-1], ...; the caller maps them to world
 *           coordinates as rm.py:2304-2317 does), faces (T,3) int64 vertex indices, outward normals
 *           (inside -> outside).  Entries beyond max_verts / max_faces are dropped.
 * Output order is deterministic: vertices by (grid point ((x*ny)+y)*nz+z, axis x<y<z) of the crossed edge's lower end,
 * faces by cell index ((x*(ny-1))+y)*(nz-1)+z.  One vertex per crossed edge (indexed, watertight mesh).
 * PARITY UNPINNED against pytorch3d (un-vendored): the triangulation table is derived from the cube topology
 * (oracle/mesh_oracle.py restates it); vertex positions are the same linear interpolation.
 * ngm_marching_cubes_tables copies the derived table (host call, no device): tri_table[256*15] edge ids (-1 padded),
 * tri_count[256]. */
int64_t ngm_marching_cubes_workspace(int32_t nx, int32_t ny, int32_t nz);
int ngm_marching_cubes_count(const float* volume, int32_t nx, int32_t ny, int32_t nz, float isolevel,
                             int64_t* counts, void* workspace, int64_t workspace_bytes, void* stream);
int ngm_marching_cubes_emit(const float* volume, int32_t nx, int32_t ny, int32_t nz, float isolevel, float* verts,
                            int64_t max_verts, int64_t* faces, int64_t max_faces, void* workspace,
                            int64_t workspace_bytes, void* stream);
int ngm_marching_cubes_tables(int8_t* tri_table, int32_t* tri_count);

/* ---- measurement hooks (bench.py roofline leg) ------------------------------------------------
 * When enabled, every launch of the listed kernels is bracketed by hipEvents recorded on the launch
 * stream; ngm_profile_read() synchronises them and returns the accumulated device time. */
enum ngm_kernel_id {
  NGM_K_RENDER_FWD = 0, /* fused forward (sampler+encode+MLP+composite)                 */
  NGM_K_STASH_BWD = 1,  /* compositing + loss backward on the stash                      */
  NGM_K_FIELD_BWD = 2,  /* MFMA backward of encoding + MLP (dominant kernel)             */
  NGM_K_GRAD_REDUCE = 3,
  NGM_K_ADAM = 4,
  NGM_K_POINTS_FWD = 5,
  NGM_K_COMPOSITE_FWD = 6,
  NGM_K_COMPOSITE_BWD = 7,
  NGM_K_HASH_GRAD = 8,   /* hash-table gradient: simplex search + LDS fixed-point scatter (one level per workgroup) */
  NGM_K_HASH_REDUCE = 9, /* sum of the per-workgroup partial tables (+ fused sparse Adam on the tables)             */
  NGM_K_LOSS_REDUCE = 10,
  NGM_K_KNN_ASSIGN = 11, /* evaluation path: exact K nearest fields per point                                       */
  NGM_K_KNN_EVAL = 12,   /* evaluation path: per-field MLP tiles over the (point, neighbour) pairs (dominant there) */
  NGM_K_SAMPLER = 13,    /* standalone ray sampler (k_sample_rays / k_sample_rays_elem / k_sample_rays_weighted)            */
  NGM_K_COUNT = 14
};
int ngm_profile_enable(int32_t on);
int ngm_profile_reset(void);
int ngm_profile_read(int32_t kernel_id, double* total_ms, int64_t* launches);

/* Debug: with NGM_PHASE_TIMING set in the environment the backward kernel's first wave records its
 * s_memtime cycles per phase (prologue, inputs, encode, forward, output layer, staging, wgrad, dgrad,
 * encoding grads, relu mask, -, epilogue, total); this copies the 16 counters of the last launch. */
int ngm_debug_phase_cycles(unsigned long long* out528);   /* 16 slots (-DNGM_PHASE_TIMING) + 8 x 64 timeline entries (-DNGM_BWD_TIMELINE) */
/* Same for the fused forward (library built with -DNGM_PHASE_TIMING): slots = prologue, ray setup, sampler, step head,
 * encoding, hidden layers, activation stash stores, output layer, compositing, variance pass, ray outputs, block
 * reduction, -, -, total shader cycles, total 100 MHz ticks; then the event timeline of the 8 waves of that workgroup. */
int ngm_debug_fwd_phase_cycles(unsigned long long* out528);   /* 16 summary slots + 8 waves x 64 timeline entries ((slot << 48) | cycles) */

/* Debug: which MLP backward kernel the last ngm_render_bwd* / ngm_field_eval_bwd call launched:
 * 0 = k_field_bwd (32-sample tiles, forward recompute), 1 = k_field_bwd16 (16-sample tiles, recompute),
 * 2 = k_field_bwd16s (16-sample tiles, hidden activations read from the forward's stash), 3 = k_field_bwd_b3 (three-way
 * bf16 split, 32-sample tiles, stash), 5 = k_hash_mlp_bwd (hash
 * encoding + one hidden layer of <= 32 units, three-way bf16 split, encoding stash), -1 = none yet. */
int ngm_debug_last_bwd_variant(void);
/* Debug: the arithmetic the last launch of a forward-type kernel resolved ngm_field_cfg.matmul_mode to (AUTO is resolved
 * per kernel and batch shape): which = 0 fused render forward (ngm_render_fwd), 1 point evaluation (ngm_field_eval_fwd),
 * 2 kNN evaluation (ngm_field_eval_knn).  Returns NGM_MATMUL_F32 or NGM_MATMUL_BF16X3, -1 before the first launch. */
int ngm_debug_last_matmul(int which);
/* 1 when the last fused render forward ran the instance whose wave step evaluates ONE 32-sample tile (batches of at most 32
 * samples per wave: small per-rank batches), else 0 */
int ngm_debug_last_fwd_one_tile(void);
/* Debug: 1 when the last ngm_render_bwd / ngm_render_bwd_adam ran the compositing backward inside k_field_bwd_b3 (loss
 * seeds, pointwise geometry modes; no k_stash_bwd launch, the forward's colour / geometry stash stays intact), 0 when
 * k_stash_bwd ran.  Environment: NGM_NO_FUSED_COMP=1 forces the separate kernel. */
int ngm_debug_last_comp_fused(void);
int ngm_debug_disable_fused_comp(int on);
/* DEVELOPER override of ngm_field_cfg.activation_stash for A/B timing of one library on one box (tools/): 0 / 1 force that
 * mode for every configuration of the process, -1 queries, -2 removes the override (environment NGM_STASH=full|half does
 * the same at load time).  The product path never calls it: the renderer sets ngm_field_cfg.activation_stash.
 * Returns the override in force before the call (-1: none). */
int ngm_debug_stash_mode(int mode);
int ngm_debug_last_stash_mode(void);   /* the stash the last MLP backward actually read: 0 / 1 as above, -1 none (recompute kernels) */
  /* 1 = always launch k_stash_bwd (as NGM_NO_FUSED_COMP=1); returns the previous setting */

/* ---- one-shot exchange of the loss sums between the ranks of one node ---------------
 * Replaces torch.distributed.all_reduce (RCCL) on the 16 floats between ngm_render_fwd and ngm_render_bwd* by ONE small
 * kernel that can be captured in the same hipGraph as the two: every rank writes its 16 values into every rank's
 * mailbox (peer memory mapped through hipIpc: xGMI stores), polls its own mailbox and sums in rank order -- bit-identical
 * sums on all ranks.  The reference has no counterpart (rm.py is single-process); the values are the sums / counts the
 * means of rm.py:1803-1871 are taken from.  Set-up (once per process group, host side, see distributed.PeerExchange):
 *   ngm_peer_alloc(ngm_peer_mailbox_bytes(), &mailbox)  ->  ngm_ipc_export(mailbox, handle)  ->  exchange the 64-byte
 *   handles by any means  ->  ngm_ipc_open(handle_of_rank_p, &px.mailbox[p]) for p != rank, px.mailbox[rank] = mailbox;
 *   px.seq / px.status: 8 + 4 bytes of zeroed device memory of this rank (ngm_peer_alloc works for them too).
 * Every rank must call ngm_loss_exchange the same number of times (idle ranks with zeros), like the collective it replaces.
 * A rank that waits longer than the time-out for a peer (default 30 s, ngm_peer_set_timeout; the all-reduce this replaces
 * would wait forever, a kernel must not: a dead peer would wedge the GPU) sets bit 0 of *status (sticky) and returns NaN in
 * all 16 sums: the losses and the update of THAT iteration are NaN on this rank, so the failure is visible at once and no
 * parameter is trained on partial normalisers.  A slot that already carries a later sequence number (ranks out of step
 * after such a time-out) is accepted and sets bit 1.  A non-zero status is fatal for the run. */
#define NGM_MAX_PEERS 8
typedef struct ngm_peer_exchange {
  int32_t world, rank;
  void* mailbox[NGM_MAX_PEERS];   /* [p] = rank p's mailbox as mapped into THIS process; [rank] = the own allocation   */
  unsigned long long* seq;        /* device counter of this rank, advanced by every exchange (starts at 0)             */
  int32_t* status;                /* device word of this rank: 0 = ok, bit 0 = time-out, bit 1 = out of step (sticky)   */
} ngm_peer_exchange;
int64_t ngm_peer_mailbox_bytes(void);
int ngm_peer_alloc(int64_t bytes, void** ptr);              /* zeroed fine-grained (uncached) device memory              */
int ngm_peer_free(void* ptr);
int ngm_ipc_export(void* ptr, unsigned char handle[64]);    /* hipIpcGetMemHandle                                        */
int ngm_ipc_open(const unsigned char handle[64], void** ptr);   /* hipIpcOpenMemHandle (lazy peer access)               */
int ngm_ipc_close(void* ptr);
int ngm_loss_exchange(const ngm_peer_exchange* px, float* loss_sums /* (16) device, summed in place */, void* stream);
/* How long an exchange waits for a peer before it gives up (process-wide, applies to launches and graph captures made
 * afterwards; <= 0 keeps the current value).  Returns the previous value in seconds.  Default 30 s, or NGM_PEER_TIMEOUT_S. */
double ngm_peer_set_timeout(double seconds);

#ifdef __cplusplus
}
#endif
#endif /* NGM_HIP_H */
