"""ctypes binding of the C ABI in include/ngm_hip.h (libngm_hip.so).

Torch-free: every call takes raw device addresses (ints) and a stream handle, so it serves both
the torch layer (``tensor.data_ptr()``) and the torch-free harness (``hiprt.DeviceArray.ptr``).
The product path has NO CPU fallback: if the library is missing, importing this module's `lib()`
raises.

NOTE on HIP runtimes: PyTorch-ROCm bundles its own libamdhip64.  A process that uses torch must
import torch BEFORE the first `lib()` call so that libngm_hip.so binds to the same runtime instance
(`ops` does this by importing torch at module import); the torch-free harness uses the system ROCm.
"""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
# NGM_LIB_PATH: developer experiments only (a variant build of the same sources, tools/variants.sh)
LIB_PATH = os.environ.get("NGM_LIB_PATH") or os.path.join(HERE, "lib", "libngm_hip.so")

NGM_MAX_LAYERS = 4
NGM_NUM_LOSS_SUMS = 16
ENC = {"none": 0, "fourier": 1, "nerf": 2, "permuto": 3, "triplane": 4}
TRI = {"sum": 0, "product": 1, "concat": 2}
SCALE = {"no": 0, "unit_ball": 1, "unit_cube": 2}
GEO = {"nrgbd": 0, "occupancy": 1, "density": 2, "neus": 3}
PHOTO = {"l1": 0, "l2": 1, "gaussian_nll": 2}                    # losses.py:26-36
DEPTH = {"huber": 0, "gaussian_nll": 1, "laplacian_nll": 2}      # losses.py:60-75
LS = dict(PHOTO_SUM=0, PHOTO_CNT=1, DEPTH_SUM=2, DEPTH_CNT=3, FS_SUM=4, FS_CNT=5, TSDF_SUM=6,
          TSDF_CNT=7, TERM_SUM=8, TERM_CNT=9, PHOTO_L1_SUM=10)

f32p = C.c_void_p
LArr = C.c_void_p * (NGM_MAX_LAYERS + 1)
SArr = C.c_int64 * (NGM_MAX_LAYERS + 1)


class FieldCfg(C.Structure):
    _fields_ = [("encoding", C.c_int32), ("dim_enc", C.c_int32), ("raw_coords", C.c_int32),
                ("num_octaves", C.c_int32), ("start_octave", C.c_int32), ("num_layers", C.c_int32),
                ("dim_hidden", C.c_int32), ("dim_out", C.c_int32), ("scale_mode", C.c_int32),
                ("field_radius", C.c_float), ("nr_levels", C.c_int32), ("nr_feat_per_level", C.c_int32),
                ("log2_hashmap_size", C.c_int32), ("coarsest_scale", C.c_float), ("finest_scale", C.c_float),
                ("level_scale", C.c_float * 48), ("skip_mode", C.c_int32), ("matmul_mode", C.c_int32),
                ("tri_resolution", C.c_int32), ("tri_mode", C.c_int32),
                ("activation_stash", C.c_int32), ("hash_grad_atomics", C.c_int32)]


class Params(C.Structure):
    _fields_ = [("enc_w", f32p), ("enc_w_stride", C.c_int64), ("w", LArr), ("w_stride", SArr),
                ("b", LArr), ("b_stride", SArr), ("field_index", C.c_void_p),
                ("lattice", f32p), ("lattice_stride", C.c_int64), ("shift", f32p), ("shift_stride", C.c_int64),
                ("dtype", C.c_int32), ("reserved_", C.c_int32), ("planes", f32p), ("planes_stride", C.c_int64),
                ("neus_sd", f32p), ("neus_sd_stride", C.c_int64)]


class Grads(C.Structure):
    _fields_ = [("enc_w", f32p), ("enc_w_stride", C.c_int64), ("w", LArr), ("w_stride", SArr),
                ("b", LArr), ("b_stride", SArr), ("lattice", f32p), ("lattice_stride", C.c_int64), ("neus_sd", f32p),
                ("planes", f32p), ("planes_stride", C.c_int64)]


class RenderCfg(C.Structure):
    _fields_ = [("geometry_mode", C.c_int32), ("num_samples_coarse", C.c_int32),
                ("num_samples_guided", C.c_int32), ("overwrite_behind_camera", C.c_int32),
                ("geometry_factor", C.c_float), ("color_factor", C.c_float),
                ("truncation_distance", C.c_float), ("range_depth_guided", C.c_float),
                ("fx", C.c_float), ("fy", C.c_float), ("cx", C.c_float), ("cy", C.c_float),
                ("w_termination", C.c_float), ("w_photometric", C.c_float), ("w_depth", C.c_float),
                ("w_freespace", C.c_float), ("w_tsdf", C.c_float), ("huber_delta", C.c_float),
                ("term_threshold", C.c_float), ("photometric_mode", C.c_int32), ("depth_mode", C.c_int32)]


class Rays(C.Structure):
    _fields_ = [("F", C.c_int32), ("R", C.c_int32), ("ijs", C.c_void_p), ("c2ws", f32p),
                ("c2w_per_ray", C.c_int32), ("philox_offset_autoinc", C.c_int32), ("near", f32p), ("far", f32p),
                ("gt", f32p), ("near_const", C.c_float), ("far_const", C.c_float),
                ("field_pos", f32p), ("field_quat", f32p), ("u_coarse", f32p), ("u_guided", f32p),
                ("lin_coarse", f32p), ("lin_guided", f32p), ("philox_seed", C.c_uint64),
                ("philox_offset", C.c_uint64), ("pose_index", C.c_void_p), ("philox_offset_dev", C.c_void_p)]


class AdamTensor(C.Structure):
    _fields_ = [("param", f32p), ("exp_avg", f32p), ("exp_avg_sq", f32p), ("grad", f32p),
                ("stride", C.c_int64), ("grad_stride", C.c_int64), ("numel", C.c_int64),
                ("param_lp", C.c_void_p), ("lp_dtype", C.c_int32), ("reserved_", C.c_int32)]


class PeerExchange(C.Structure):
    _fields_ = [("world", C.c_int32), ("rank", C.c_int32), ("mailbox", C.c_void_p * 8), ("seq", C.c_void_p),
                ("status", C.c_void_p)]


class Targets(C.Structure):
    _fields_ = [("rgbds", f32p), ("depth_mask", C.c_void_p), ("term_mask", C.c_void_p),
                ("term_probs", f32p)]


class Prediction(C.Structure):
    _fields_ = [("rgbds", f32p), ("color_vars", f32p), ("depth_vars", f32p), ("term_probs", f32p)]


class Keyframes(C.Structure):
    _fields_ = [("num_frames", C.c_int32), ("height", C.c_int32), ("width", C.c_int32), ("reserved0", C.c_int32),
                ("c2ws", f32p), ("rgbd", f32p), ("frame_to_store", C.c_void_p),
                ("fx", C.c_float), ("fy", C.c_float), ("cx", C.c_float), ("cy", C.c_float)]


class TargetOut(C.Structure):
    _fields_ = [("ijs", C.c_void_p), ("c2ws", f32p), ("near", f32p), ("far", f32p), ("gt", f32p), ("rgbds", f32p),
                ("rgb_mask", C.c_void_p), ("depth_mask", C.c_void_p), ("term_probs", f32p), ("term_mask", C.c_void_p)]


_lib = None


def lib():
    """Load libngm_hip.so; fail loudly when the HIP extension has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} is missing: the HIP extension is not built. Run "
            "`python -m neural_graph_mapping_amd.build` (hipcc, gfx950). There is no CPU fallback.")
    L = C.CDLL(LIB_PATH)
    vp, i32, i64, f32 = C.c_void_p, C.c_int32, C.c_int64, C.c_float
    P = C.POINTER
    L.ngm_abi_version.restype = C.c_int
    L.ngm_permuto_fill_scales.argtypes = [P(FieldCfg)]
    L.ngm_permuto_fill_scales.restype = C.c_int
    L.ngm_last_error.restype = C.c_char_p
    L.ngm_device_info.argtypes = [P(C.c_int), C.c_char_p, C.c_int]
    L.ngm_sample_rays.argtypes = [P(RenderCfg), P(Rays), vp, vp, vp, vp]
    L.ngm_sample_rays_world.argtypes = [P(RenderCfg), P(Rays), vp, vp, vp, vp, vp]
    L.ngm_sample_rays_weighted.argtypes = [P(RenderCfg), P(Rays), i32, vp, vp, vp, vp, vp, vp]
    L.ngm_sample_rays_weighted.restype = C.c_int
    L.ngm_composite_fwd_packed.argtypes = [P(RenderCfg), i64, i32, vp, vp, vp, vp, vp, vp, vp, vp]
    L.ngm_sample_rays_world.restype = C.c_int
    L.ngm_composite_fwd_packed.restype = C.c_int
    L.ngm_field_eval_fwd.argtypes = [P(FieldCfg), P(Params), i32, i64, vp, vp, vp, vp, vp]
    L.ngm_encode_fwd.argtypes = [P(FieldCfg), P(Params), i32, i64, vp, vp, vp, vp, vp]
    L.ngm_encode_fwd.restype = C.c_int
    L.ngm_encode_bwd.argtypes = [P(FieldCfg), P(Params), i32, i64, vp, vp, vp, vp, P(Grads), vp, i64, vp]
    L.ngm_encode_bwd.restype = C.c_int
    L.ngm_encode_bwd_workspace.argtypes = [P(FieldCfg), i32, i64]
    L.ngm_encode_bwd_workspace.restype = i64
    L.ngm_field_eval_bwd.argtypes = [P(FieldCfg), P(Params), i32, i64, vp, vp, vp, vp, P(Grads), vp, i64, vp]
    L.ngm_field_eval_bwd_workspace.argtypes = [P(FieldCfg), i32, i64]
    L.ngm_field_eval_bwd_workspace.restype = i64
    L.ngm_field_eval_stash_bytes.argtypes = [P(FieldCfg), i32, i64]
    L.ngm_field_eval_stash_bytes.restype = i64
    L.ngm_field_eval_fwd_train.argtypes = [P(FieldCfg), P(Params), i32, i64, vp, vp, vp, vp, vp, i64, vp]
    L.ngm_field_eval_fwd_train.restype = C.c_int
    L.ngm_field_eval_bwd_stash.argtypes = [P(FieldCfg), P(Params), i32, i64, vp, vp, vp, vp, P(Grads), vp, i64, vp, i64, vp]
    L.ngm_field_eval_bwd_stash.restype = C.c_int
    L.ngm_composite_fwd.argtypes = [P(RenderCfg), i64, i32] + [vp] * 11 + [vp]
    L.ngm_composite_bwd.argtypes = [P(RenderCfg), i64, i32] + [vp] * 11 + [vp]
    L.ngm_render_workspace.argtypes = [P(FieldCfg), P(RenderCfg), i32, i32, i32]
    L.ngm_render_workspace.restype = i64
    L.ngm_render_fwd.argtypes = [P(FieldCfg), P(RenderCfg), P(Params), P(Rays), P(Targets),
                                 P(Prediction), vp, vp, i64, vp]
    L.ngm_render_bwd.argtypes = [P(FieldCfg), P(RenderCfg), P(Params), P(Rays), P(Targets),
                                 P(Prediction), vp, P(Grads), vp, vp, i64, vp]
    L.ngm_render_bwd_seeded.argtypes = [P(FieldCfg), P(RenderCfg), P(Params), P(Rays), vp, vp, vp,
                                        P(Grads), vp, i64, vp]
    L.ngm_render_bwd_seeded_vars.argtypes = [P(FieldCfg), P(RenderCfg), P(Params), P(Rays), P(Prediction), vp, vp, vp, vp, vp,
                                             P(Grads), vp, i64, vp]
    L.ngm_render_bwd_seeded_vars.restype = C.c_int
    L.ngm_render_bwd_adam.argtypes = [P(FieldCfg), P(RenderCfg), P(Params), P(Rays), P(Targets), P(Prediction), vp,
                                      P(Grads), P(AdamTensor), i32, P(AdamTensor), vp, i64, vp, f32, f32, f32, f32, f32,
                                      vp, vp, i64, vp]
    L.ngm_render_bwd_adam.restype = C.c_int
    L.ngm_render_read_samples.argtypes = [P(FieldCfg), P(RenderCfg), i32, i32, i32, vp, vp, vp, vp]
    L.ngm_adam_sparse.argtypes = [vp, vp, vp, i64, vp, i64, vp, i32, i64, i64, f32, f32, f32, f32, f32, vp]
    L.ngm_adam_sparse_multi.argtypes = [P(AdamTensor), i32, vp, i32, i64, vp, f32, f32, f32, f32, f32, i32, vp, vp]
    L.ngm_step_advance.argtypes = [vp, vp, vp]
    L.ngm_peer_mailbox_bytes.restype = i64
    L.ngm_peer_alloc.argtypes = [i64, P(C.c_void_p)]
    L.ngm_peer_free.argtypes = [vp]
    L.ngm_ipc_export.argtypes = [vp, C.c_char_p]
    L.ngm_ipc_open.argtypes = [C.c_char_p, P(C.c_void_p)]
    L.ngm_ipc_close.argtypes = [vp]
    L.ngm_loss_exchange.argtypes = [P(PeerExchange), vp, vp]
    L.ngm_peer_set_timeout.argtypes = [C.c_double]
    L.ngm_peer_set_timeout.restype = C.c_double
    L.ngm_adam_sparse_multi.restype = C.c_int
    L.ngm_step_advance.restype = C.c_int
    L.ngm_field_eval_knn.argtypes = [P(FieldCfg), P(Params), i32, i64, vp, vp, vp, i32, f32, f32, f32, vp, vp, i64, vp]
    L.ngm_field_eval_knn_workspace.argtypes = [i32, i64, i32]
    L.ngm_field_eval_knn_workspace.restype = i64
    L.ngm_render_eval_knn.argtypes = [P(FieldCfg), P(RenderCfg), P(Params), i32, vp, vp, P(Rays), i32, f32, f32, f32, i32,
                                      P(Prediction), vp, i64, vp]
    L.ngm_render_eval_knn.restype = C.c_int
    L.ngm_render_eval_knn_workspace.argtypes = [P(RenderCfg), i32, i32, i32]
    L.ngm_render_eval_knn_workspace.restype = i64
    L.ngm_target_visibility.argtypes = [P(Keyframes), i32, vp, i32, vp, f32, vp, vp, vp]
    L.ngm_target_rays.argtypes = [P(Keyframes), i32, i32, vp, f32, vp, vp, vp, P(TargetOut), vp]
    L.ngm_target_visibility.restype = C.c_int
    L.ngm_target_rays.restype = C.c_int
    L.ngm_target_sv_intersect.argtypes = [i32, i64, vp, vp, f32, vp, vp]
    L.ngm_target_sv_intersect.restype = C.c_int
    L.ngm_target_sv_rays.argtypes = [i32, i32, vp, f32, vp, vp, vp, i32, i32, f32, f32, f32, f32, P(TargetOut), vp]
    L.ngm_target_sv_rays.restype = C.c_int
    L.ngm_marching_cubes_workspace.argtypes = [i32, i32, i32]
    L.ngm_marching_cubes_workspace.restype = i64
    L.ngm_marching_cubes_count.argtypes = [vp, i32, i32, i32, f32, vp, vp, i64, vp]
    L.ngm_marching_cubes_emit.argtypes = [vp, i32, i32, i32, f32, vp, i64, vp, i64, vp, i64, vp]
    L.ngm_marching_cubes_tables.argtypes = [vp, vp]
    for name in ("ngm_marching_cubes_count", "ngm_marching_cubes_emit", "ngm_marching_cubes_tables"):
        getattr(L, name).restype = C.c_int
    L.ngm_profile_enable.argtypes = [i32]
    L.ngm_profile_read.argtypes = [i32, P(C.c_double), P(i64)]
    for name in ("ngm_profile_enable", "ngm_profile_reset", "ngm_profile_read"):
        getattr(L, name).restype = C.c_int
    for name in ("ngm_sample_rays", "ngm_field_eval_fwd", "ngm_field_eval_bwd", "ngm_composite_fwd",
                 "ngm_composite_bwd", "ngm_render_fwd", "ngm_render_bwd", "ngm_render_bwd_seeded",
                 "ngm_render_read_samples", "ngm_adam_sparse", "ngm_field_eval_knn", "ngm_device_info"):
        getattr(L, name).restype = C.c_int
    _lib = L
    return L


EXPORTED = ["ngm_abi_version", "ngm_last_error", "ngm_device_info", "ngm_permuto_fill_scales", "ngm_sample_rays", "ngm_sample_rays_world", "ngm_sample_rays_weighted",
            "ngm_composite_fwd_packed",
            "ngm_field_eval_fwd", "ngm_encode_fwd", "ngm_encode_bwd", "ngm_encode_bwd_workspace", "ngm_field_eval_bwd", "ngm_field_eval_bwd_workspace",
            "ngm_field_eval_stash_bytes", "ngm_field_eval_fwd_train", "ngm_field_eval_bwd_stash",
            "ngm_composite_fwd", "ngm_composite_bwd", "ngm_render_workspace", "ngm_render_fwd",
            "ngm_render_bwd", "ngm_render_bwd_adam", "ngm_render_bwd_seeded", "ngm_render_bwd_seeded_vars", "ngm_render_read_samples", "ngm_adam_sparse",
            "ngm_field_eval_knn", "ngm_field_eval_knn_workspace", "ngm_render_eval_knn", "ngm_render_eval_knn_workspace", "ngm_adam_sparse_multi", "ngm_step_advance", "ngm_profile_enable", "ngm_profile_reset", "ngm_profile_read",
            "ngm_debug_phase_cycles", "ngm_debug_fwd_phase_cycles", "ngm_debug_last_bwd_variant", "ngm_debug_last_matmul", "ngm_debug_last_fwd_one_tile", "ngm_debug_last_comp_fused", "ngm_debug_disable_fused_comp", "ngm_debug_stash_mode", "ngm_debug_last_stash_mode", "ngm_target_visibility", "ngm_target_rays", "ngm_target_sv_intersect", "ngm_target_sv_rays",
            "ngm_peer_mailbox_bytes", "ngm_peer_alloc", "ngm_peer_free", "ngm_ipc_export", "ngm_ipc_open", "ngm_ipc_close", "ngm_loss_exchange", "ngm_peer_set_timeout",
            "ngm_marching_cubes_workspace", "ngm_marching_cubes_count", "ngm_marching_cubes_emit", "ngm_marching_cubes_tables"]

# parameters that never receive a gradient (the CUDA package gives none to the per-level shifts either;
# torch.optim.Adam skips grad-less parameters, so the sparse Adam must skip them too)
NO_GRAD_PARAMS = frozenset({"_encoding.random_shift_per_level"})

KERNEL_IDS = dict(render_fwd=0, stash_bwd=1, field_bwd=2, grad_reduce=3, adam=4, points_fwd=5, composite_fwd=6,
                  composite_bwd=7, hash_grad=8, hash_reduce=9, loss_reduce=10, knn_assign=11, knn_eval=12, sampler=13)


NGM_OK, NGM_E_INVALID, NGM_E_UNSUPPORTED, NGM_E_WORKSPACE, NGM_E_HIP = 0, -1, -2, -3, -4     # include/ngm_hip.h ngm_status


class NgmError(RuntimeError):
    """A C-ABI call returned a negative status; `code` = the NGM_E_* value (None: raised by the host layer itself)."""

    def __init__(self, msg, code=None):
        super().__init__(msg)
        self.code = code


def check(rc, what=""):
    if rc != 0:
        msg = lib().ngm_last_error().decode(errors="replace")
        raise NgmError(f"{what} failed with status {rc}: {msg}", code=int(rc))


# ------------------------------------------------------------------------------------------------
# struct builders from plain python values / raw device addresses
# ------------------------------------------------------------------------------------------------
SKIP = {"no": 0, "add": 1, "concat": 2}
STASH = {"full": 0, "half": 1}               # ngm_activation_stash
HASH_ATOMICS = {"exact": 0, "float": 1}      # ngm_hash_grad_atomics
MATMUL = {"f32": 0, "bf16x3": 1, "auto": 2}


def field_cfg(encoding="fourier", dim_enc=64, raw_coords=True, num_octaves=8, start_octave=0,
              num_layers=2, dim_hidden=None, dim_out=4, scale_mode="unit_cube", field_radius=1.0,
              nr_levels=16, nr_feat_per_level=2, log2_hashmap_size=12, coarsest_scale=1.0, finest_scale=1e-4,
              skip_mode="no", matmul_mode="f32", resolution=32, num_components=64, tri_mode="sum"):
    if skip_mode not in SKIP:
        raise NotImplementedError(f"skip_mode={skip_mode!r}: 'no', 'add' and 'concat' have kernels; the reference's own "
                                  "constructor raises for 'rezero' (models.py:131-132)")
    if encoding == "nerf":
        dim_enc = 6 * num_octaves
    if encoding == "none":
        dim_enc = 3
    if encoding == "permuto":
        dim_enc = nr_levels * nr_feat_per_level
    if encoding == "triplane":
        dim_enc = num_components * (3 if tri_mode == "concat" else 1)      # positional_encodings.py:119-126
    if dim_hidden is None:
        dim_hidden = dim_enc
    fc = FieldCfg(ENC[encoding], dim_enc, int(bool(raw_coords)), num_octaves, start_octave,
                  num_layers, dim_hidden, dim_out, SCALE[scale_mode], float(field_radius), nr_levels,
                  nr_feat_per_level, log2_hashmap_size, float(coarsest_scale), float(finest_scale))
    fc.skip_mode = SKIP[skip_mode]
    fc.matmul_mode = MATMUL[matmul_mode]
    fc.tri_resolution, fc.tri_mode = int(resolution), TRI[tri_mode]
    if encoding == "permuto":
        import math
        import numpy as np
        sig = np.geomspace(coarsest_scale, finest_scale, num=nr_levels)          # positional_encodings.py:50
        for lvl in range(min(nr_levels, 16)):
            for i in range(3):
                fc.level_scale[3 * lvl + i] = np.float32((1.0 / math.sqrt((i + 1) * (i + 2))) / sig[lvl])
    return fc


def param_names(fc: FieldCfg):
    names = []
    if fc.encoding == ENC["fourier"]:
        names.append("_encoding._linear.weight")
    if fc.encoding == ENC["permuto"]:
        names += ["_encoding.lattice_values", "_encoding.random_shift_per_level"]
    if fc.encoding == ENC["triplane"]:
        names.append("_encoding.plane_coef")
    for i in range(fc.num_layers + 1):
        names += [f"_linears.{i}.weight", f"_linears.{i}.bias"]
    return names


def param_shapes(fc: FieldCfg):
    shapes = {}
    if fc.encoding == ENC["fourier"]:
        shapes["_encoding._linear.weight"] = ((fc.dim_enc - 3) if fc.raw_coords else fc.dim_enc, 3)
    if fc.encoding == ENC["permuto"]:
        shapes["_encoding.lattice_values"] = (fc.nr_levels, 2 ** fc.log2_hashmap_size, fc.nr_feat_per_level)
        shapes["_encoding.random_shift_per_level"] = (fc.nr_levels, 3)
    if fc.encoding == ENC["triplane"]:
        comps = fc.dim_enc // 3 if fc.tri_mode == TRI["concat"] else fc.dim_enc
        shapes["_encoding.plane_coef"] = (3, comps, fc.tri_resolution, fc.tri_resolution)
    for i in range(fc.num_layers + 1):
        din = fc.dim_enc if i == 0 else fc.dim_hidden + (fc.dim_enc if fc.skip_mode == SKIP["concat"] else 0)   # models.py:115-119
        dout = fc.dim_out if i == fc.num_layers else fc.dim_hidden
        shapes[f"_linears.{i}.weight"] = (dout, din)
        shapes[f"_linears.{i}.bias"] = (dout,)
    return shapes


def _fill_ptrs(struct, fc, ptrs, strides):
    """ptrs/strides: dict name -> device address / element stride between fields."""
    if fc.encoding == ENC["fourier"]:
        struct.enc_w = ptrs["_encoding._linear.weight"]
        struct.enc_w_stride = strides["_encoding._linear.weight"]
    if fc.encoding == ENC["permuto"]:
        struct.lattice = ptrs["_encoding.lattice_values"]
        struct.lattice_stride = strides["_encoding.lattice_values"]
        if hasattr(struct, "shift"):
            struct.shift = ptrs["_encoding.random_shift_per_level"]
            struct.shift_stride = strides["_encoding.random_shift_per_level"]
    if fc.encoding == ENC["triplane"]:
        struct.planes = ptrs["_encoding.plane_coef"]
        struct.planes_stride = strides["_encoding.plane_coef"]
    for i in range(fc.num_layers + 1):
        struct.w[i] = ptrs[f"_linears.{i}.weight"]
        struct.w_stride[i] = strides[f"_linears.{i}.weight"]
        struct.b[i] = ptrs[f"_linears.{i}.bias"]
        struct.b_stride[i] = strides[f"_linears.{i}.bias"]
    return struct


DTYPE = {"float32": 0, "bfloat16": 1, "float16": 2}      # ngm_param_dtype: storage type of the weights


def params_struct(fc, ptrs, strides, field_index=None, dtype=0):
    p = _fill_ptrs(Params(), fc, ptrs, strides)
    p.field_index = field_index
    p.dtype = dtype
    return p


def grads_struct(fc, ptrs, strides):
    return _fill_ptrs(Grads(), fc, ptrs, strides)


def render_cfg(geometry_mode="nrgbd", num_samples_coarse=8, num_samples_guided=16,
               geometry_factor=20.0, color_factor=1.0, truncation_distance=0.1,
               range_depth_guided=None, fx=554.2562584220408, fy=554.2562584220408, cx=319.5,
               cy=239.5, w_termination=0.0, w_photometric=1.0, w_depth=1.0, w_freespace=40.0,
               w_tsdf=50.0, huber_delta=0.05, term_threshold=0.8, overwrite_behind_camera=True,
               photometric_loss="l1", depth_loss="huber"):
    if photometric_loss not in PHOTO:
        raise NotImplementedError(f"photometric_loss {photometric_loss!r}: the kernels build {sorted(PHOTO)} (losses.py:26-36)")
    if depth_loss not in DEPTH:
        raise NotImplementedError(f"depth_loss {depth_loss!r}: the kernels build {sorted(DEPTH)} (losses.py:60-75)")
    if range_depth_guided is None:
        range_depth_guided = truncation_distance
    return RenderCfg(GEO[geometry_mode], num_samples_coarse, num_samples_guided, int(bool(overwrite_behind_camera)),
                     geometry_factor,
                     color_factor, truncation_distance, range_depth_guided, fx, fy, cx, cy,
                     w_termination, w_photometric, w_depth, w_freespace, w_tsdf, huber_delta,
                     term_threshold, PHOTO[photometric_loss], DEPTH[depth_loss])
