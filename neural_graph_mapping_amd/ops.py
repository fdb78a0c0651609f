"""PyTorch-ROCm custom ops over the C ABI (include/ngm_hip.h).

torch is plumbing here (device memory, streams, autograd glue); every op below is a hand-written
gfx950 kernel reached through ``_capi``.  There is NO CPU / eager fallback: tensors must live on a
ROCm device ("cuda" under PyTorch-ROCm) and the HIP library must be built, otherwise the call raises.
"""
import ctypes as C
from typing import Dict, List, Optional, Tuple

import torch

from . import _capi as K


# ------------------------------------------------------------------------------------------------
# torch.library registration: every op below is visible to the dispatcher as torch.ops.ngm355.<name>
# (schema, fake/meta shapes, autograd formula) and calls the C ABI underneath.  "cuda" (= ROCm) kernels only:
# there is no CPU registration on purpose.  Configuration structs travel as small uint8 CPU tensors (the bytes
# of ngm_field_cfg / ngm_render_cfg), stacked parameter dictionaries as Tensor[] in K.param_names() order.
# ------------------------------------------------------------------------------------------------
NS = "ngm355"


_BLOBS: Dict[bytes, torch.Tensor] = {}       # struct bytes -> its uint8 CPU tensor (a renderer uses a handful of distinct structs)
_STRUCTS: Dict[int, object] = {}             # id(blob tensor) -> decoded ctypes struct (the blobs above live for the process)


def cfg_blob(struct) -> torch.Tensor:
    raw = bytes(struct)
    t = _BLOBS.get(raw)
    if t is None:
        if len(_BLOBS) > 4096:
            _BLOBS.clear(); _STRUCTS.clear()
        t = _BLOBS[raw] = torch.frombuffer(bytearray(raw), dtype=torch.uint8)
        _STRUCTS[id(t)] = type(struct).from_buffer_copy(raw)
    return t


def _decode(blob: torch.Tensor, cls):
    s = _STRUCTS.get(id(blob))
    if s is not None and isinstance(s, cls):
        return s                             # read-only use below: the cached struct is never mutated
    return cls.from_buffer_copy(blob.numpy().tobytes())


def _field_cfg(blob: torch.Tensor) -> "K.FieldCfg":
    return _decode(blob, K.FieldCfg)


def _render_cfg(blob: torch.Tensor) -> "K.RenderCfg":
    return _decode(blob, K.RenderCfg)


def _op(name, mutates_args=()):
    return torch.library.custom_op(f"{NS}::{name}", mutates_args=mutates_args, device_types="cuda")


def _require_gpu(*tensors):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise RuntimeError("neural_graph_mapping_amd ops need ROCm device tensors; there is no CPU fallback "
                               "(the CPU restatement under oracle/ is test infrastructure only)")


def _ptr(t: Optional[torch.Tensor]):
    return None if t is None else t.data_ptr()


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _f32c(t, name="tensor"):
    if t is None:
        return None
    if t.dtype != torch.float32:
        raise TypeError(f"{name} must be float32, got {t.dtype}")
    return t.contiguous()


# ------------------------------------------------------------------------------------------------
# parameter plumbing: dict of stacked (N, ...) tensors <-> ngm_params / ngm_grads
# ------------------------------------------------------------------------------------------------
_TORCH_DT = {torch.float32: 0, torch.bfloat16: 1, torch.float16: 2}


def params_struct(fc: K.FieldCfg, params: Dict[str, torch.Tensor], field_index: Optional[torch.Tensor] = None):
    """dict of stacked (N, ...) tensors -> ngm_params.  The weights may be stored as float32, bfloat16 or float16 (all
    of them in the same type; the constant hash shifts stay float32): ngm_params.dtype tells the kernels, which widen to
    fp32 when they stage a field (storage only: the arithmetic is fp32 either way)."""
    ptrs, strides = {}, {}
    dts = set()
    for n, shp in K.param_shapes(fc).items():
        if n not in params:
            raise KeyError(f"missing parameter tensor '{n}'")
        t = params[n]
        _require_gpu(t)
        const = n in K.NO_GRAD_PARAMS
        if t.dtype not in _TORCH_DT or (const and t.dtype != torch.float32) or tuple(t.shape[1:]) != tuple(shp):
            raise ValueError(f"parameter '{n}' must be {'float32' if const else 'float32 / bfloat16 / float16'} (N,{shp}), "
                             f"got {t.dtype} {tuple(t.shape)}")
        if not const:
            dts.add(t.dtype)
        inner = t[0] if t.shape[0] > 0 else t
        if t.shape[0] > 0 and not inner.is_contiguous():
            raise ValueError(f"parameter '{n}' rows must be contiguous")
        ptrs[n] = t.data_ptr()
        strides[n] = t.stride(0) if t.shape[0] > 1 else int(torch.tensor(shp).prod())
    if len(dts) != 1:
        raise ValueError(f"the weight tensors must share one storage type, got {sorted(str(d) for d in dts)}")
    fi = None
    if field_index is not None:
        _require_gpu(field_index)
        if field_index.dtype != torch.int64:
            raise TypeError("field_index must be int64")
        fi = field_index.contiguous().data_ptr()
    ps = K.params_struct(fc, ptrs, strides, fi, _TORCH_DT[dts.pop()])
    sd = params.get("_neus_sd")
    if sd is not None:                 # per-field standard deviation of the neus geometry mode (fused render only)
        _require_gpu(sd)
        if sd.dtype != torch.float32 or sd.dim() != 1:
            raise ValueError("'_neus_sd' must be float32 (N,)")
        ps.neus_sd, ps.neus_sd_stride = sd.data_ptr(), sd.stride(0) if sd.shape[0] > 1 else 1
    return ps


def alloc_grads(fc: K.FieldCfg, F: int, device, flat: Optional[torch.Tensor] = None):
    """Gradient tensors (F, ...) per parameter name; views into one flat (F, P) arena."""
    shapes = K.param_shapes(fc)
    total = sum(int(torch.tensor(s).prod()) for s in shapes.values())
    if flat is None:
        flat = torch.zeros(F, total, device=device, dtype=torch.float32)
    grads, ptrs, strides, off = {}, {}, {}, 0
    for n, shp in shapes.items():
        numel = int(torch.tensor(shp).prod())
        grads[n] = flat[:, off:off + numel].view(F, *shp)
        ptrs[n] = flat.data_ptr() + off * 4
        strides[n] = flat.stride(0)
        off += numel
    return grads, K.grads_struct(fc, ptrs, strides), flat


def alloc_grads_separate(fc: K.FieldCfg, F: int, device):
    """One tensor per parameter name (custom-op outputs may not alias each other, so no shared arena here)."""
    shapes = K.param_shapes(fc)
    grads = {n: torch.zeros(F, *shp, device=device, dtype=torch.float32) for n, shp in shapes.items()}
    ptrs = {n: g.data_ptr() for n, g in grads.items()}
    strides = {n: int(torch.tensor(shp).prod()) for n, shp in shapes.items()}
    return grads, K.grads_struct(fc, ptrs, strides)


# ------------------------------------------------------------------------------------------------
# K2+K3: NeuralFieldSet.forward(use_vmap=True) with autograd
# ------------------------------------------------------------------------------------------------
@_op("field_eval")
def _field_eval_op(fcfg: torch.Tensor, points: torch.Tensor, pos: Optional[torch.Tensor], quat: Optional[torch.Tensor],
                   params: List[torch.Tensor]) -> torch.Tensor:
    fc = _field_cfg(fcfg)
    pd = dict(zip(K.param_names(fc), params))
    F, P, _ = points.shape
    out = torch.empty(F, P, 4, device=points.device, dtype=torch.float32)
    ps = params_struct(fc, pd)
    K.check(K.lib().ngm_field_eval_fwd(C.byref(fc), C.byref(ps), F, P, _ptr(points), _ptr(_f32c(pos)),
                                       _ptr(_f32c(quat)), _ptr(out), _stream()), "ngm_field_eval_fwd")
    return out


@_field_eval_op.register_fake
def _(fcfg, points, pos, quat, params):
    return points.new_empty(points.shape[0], points.shape[1], 4)


@_op("field_eval_bwd")
def _field_eval_bwd_op(fcfg: torch.Tensor, points: torch.Tensor, pos: Optional[torch.Tensor], quat: Optional[torch.Tensor],
                       d_out: torch.Tensor, params: List[torch.Tensor]) -> List[torch.Tensor]:
    fc = _field_cfg(fcfg)
    names = K.param_names(fc)
    pd = dict(zip(names, params))
    F, P, _ = points.shape
    grads, gs = alloc_grads_separate(fc, F, points.device)
    ps = params_struct(fc, pd)
    L = K.lib()
    wsb = L.ngm_field_eval_bwd_workspace(C.byref(fc), F, P)
    ws = torch.empty(wsb, device=points.device, dtype=torch.uint8)
    K.check(L.ngm_field_eval_bwd(C.byref(fc), C.byref(ps), F, P, _ptr(points), _ptr(_f32c(pos)), _ptr(_f32c(quat)),
                                 _ptr(_f32c(d_out, "d_out")), C.byref(gs), _ptr(ws), wsb, _stream()), "ngm_field_eval_bwd")
    return [grads[n] for n in names]


@_field_eval_bwd_op.register_fake
def _(fcfg, points, pos, quat, d_out, params):
    return [torch.empty_like(p, dtype=torch.float32) for p in params]      # gradients are fp32 whatever the storage dtype


def _field_eval_setup(ctx, inputs, output):
    fcfg, points, pos, quat, params = inputs
    ctx.fcfg, ctx.has_pose, ctx.n = fcfg, pos is not None, len(params)
    ctx.save_for_backward(points, *([pos, quat] if pos is not None else []), *params)


def _field_eval_backward(ctx, d_out):
    points, *rest = ctx.saved_tensors
    pos, quat = (rest[0], rest[1]) if ctx.has_pose else (None, None)
    params = rest[2:] if ctx.has_pose else rest
    grads = torch.ops.ngm355.field_eval_bwd(ctx.fcfg, points, pos, quat, d_out.contiguous(), list(params))
    return None, None, None, None, grads


_field_eval_op.register_autograd(_field_eval_backward, setup_context=_field_eval_setup)


# ---- the same under autograd with an activation stash (ABI 11): the forward writes the hidden activations, the backward is the
# fused training step's MLP backward (k_field_bwd_b3 in point mode) instead of the recomputing fp32 kernel
@_op("field_eval_train")
def _field_eval_train_op(fcfg: torch.Tensor, points: torch.Tensor, pos: Optional[torch.Tensor], quat: Optional[torch.Tensor],
                         params: List[torch.Tensor]) -> Tuple[torch.Tensor, torch.Tensor]:
    fc = _field_cfg(fcfg)
    pd = dict(zip(K.param_names(fc), params))
    F, P, _ = points.shape
    L = K.lib()
    out = torch.empty(F, P, 4, device=points.device, dtype=torch.float32)
    sb = int(L.ngm_field_eval_stash_bytes(C.byref(fc), F, P))
    if sb <= 0:
        raise K.NgmError("field_eval_train: this configuration has no stash-reading backward (ngm_field_eval_stash_bytes == 0)")
    stash = torch.empty(sb, device=points.device, dtype=torch.uint8)      # an output: lives until the backward has run
    ps = params_struct(fc, pd)
    K.check(L.ngm_field_eval_fwd_train(C.byref(fc), C.byref(ps), F, P, _ptr(points), _ptr(_f32c(pos)), _ptr(_f32c(quat)), _ptr(out),
                                       _ptr(stash), sb, _stream()), "ngm_field_eval_fwd_train")
    return out, stash


@_field_eval_train_op.register_fake
def _(fcfg, points, pos, quat, params):
    return points.new_empty(points.shape[0], points.shape[1], 4), points.new_empty(0, dtype=torch.uint8)


@_op("field_eval_bwd_stash")
def _field_eval_bwd_stash_op(fcfg: torch.Tensor, points: torch.Tensor, pos: Optional[torch.Tensor], quat: Optional[torch.Tensor],
                             d_out: torch.Tensor, stash: torch.Tensor, params: List[torch.Tensor]) -> List[torch.Tensor]:
    fc = _field_cfg(fcfg)
    names = K.param_names(fc)
    pd = dict(zip(names, params))
    F, P, _ = points.shape
    grads, gs = alloc_grads_separate(fc, F, points.device)
    ps = params_struct(fc, pd)
    L = K.lib()
    wsb = L.ngm_field_eval_bwd_workspace(C.byref(fc), F, P)
    ws = torch.empty(wsb, device=points.device, dtype=torch.uint8)
    K.check(L.ngm_field_eval_bwd_stash(C.byref(fc), C.byref(ps), F, P, _ptr(points), _ptr(_f32c(pos)), _ptr(_f32c(quat)),
                                       _ptr(_f32c(d_out, "d_out")), C.byref(gs), _ptr(stash), stash.numel(), _ptr(ws), wsb, _stream()),
            "ngm_field_eval_bwd_stash")
    return [grads[n] for n in names]


@_field_eval_bwd_stash_op.register_fake
def _(fcfg, points, pos, quat, d_out, stash, params):
    return [torch.empty_like(p, dtype=torch.float32) for p in params]


def _field_eval_train_setup(ctx, inputs, output):
    fcfg, points, pos, quat, params = inputs
    ctx.fcfg, ctx.has_pose, ctx.n = fcfg, pos is not None, len(params)
    ctx.save_for_backward(points, output[1], *([pos, quat] if pos is not None else []), *params)
    ctx.mark_non_differentiable(output[1])
    ctx.set_materialize_grads(False)         # no zero "gradient" of the stash's size (a 2 GB fill per backward at 4 M points)


def _field_eval_train_backward(ctx, d_out, _d_stash):
    if d_out is None:
        return None, None, None, None, None
    points, stash, *rest = ctx.saved_tensors
    pos, quat = (rest[0], rest[1]) if ctx.has_pose else (None, None)
    params = rest[2:] if ctx.has_pose else rest
    grads = torch.ops.ngm355.field_eval_bwd_stash(ctx.fcfg, points, pos, quat, d_out.contiguous(), stash, list(params))
    return None, None, None, None, grads


_field_eval_train_op.register_autograd(_field_eval_train_backward, setup_context=_field_eval_train_setup)

# Above this many bytes of stash (256 B per point and hidden layer) a differentiable field_eval keeps the recomputing backward
# (no stash is held between forward and backward).  The reference's own training batch (32 fields x 512 rays x 24 samples)
# stashes 0.2 GB; the MI355X has 288 GB.
FIELD_EVAL_STASH_MAX_BYTES = 16 << 30


def field_eval(fc: K.FieldCfg, params: Dict[str, torch.Tensor], points, pos=None, quat=None):
    """(F,P,3) points -> (F,P,4); differentiable w.r.t. the parameters (not the points/poses).
    Dispatches through torch.ops.ngm355.field_eval; when a gradient will be asked for and the network has a stash-reading
    backward (ngm_field_eval_stash_bytes > 0) through torch.ops.ngm355.field_eval_train: same outputs bit for bit, the hidden
    activations kept for the backward (k_field_bwd_b3 in point mode) instead of being recomputed there."""
    names = K.param_names(fc)
    plist = [params[n] for n in names]
    _require_gpu(points, pos, quat, *plist)
    points = _f32c(points, "points")
    if torch.is_grad_enabled() and any(p.requires_grad for p in plist) and points.shape[1] > 0:
        sb = int(K.lib().ngm_field_eval_stash_bytes(C.byref(fc), points.shape[0], points.shape[1]))
        if 0 < sb <= FIELD_EVAL_STASH_MAX_BYTES:
            return torch.ops.ngm355.field_eval_train(cfg_blob(fc), points, pos, quat, plist)[0]
    return torch.ops.ngm355.field_eval(cfg_blob(fc), points, pos, quat, plist)


def encode(fc: K.FieldCfg, params: Dict[str, torch.Tensor], points, pos=None, quat=None):
    """The positional encoding alone (positional_encodings.py:19-66, 164-276): (F,P,3) points -> (F,P,dim_enc), no autograd.
    Standalone stage entry point (ngm_encode_fwd) -- the fused kernels never materialise this tensor."""
    names = K.param_names(fc)
    plist = [params[n] for n in names]
    _require_gpu(points, pos, quat, *plist)
    pts = _f32c(points, "points")
    F, P = pts.shape[0], pts.shape[1]
    out = torch.empty(F, P, fc.dim_enc, device=pts.device, dtype=torch.float32)
    ps = params_struct(fc, dict(zip(names, plist)))
    K.check(K.lib().ngm_encode_fwd(C.byref(fc), C.byref(ps), F, P, _ptr(pts), _ptr(None if pos is None else _f32c(pos)),
                                   _ptr(None if quat is None else _f32c(quat)), _ptr(out), _stream()), "ngm_encode_fwd")
    return out


def encode_bwd(fc: K.FieldCfg, params: Dict[str, torch.Tensor], points, d_enc, pos=None, quat=None) -> Dict[str, torch.Tensor]:
    """Backward of `encode` w.r.t. the encoding's own parameters (ngm_encode_bwd): d_enc (F,P,dim_enc) ->
    {"_encoding._linear.weight": (F, dim_enc - 3, 3)} (Fourier) or {"_encoding.lattice_values": (F, L, T, 2)} (hash);
    {} for the parameter-free encodings."""
    names = K.param_names(fc)
    plist = [params[n] for n in names]
    _require_gpu(points, d_enc, pos, quat, *plist)
    pts, de = _f32c(points, "points"), _f32c(d_enc, "d_enc")
    F, P = pts.shape[0], pts.shape[1]
    if tuple(de.shape) != (F, P, fc.dim_enc):
        raise ValueError(f"d_enc must be (F, P, dim_enc) = {(F, P, fc.dim_enc)}, got {tuple(de.shape)}")
    ps = params_struct(fc, dict(zip(names, plist)))
    out, g = {}, K.Grads()
    if fc.encoding == K.ENC["fourier"]:
        t = out["_encoding._linear.weight"] = torch.empty(F, *K.param_shapes(fc)["_encoding._linear.weight"], device=pts.device)
        g.enc_w, g.enc_w_stride = t.data_ptr(), t.stride(0)
    elif fc.encoding == K.ENC["permuto"]:
        t = out["_encoding.lattice_values"] = torch.empty(F, *K.param_shapes(fc)["_encoding.lattice_values"], device=pts.device)
        g.lattice, g.lattice_stride = t.data_ptr(), t.stride(0)
    L = K.lib()
    wsb = L.ngm_encode_bwd_workspace(C.byref(fc), F, P)
    ws = torch.empty(max(int(wsb), 256), device=pts.device, dtype=torch.uint8)
    K.check(L.ngm_encode_bwd(C.byref(fc), C.byref(ps), F, P, _ptr(pts), _ptr(None if pos is None else _f32c(pos)),
                             _ptr(None if quat is None else _f32c(quat)), _ptr(de), C.byref(g), _ptr(ws), ws.numel(), _stream()),
            "ngm_encode_bwd")
    return out


@_op("field_eval_knn")
def _field_eval_knn_op(fcfg: torch.Tensor, points: torch.Tensor, pos: torch.Tensor, quat: torch.Tensor,
                       params: List[torch.Tensor], num_knn: int, distance_factor: float, outside_value: float,
                       field_index: Optional[torch.Tensor], mask_radius: float) -> torch.Tensor:
    fc = _field_cfg(fcfg)
    P, NF = points.shape[0], pos.shape[0]
    out = torch.empty(P, 4, device=points.device, dtype=torch.float32)
    if P == 0:
        return out
    ps = params_struct(fc, dict(zip(K.param_names(fc), params)), field_index)
    L = K.lib()
    wsb = L.ngm_field_eval_knn_workspace(NF, P, num_knn)
    ws = torch.empty(wsb, device=points.device, dtype=torch.uint8)
    K.check(L.ngm_field_eval_knn(C.byref(fc), C.byref(ps), NF, P, _ptr(points), _ptr(_f32c(pos)), _ptr(_f32c(quat)),
                                 num_knn, distance_factor, outside_value, mask_radius, _ptr(out), _ptr(ws), wsb, _stream()),
            "ngm_field_eval_knn")
    return out


@_field_eval_knn_op.register_fake
def _(fcfg, points, pos, quat, params, num_knn, distance_factor, outside_value, field_index, mask_radius):
    return points.new_empty(points.shape[0], 4)


def field_eval_knn(fc, params, points, pos, quat, num_knn=2, distance_factor=10.0, outside_value=1.0,
                   field_index=None, mask_radius=None):
    """models.py:347-405: kNN-blended evaluation of world points (P,3) over all fields -> (P,4).
    `mask_radius` = the `field_radius` argument of the reference's forward (models.py:293, 368): which points count as
    inside a field; the local coordinates are scaled with the model's own radius (fc.field_radius) either way.
    Dispatches through torch.ops.ngm355.field_eval_knn."""
    plist = [params[n] for n in K.param_names(fc)]
    _require_gpu(points, pos, quat, *plist)
    return torch.ops.ngm355.field_eval_knn(cfg_blob(fc), _f32c(points.reshape(-1, 3)), pos, quat, plist, int(num_knn),
                                           float(distance_factor), float(outside_value), field_index,
                                           float(mask_radius) if mask_radius else 0.0)


# ------------------------------------------------------------------------------------------------
# K1: sampler
# ------------------------------------------------------------------------------------------------
_LIN_CACHE = {}


def linspace_table(n: int, device):
    """torch.linspace(0,1,n+1) of camera.py:271 as a device table for the kernels.  Computed by torch on
    the host (bit-identical to the CPU reference / oracle; ROCm's device linspace may differ in the
    last bit for non power-of-two n) and cached per (n, device)."""
    key = (int(n), str(device))
    if key not in _LIN_CACHE:
        _LIN_CACHE[key] = torch.linspace(0.0, 1.0, steps=n + 1, dtype=torch.float32).to(device)
    return _LIN_CACHE[key]


def make_rays(rc: K.RenderCfg, ijs, c2ws, near, far, gt, pos, quat, u_coarse=None, u_guided=None, seed=0, offset=0,
              near_const=0.0, far_const=8.0, keep=None, pose_index=None, philox_offset_dev=None, philox_autoinc=False):
    """Build an ngm_rays record; `keep` (list) receives every temporary that must outlive the launch."""
    keep = keep if keep is not None else []
    _require_gpu(ijs, c2ws, near, far, gt, pos, quat, u_coarse, u_guided)
    if ijs.dtype != torch.int64:
        raise TypeError("ijs must be int64 (row, col)")
    ijs = ijs.contiguous()
    F, R = (ijs.shape[0], ijs.shape[1]) if ijs.dim() == 3 else (1, ijs.shape[0])
    per_ray = 0 if c2ws.dim() == 2 else 1
    c2ws = _f32c(c2ws)
    if per_ray and c2ws.numel() != F * R * 16:
        c2ws = c2ws.expand(F, R, 4, 4).contiguous()
    dev = ijs.device
    lin_c = linspace_table(rc.num_samples_coarse, dev)
    lin_g = linspace_table(rc.num_samples_guided, dev) if rc.num_samples_guided > 0 else None
    ts = [ijs, c2ws, _f32c(near), _f32c(far), _f32c(gt), _f32c(pos), _f32c(quat), _f32c(u_coarse), _f32c(u_guided),
          lin_c, lin_g]
    if pose_index is not None:
        pose_index = pose_index.contiguous()
        _require_gpu(pose_index)
    keep.extend(ts + [pose_index, philox_offset_dev])
    return K.Rays(F, R, _ptr(ts[0]), _ptr(ts[1]), per_ray, 1 if (philox_autoinc and philox_offset_dev is not None) else 0, _ptr(ts[2]), _ptr(ts[3]), _ptr(ts[4]),
                  float(near_const), float(far_const), _ptr(ts[5]), _ptr(ts[6]), _ptr(ts[7]), _ptr(ts[8]),
                  _ptr(ts[9]), _ptr(ts[10]), int(seed), int(offset), _ptr(pose_index), _ptr(philox_offset_dev))


@_op("sample_rays")
def _sample_rays_op(rcfg: torch.Tensor, ijs: torch.Tensor, c2ws: Optional[torch.Tensor], near: Optional[torch.Tensor],
                    far: Optional[torch.Tensor], gt: Optional[torch.Tensor], u_coarse: Optional[torch.Tensor],
                    u_guided: Optional[torch.Tensor], seed: int, near_const: float, far_const: float,
                    num_samples: int) -> List[torch.Tensor]:
    """[points_cam (F,R,S,3), points_world (F,R,S,3) or empty (no c2ws), distances (F,R,S), dirs (F,R,3)]"""
    rc = _render_cfg(rcfg)
    dev = ijs.device
    F, R = ijs.shape[0], ijs.shape[1]
    world = c2ws is not None
    pos, quat = torch.zeros(F, 3, device=dev), torch.zeros(F, 4, device=dev)
    keep = []
    rays = make_rays(rc, ijs, c2ws if world else torch.eye(4, device=dev), near, far, gt, pos, quat, u_coarse, u_guided,
                     seed, near_const=near_const, far_const=far_const, keep=keep)
    S = rc.num_samples_coarse + (rc.num_samples_guided if gt is not None else 0)
    assert S == num_samples
    pc = torch.empty(F, R, S, 3, device=dev)
    pw = torch.empty((F, R, S, 3) if world else (0,), device=dev)
    dist = torch.empty(F, R, S, device=dev)
    dirs = torch.empty(F, R, 3, device=dev)
    K.check(K.lib().ngm_sample_rays_world(C.byref(rc), C.byref(rays), _ptr(pc), _ptr(pw) if world else None, _ptr(dist),
                                          _ptr(dirs), _stream()), "ngm_sample_rays_world")
    return [pc, pw, dist, dirs]


@_sample_rays_op.register_fake
def _(rcfg, ijs, c2ws, near, far, gt, u_coarse, u_guided, seed, near_const, far_const, num_samples):
    F, R, S = ijs.shape[0], ijs.shape[1], num_samples
    f = lambda *shape: torch.empty(*shape, device=ijs.device, dtype=torch.float32)
    return [f(F, R, S, 3), f(F, R, S, 3) if c2ws is not None else f(0), f(F, R, S), f(F, R, 3)]


def sample_rays(rc: K.RenderCfg, ijs, near, far, gt=None, u_coarse=None, u_guided=None, seed=0):
    """camera.py:215-292 + rm.py:521-545 -> (points_cam (F,R,S,3), distances (F,R,S), dirs (F,R,3))."""
    _require_gpu(ijs, near, far, gt, u_coarse, u_guided)
    S = rc.num_samples_coarse + (rc.num_samples_guided if gt is not None else 0)
    pc, _, dist, dirs = torch.ops.ngm355.sample_rays(cfg_blob(rc), ijs, None, near, far, gt, u_coarse, u_guided, int(seed),
                                                     0.0, 8.0, S)
    return pc, dist, dirs


def sample_rays_world(rc: K.RenderCfg, ijs, c2ws, near=None, far=None, gt=None, u_coarse=None, u_guided=None, seed=0,
                      near_const=0.0, far_const=8.0):
    """sampler + utils.transform_points (rm.py:513-547): (points_cam, points_world, distances), each (F,R,S,.)."""
    _require_gpu(ijs, c2ws, near, far, gt, u_coarse, u_guided)
    if ijs.dim() == 2:
        ijs = ijs[None]
    S = rc.num_samples_coarse + (rc.num_samples_guided if gt is not None else 0)
    pc, pw, dist, _ = torch.ops.ngm355.sample_rays(cfg_blob(rc), ijs, c2ws, near, far, gt, u_coarse, u_guided, int(seed),
                                                   float(near_const), float(far_const), S)
    return pc, pw, dist


@_op("sample_rays_weighted")
def _sample_rays_weighted_op(rcfg: torch.Tensor, ijs: torch.Tensor, boundaries: torch.Tensor, weights: torch.Tensor,
                             u_bin: Optional[torch.Tensor], u_off: Optional[torch.Tensor], seed: int) -> List[torch.Tensor]:
    """[points_cam (F,R,S,3), distances (F,R,S), dirs (F,R,3)]"""
    rc = _render_cfg(rcfg)
    dev = ijs.device
    F, R, S, B = ijs.shape[0], ijs.shape[1], rc.num_samples_coarse, weights.shape[-1]
    pos, quat = torch.zeros(F, 3, device=dev), torch.zeros(F, 4, device=dev)
    keep = []
    rays = make_rays(rc, ijs, torch.eye(4, device=dev), None, None, None, pos, quat, u_bin, u_off, seed, keep=keep)
    bd, w = _f32c(boundaries), _f32c(weights)
    pc = torch.empty(F, R, S, 3, device=dev)
    dist = torch.empty(F, R, S, device=dev)
    dirs = torch.empty(F, R, 3, device=dev)
    K.check(K.lib().ngm_sample_rays_weighted(C.byref(rc), C.byref(rays), B, _ptr(bd), _ptr(w), _ptr(pc), _ptr(dist), _ptr(dirs),
                                             _stream()), "ngm_sample_rays_weighted")
    return [pc, dist, dirs]


@_sample_rays_weighted_op.register_fake
def _(rcfg, ijs, boundaries, weights, u_bin, u_off, seed):
    F, R, S = ijs.shape[0], ijs.shape[1], _render_cfg(rcfg).num_samples_coarse
    f = lambda *shape: torch.empty(*shape, device=ijs.device, dtype=torch.float32)
    return [f(F, R, S, 3), f(F, R, S), f(F, R, 3)]


def sample_rays_weighted(rc: K.RenderCfg, ijs, boundaries, weights, u_bin=None, u_off=None, seed=0):
    """Camera.sample_ijs_uniform(ijs, num_samples, weights=..., boundaries=...) (camera.py:277-289): weighted sampling from
    per-ray distance bins -> (points_cam (F,R,S,3), distances (F,R,S) in draw order, dirs (F,R,3)); S = rc.num_samples_coarse.
    ijs (F,R,2) int64, boundaries (F,R,B+1) sorted, weights (F,R,B); u_bin / u_off (F,R,S): the reference's two torch.rand
    draws, in its order, or both None = in-kernel Philox."""
    if (weights is None) != (boundaries is None):
        raise ValueError("Either both or none of weights and boundaries must be None.")       # camera.py:260-261
    if weights is None:
        raise ValueError("sample_rays_weighted: boundaries and weights are required (the stratified branch is sample_rays)")
    if (u_bin is None) != (u_off is None):
        raise ValueError("sample_rays_weighted: u_bin and u_off go together (both None = in-kernel Philox)")
    _require_gpu(ijs, boundaries, weights, u_bin, u_off)
    if ijs.dim() == 2:
        ijs, boundaries, weights = ijs[None], boundaries[None], weights[None]
        u_bin, u_off = (None if u_bin is None else u_bin[None]), (None if u_off is None else u_off[None])
    if boundaries.shape[:-1] != ijs.shape[:-1] or weights.shape[:-1] != ijs.shape[:-1] or boundaries.shape[-1] != weights.shape[-1] + 1:
        raise ValueError("boundaries (..., num_bins + 1) and weights (..., num_bins) must match the leading dims of ijs")
    want = (*ijs.shape[:-1], int(rc.num_samples_coarse))            # the kernel indexes the draws as (F, R, S)
    for name, u in (("u_bin", u_bin), ("u_off", u_off)):
        if u is not None and tuple(u.shape) != want:
            raise ValueError(f"sample_rays_weighted: {name} must have shape {want} (leading dims of ijs, num_samples), got {tuple(u.shape)}")
    pc, dist, dirs = torch.ops.ngm355.sample_rays_weighted(cfg_blob(rc), ijs, boundaries, weights, u_bin, u_off, int(seed))
    return pc, dist, dirs


@_op("composite_packed")
def _composite_packed_op(rcfg: torch.Tensor, field_out4: torch.Tensor, dists: torch.Tensor,
                         points_cam: torch.Tensor) -> List[torch.Tensor]:
    rc = _render_cfg(rcfg)
    N, S = dists.shape
    dev = dists.device
    rgbd, cv, dv, term = (torch.empty(N, 4, device=dev), torch.empty(N, 3, device=dev), torch.empty(N, device=dev),
                          torch.empty(N, device=dev))
    K.check(K.lib().ngm_composite_fwd_packed(C.byref(rc), N, S, _ptr(field_out4), _ptr(dists), _ptr(points_cam), _ptr(rgbd),
                                             _ptr(cv), _ptr(dv), _ptr(term), _stream()), "ngm_composite_fwd_packed")
    return [rgbd, cv, dv, term]


@_composite_packed_op.register_fake
def _(rcfg, field_out4, dists, points_cam):
    N = dists.shape[0]
    return [dists.new_empty(N, 4), dists.new_empty(N, 3), dists.new_empty(N), dists.new_empty(N)]


def composite_packed(rc: K.RenderCfg, field_out4, dists, points_cam):
    """_quadrature on the raw (N,S,4) field outputs (eval path): -> rgbd (N,4), Cvar (N,3), Dvar (N), term (N)."""
    _require_gpu(field_out4, dists, points_cam)
    S = dists.shape[-1]
    N = dists.numel() // S
    return tuple(torch.ops.ngm355.composite_packed(cfg_blob(rc), _f32c(field_out4).reshape(N, S, 4), _f32c(dists).reshape(N, S),
                                                   _f32c(points_cam).reshape(N, S, 3)))


@_op("render_eval_knn")
def _render_eval_knn_op(fcfg: torch.Tensor, rcfg: torch.Tensor, ijs: torch.Tensor, c2ws: torch.Tensor,
                        near: Optional[torch.Tensor], far: Optional[torch.Tensor], u: Optional[torch.Tensor], seed: int,
                        near_const: float, far_const: float, pos: torch.Tensor, quat: torch.Tensor, params: List[torch.Tensor],
                        num_knn: int, distance_factor: float, outside_value: float, field_index: Optional[torch.Tensor],
                        mask_radius: float, ray_block: int) -> List[torch.Tensor]:
    fc, rc = _field_cfg(fcfg), _render_cfg(rcfg)
    N, NF = ijs.shape[0], pos.shape[0]
    dev = ijs.device
    rgbd, cv, dv, term = (torch.empty(N, 4, device=dev), torch.empty(N, 3, device=dev), torch.empty(N, device=dev),
                          torch.empty(N, device=dev))
    if N == 0:
        return [rgbd, cv, dv, term]
    ps = params_struct(fc, dict(zip(K.param_names(fc), params)), field_index)
    keep = []
    rays = make_rays(rc, ijs[None], c2ws, None if near is None else near[None], None if far is None else far[None], None,
                     pos[:1], quat[:1], None if u is None else u[None], None, seed, near_const=near_const, far_const=far_const,
                     keep=keep)
    pred = K.Prediction(_ptr(rgbd), _ptr(cv), _ptr(dv), _ptr(term))
    L = K.lib()
    ray_block = max(1, min(int(ray_block), N))
    wsb = L.ngm_render_eval_knn_workspace(C.byref(rc), NF, ray_block, num_knn)
    if wsb < 0:
        raise K.NgmError("ngm_render_eval_knn_workspace: bad argument")
    ws = torch.empty(wsb, device=dev, dtype=torch.uint8)
    K.check(L.ngm_render_eval_knn(C.byref(fc), C.byref(rc), C.byref(ps), NF, _ptr(_f32c(pos)), _ptr(_f32c(quat)), C.byref(rays),
                                  num_knn, distance_factor, outside_value, mask_radius, ray_block, C.byref(pred), _ptr(ws), wsb,
                                  _stream()), "ngm_render_eval_knn")
    return [rgbd, cv, dv, term]


@_render_eval_knn_op.register_fake
def _(fcfg, rcfg, ijs, c2ws, near, far, u, seed, near_const, far_const, pos, quat, params, num_knn, distance_factor,
      outside_value, field_index, mask_radius, ray_block):
    N = ijs.shape[0]
    f = lambda *shape: torch.empty(*shape, device=ijs.device, dtype=torch.float32)
    return [f(N, 4), f(N, 3), f(N), f(N)]


def render_eval_knn(fc, rc: K.RenderCfg, params, ijs, c2ws, pos, quat, num_knn=2, distance_factor=10.0, outside_value=1.0,
                    near=None, far=None, u=None, seed=0, near_const=0.0, far_const=8.0, field_index=None, mask_radius=None,
                    ray_block=8192):
    """render_image's block loop (rm.py:402-437) in one call: eval-style samples of the rays `ijs` (N,2) -> kNN-blended
    fields -> quadrature; -> (rgbd (N,4), color_vars (N,3), depth_vars (N,), term (N,)).  The results equal
    sample_rays_world -> field_eval_knn -> composite_packed per block of `ray_block` rays (block b drawing with seed +
    b * ray_block); the samples, the blended outputs and the camera-frame points never reach memory.
    Dispatches through torch.ops.ngm355.render_eval_knn."""
    plist = [params[n] for n in K.param_names(fc)]
    _require_gpu(ijs, c2ws, pos, quat, near, far, u, *plist)
    if ijs.dtype != torch.int64 or ijs.dim() != 2:
        raise TypeError("ijs must be int64 (N,2) [row, col]")
    return tuple(torch.ops.ngm355.render_eval_knn(cfg_blob(fc), cfg_blob(rc), ijs.contiguous(), c2ws, near, far, u, int(seed),
                                                  float(near_const), float(far_const), pos, quat, plist, int(num_knn),
                                                  float(distance_factor), float(outside_value), field_index,
                                                  float(mask_radius) if mask_radius else 0.0, int(ray_block)))


# ------------------------------------------------------------------------------------------------
# fused render (NeuralGraphMap._render_ijs(use_vmap=True), rm.py:439-666) with autograd
# ------------------------------------------------------------------------------------------------
def _rays_from(rc, ijs, c2ws, near, far, gt, pos, quat, u_coarse, u_guided, seed, near_const, far_const, keep):
    return make_rays(rc, ijs, c2ws, near, far, gt, pos, quat, u_coarse, u_guided, seed, near_const=near_const,
                     far_const=far_const, keep=keep)


@_op("render_ijs")
def _render_ijs_op(fcfg: torch.Tensor, rcfg: torch.Tensor, ijs: torch.Tensor, c2ws: torch.Tensor, near: Optional[torch.Tensor],
                   far: Optional[torch.Tensor], gt: Optional[torch.Tensor], pos: torch.Tensor, quat: torch.Tensor,
                   u_coarse: Optional[torch.Tensor], u_guided: Optional[torch.Tensor], seed: int, near_const: float,
                   far_const: float, save: bool, num_samples: int, workspace_bytes: int,
                   params: List[torch.Tensor]) -> List[torch.Tensor]:
    """[rgbds (F,R,4), color_vars (F,R,3), depth_vars (F,R), term_probs (F,R), geoms (F,R,S) | empty, dists (F,R,S) | empty,
    workspace (bytes) | empty]; geoms / dists / workspace are filled when `save` (needed for a backward or for the
    free-space / TSDF vectors of the Prediction)."""
    fc, rc = _field_cfg(fcfg), _render_cfg(rcfg)
    keep = []
    rays = _rays_from(rc, ijs, c2ws, near, far, gt, pos, quat, u_coarse, u_guided, seed, near_const, far_const, keep)
    F, R = rays.F, rays.R
    dev = ijs.device
    S = rc.num_samples_coarse + (rc.num_samples_guided if gt is not None else 0)
    assert S == num_samples
    rgbds, cvars, dvars, term = (torch.empty(F, R, 4, device=dev), torch.empty(F, R, 3, device=dev),
                                 torch.empty(F, R, device=dev), torch.empty(F, R, device=dev))
    pred = K.Prediction(rgbds.data_ptr(), cvars.data_ptr(), dvars.data_ptr(), term.data_ptr())
    L = K.lib()
    ps = params_struct(fc, dict(zip(K.param_names(fc), params)))
    wsb = L.ngm_render_workspace(C.byref(fc), C.byref(rc), F, R, 1) if save else 0
    if wsb != workspace_bytes:
        raise ValueError(f"render_ijs: workspace_bytes {workspace_bytes} != ngm_render_workspace {wsb}")
    ws = torch.empty(wsb, device=dev, dtype=torch.uint8)
    K.check(L.ngm_render_fwd(C.byref(fc), C.byref(rc), C.byref(ps), C.byref(rays), None, C.byref(pred), None,
                             _ptr(ws) if save else None, wsb, _stream()), "ngm_render_fwd")
    geoms = torch.empty((F, R, S) if save else (0,), device=dev)
    dists = torch.empty((F, R, S) if save else (0,), device=dev)
    if save:
        K.check(L.ngm_render_read_samples(C.byref(fc), C.byref(rc), F, R, S, ws.data_ptr(), geoms.data_ptr(),
                                          dists.data_ptr(), _stream()), "ngm_render_read_samples")
    return [rgbds, cvars, dvars, term, geoms, dists, ws]


@_render_ijs_op.register_fake
def _(fcfg, rcfg, ijs, c2ws, near, far, gt, pos, quat, u_coarse, u_guided, seed, near_const, far_const, save, num_samples,
      workspace_bytes, params):
    F, R, S = ijs.shape[0], ijs.shape[1], num_samples
    f = lambda *shape: torch.empty(*shape, device=ijs.device, dtype=torch.float32)
    n = (F, R, S) if save else (0,)
    return [f(F, R, 4), f(F, R, 3), f(F, R), f(F, R), f(*n), f(*n),
            torch.empty(workspace_bytes, device=ijs.device, dtype=torch.uint8)]


@_op("render_ijs_bwd", mutates_args=("workspace",))
def _render_ijs_bwd_op(fcfg: torch.Tensor, rcfg: torch.Tensor, ijs: torch.Tensor, c2ws: torch.Tensor,
                       near: Optional[torch.Tensor], far: Optional[torch.Tensor], gt: Optional[torch.Tensor],
                       pos: torch.Tensor, quat: torch.Tensor, u_coarse: Optional[torch.Tensor],
                       u_guided: Optional[torch.Tensor], seed: int, near_const: float, far_const: float,
                       d_rgbds: torch.Tensor, d_term: torch.Tensor, d_geoms: Optional[torch.Tensor],
                       workspace: torch.Tensor, params: List[torch.Tensor], d_cvars: Optional[torch.Tensor] = None,
                       d_dvars: Optional[torch.Tensor] = None, rgbds: Optional[torch.Tensor] = None,
                       term: Optional[torch.Tensor] = None) -> List[torch.Tensor]:
    """gradients w.r.t. `params`; the backward kernels overwrite the saved forward values in `workspace` (single use).
    d_cvars / d_dvars: seeds on the rendered variances (losses.py's *_nll modes); then `rgbds` / `term` = the forward's outputs."""
    fc, rc = _field_cfg(fcfg), _render_cfg(rcfg)
    keep = []
    rays = _rays_from(rc, ijs, c2ws, near, far, gt, pos, quat, u_coarse, u_guided, seed, near_const, far_const, keep)
    names = K.param_names(fc)
    grads, gs = alloc_grads_separate(fc, rays.F, ijs.device)
    ps = params_struct(fc, dict(zip(names, params)))
    if d_cvars is not None or d_dvars is not None:
        pred = K.Prediction(_ptr(rgbds), None, None, _ptr(term))
        K.check(K.lib().ngm_render_bwd_seeded_vars(C.byref(fc), C.byref(rc), C.byref(ps), C.byref(rays), C.byref(pred), _ptr(d_rgbds),
                                                   _ptr(d_cvars), _ptr(d_dvars), _ptr(d_term), _ptr(d_geoms), C.byref(gs),
                                                   workspace.data_ptr(), workspace.numel(), _stream()), "ngm_render_bwd_seeded_vars")
        return [grads[n] for n in names]
    K.check(K.lib().ngm_render_bwd_seeded(C.byref(fc), C.byref(rc), C.byref(ps), C.byref(rays), _ptr(d_rgbds), _ptr(d_term),
                                          _ptr(d_geoms), C.byref(gs), workspace.data_ptr(), workspace.numel(), _stream()),
            "ngm_render_bwd_seeded")
    return [grads[n] for n in names]


@_render_ijs_bwd_op.register_fake
def _(fcfg, rcfg, ijs, c2ws, near, far, gt, pos, quat, u_coarse, u_guided, seed, near_const, far_const, d_rgbds, d_term,
      d_geoms, workspace, params, d_cvars=None, d_dvars=None, rgbds=None, term=None):
    return [torch.empty_like(p, dtype=torch.float32) for p in params]      # gradients are fp32 whatever the storage dtype


_RAY_SLOTS = ("ijs", "c2ws", "near", "far", "gt", "pos", "quat", "u_coarse", "u_guided")


def _render_ijs_setup(ctx, inputs, output):
    fcfg, rcfg, *rest = inputs
    ray_t, (seed, near_const, far_const, save, _, _), params = rest[:9], rest[9:15], rest[15]
    ctx.fcfg, ctx.rcfg, ctx.scalars, ctx.save = fcfg, rcfg, (seed, near_const, far_const), save
    ctx.present = [t is not None for t in ray_t]
    ctx.consumed = False
    # (rgbds and term_probs ride along for seeds on the variances: the means and the weight sum they are taken around)
    ctx.save_for_backward(*[t for t in ray_t if t is not None], output[0], output[3], output[6], *params)
    ctx.n_params = len(params)
    ctx.set_materialize_grads(False)       # an output the loss does not touch arrives as None (color_vars / depth_vars mostly)


def _render_ijs_backward(ctx, grads):
    if not ctx.save:
        raise RuntimeError("render_ijs: backward through a render that saved nothing (call with requires_grad parameters)")
    if ctx.consumed:
        raise RuntimeError("render_ijs: second backward through the same render -- the backward kernels overwrite the "
                           "saved per-sample values in place (single-use workspace, include/ngm_hip.h); render again")
    ctx.consumed = True
    saved = list(ctx.saved_tensors)
    params = saved[len(saved) - ctx.n_params:]
    rgbds, term, ws = saved[len(saved) - ctx.n_params - 3:len(saved) - ctx.n_params]
    it = iter(saved)
    ray_t = [next(it) if p else None for p in ctx.present]
    d_rgbds, d_cvars, d_dvars, d_term, d_geoms, _, _ = grads
    seed, near_const, far_const = ctx.scalars
    d_rgbds = torch.zeros_like(rgbds) if d_rgbds is None else d_rgbds.contiguous()
    d_term = torch.zeros_like(term) if d_term is None else d_term.contiguous()
    d_geoms = None if (d_geoms is None or not d_geoms.numel()) else d_geoms.contiguous()
    if d_cvars is not None or d_dvars is not None:      # the loss reads the rendered variances (losses.py's *_nll modes)
        g = torch.ops.ngm355.render_ijs_bwd(ctx.fcfg, ctx.rcfg, *ray_t, seed, near_const, far_const, d_rgbds, d_term, d_geoms, ws,
                                            params, None if d_cvars is None else d_cvars.contiguous(),
                                            None if d_dvars is None else d_dvars.contiguous(), rgbds, term)
    else:
        g = torch.ops.ngm355.render_ijs_bwd(ctx.fcfg, ctx.rcfg, *ray_t, seed, near_const, far_const, d_rgbds, d_term, d_geoms, ws,
                                            params)
    return (None,) * 17 + (g,)


_render_ijs_op.register_autograd(_render_ijs_backward, setup_context=_render_ijs_setup)


def render_ijs_fused(fc: K.FieldCfg, rc: K.RenderCfg, params: Dict[str, torch.Tensor], ijs, c2ws, near, far, gt, pos, quat,
                     u_coarse=None, u_guided=None, seed=0, near_const=0.0, far_const=8.0):
    """-> (rgbds, color_vars, depth_vars, term_probs, geoms | None, dists | None); differentiable w.r.t. `params` through
    rgbds / color_vars / depth_vars / term_probs / geoms.  Dispatches through torch.ops.ngm355.render_ijs."""
    plist = [params[n] for n in K.param_names(fc)]
    _require_gpu(ijs, c2ws, near, far, gt, pos, quat, u_coarse, u_guided, *plist)
    save = bool(gt is not None or (torch.is_grad_enabled() and any(t.requires_grad for t in plist)))
    if c2ws.dim() != 2 and c2ws.numel() != ijs.shape[0] * ijs.shape[1] * 16:
        c2ws = c2ws.expand(ijs.shape[0], ijs.shape[1], 4, 4)
    out = torch.ops.ngm355.render_ijs(cfg_blob(fc), cfg_blob(rc), ijs.contiguous(), _f32c(c2ws), _f32c(near), _f32c(far),
                                      _f32c(gt), _f32c(pos), _f32c(quat), _f32c(u_coarse), _f32c(u_guided), int(seed),
                                      float(near_const), float(far_const), save,
                                      rc.num_samples_coarse + (rc.num_samples_guided if gt is not None else 0),
                                      int(K.lib().ngm_render_workspace(C.byref(fc), C.byref(rc), ijs.shape[0], ijs.shape[1], 1)) if save else 0,
                                      plist)
    rgbds, cvars, dvars, term, geoms, dists, _ = out
    # color_vars / depth_vars stay in the graph: a loss that reads them (losses.py's *_nll modes) seeds them, and the backward
    # then runs ngm_render_bwd_seeded_vars; a loss that does not leaves their gradients None (set_materialize_grads(False))
    return (rgbds, cvars, dvars, term, geoms if save else None, dists.detach() if save else None)


# ------------------------------------------------------------------------------------------------
# K4: quadrature with autograd (rm.py:709-799)
# ------------------------------------------------------------------------------------------------
def _s_eff(rc, S):
    return S - 1 if rc.geometry_mode in (K.GEO["density"], K.GEO["neus"]) else S


@_op("quadrature")
def _quadrature_op(rcfg: torch.Tensor, colors: torch.Tensor, geoms: torch.Tensor, dists: torch.Tensor, depths: torch.Tensor,
                   isds: Optional[torch.Tensor], num_weights: int) -> List[torch.Tensor]:
    rc = _render_cfg(rcfg)
    N, S = geoms.shape
    assert num_weights == _s_eff(rc, S)
    dev = geoms.device
    Cc, D, Cv, Dv, T = (torch.empty(N, 3, device=dev), torch.empty(N, device=dev), torch.empty(N, 3, device=dev),
                        torch.empty(N, device=dev), torch.empty(N, device=dev))
    W = torch.empty(N, _s_eff(rc, S), device=dev)
    K.check(K.lib().ngm_composite_fwd(C.byref(rc), N, S, _ptr(colors), _ptr(geoms), _ptr(dists), _ptr(depths),
                                      _ptr(isds), _ptr(Cc), _ptr(D), _ptr(Cv), _ptr(Dv), _ptr(T), _ptr(W),
                                      _stream()), "ngm_composite_fwd")
    return [Cc, D, Cv, Dv, T, W]


@_quadrature_op.register_fake
def _(rcfg, colors, geoms, dists, depths, isds, num_weights):
    N = geoms.shape[0]          # shapes from tensor sizes and ints only: the cfg bytes are not readable under fake mode
    return [geoms.new_empty(N, 3), geoms.new_empty(N), geoms.new_empty(N, 3), geoms.new_empty(N), geoms.new_empty(N),
            geoms.new_empty(N, num_weights)]


@_op("quadrature_bwd")
def _quadrature_bwd_op(rcfg: torch.Tensor, colors: torch.Tensor, geoms: torch.Tensor, dists: torch.Tensor,
                       depths: torch.Tensor, isds: Optional[torch.Tensor], dC: torch.Tensor, dD: torch.Tensor,
                       dT: torch.Tensor, want_isd: bool) -> List[torch.Tensor]:
    rc = _render_cfg(rcfg)
    N, S = geoms.shape
    d_colors, d_geoms = torch.empty_like(colors), torch.empty_like(geoms)
    d_isd = torch.empty(N if want_isd else 0, device=geoms.device)
    K.check(K.lib().ngm_composite_bwd(C.byref(rc), N, S, _ptr(colors), _ptr(geoms), _ptr(dists), _ptr(depths),
                                      _ptr(isds), _ptr(dC), _ptr(dD), _ptr(dT), _ptr(d_colors), _ptr(d_geoms),
                                      _ptr(d_isd) if want_isd else None, _stream()), "ngm_composite_bwd")
    return [d_colors, d_geoms, d_isd]


@_quadrature_bwd_op.register_fake
def _(rcfg, colors, geoms, dists, depths, isds, dC, dD, dT, want_isd):
    return [torch.empty_like(colors), torch.empty_like(geoms), geoms.new_empty(geoms.shape[0] if want_isd else 0)]


def _quadrature_setup(ctx, inputs, output):
    rcfg, colors, geoms, dists, depths, isds, _ = inputs
    ctx.rcfg, ctx.has_isd = rcfg, isds is not None
    ctx.save_for_backward(colors, geoms, dists, depths, *([isds] if isds is not None else []))


def _quadrature_backward(ctx, grads):
    dC, dD, _, _, dT, _ = grads
    colors, geoms, dists, depths, *rest = ctx.saved_tensors
    isds = rest[0] if ctx.has_isd else None
    N = geoms.shape[0]
    z = lambda g, *shape: torch.zeros(*shape, device=geoms.device) if g is None else g.contiguous()
    want = ctx.has_isd and ctx.needs_input_grad[5] and _render_cfg(ctx.rcfg).geometry_mode == K.GEO["neus"]
    d_colors, d_geoms, d_isd = torch.ops.ngm355.quadrature_bwd(ctx.rcfg, colors, geoms, dists, depths, isds, z(dC, N, 3),
                                                               z(dD, N), z(dT, N), bool(want))
    return None, d_colors, d_geoms, None, None, (d_isd if want else None), None


_quadrature_op.register_autograd(_quadrature_backward, setup_context=_quadrature_setup)


def quadrature(rc: K.RenderCfg, sample_colors, sample_geometries, sample_distances, sample_depths, neus_isds=None):
    """Returns (ray_colors, ray_depths, ray_color_vars, ray_depth_vars, ray_term_probs, sample_weights).
    NeuralGraphMap._quadrature (rm.py:709-799) with autograd; dispatches through torch.ops.ngm355.quadrature."""
    _require_gpu(sample_colors, sample_geometries, sample_distances, sample_depths, neus_isds)
    lead = sample_geometries.shape[:-1]
    S = sample_geometries.shape[-1]
    N = sample_geometries.numel() // S
    isd_flat = None
    if neus_isds is not None:                        # one value per ray (autograd undoes the broadcast)
        isd_flat = neus_isds.expand(*lead, 1).reshape(N).contiguous().float()
    Cc, D, Cv, Dv, T, W = torch.ops.ngm355.quadrature(
        cfg_blob(rc), _f32c(sample_colors).reshape(N, S, 3), _f32c(sample_geometries).reshape(N, S),
        _f32c(sample_distances).reshape(N, S), _f32c(sample_depths).reshape(N, S), isd_flat, _s_eff(rc, S))
    return (Cc.view(*lead, 3), D.view(*lead), Cv.detach().view(*lead, 3), Dv.detach().view(*lead), T.view(*lead),
            W.detach().view(*lead, W.shape[-1]))


# ------------------------------------------------------------------------------------------------
# sparse Adam
# ------------------------------------------------------------------------------------------------
@_op("adam_sparse_", mutates_args=("param", "exp_avg", "exp_avg_sq"))
def _adam_sparse_op(param: torch.Tensor, exp_avg: torch.Tensor, exp_avg_sq: torch.Tensor, grad: torch.Tensor,
                    field_index: torch.Tensor, step: int, lr: float, beta1: float, beta2: float, eps: float,
                    weight_decay: float) -> None:
    F = grad.shape[0]
    numel = grad[0].numel()
    K.check(K.lib().ngm_adam_sparse(_ptr(param), _ptr(exp_avg), _ptr(exp_avg_sq), param.stride(0), _ptr(grad),
                                    grad.stride(0), _ptr(field_index), F, numel, int(step), lr, beta1, beta2, eps,
                                    weight_decay, _stream()), "ngm_adam_sparse")


def adam_sparse_(param, exp_avg, exp_avg_sq, grad, field_index, step, lr=1e-3, betas=(0.9, 0.999), eps=1e-15,
                 weight_decay=1e-5):
    """In-place Adam on rows `field_index` of stacked tensors (N, ...); grad is (F, ...).
    torch.ops.ngm355.adam_sparse_ (declared as mutating param / exp_avg / exp_avg_sq)."""
    _require_gpu(param, exp_avg, exp_avg_sq, grad, field_index)
    torch.ops.ngm355.adam_sparse_(param, exp_avg, exp_avg_sq, grad, field_index, int(step), float(lr), float(betas[0]),
                                  float(betas[1]), float(eps), float(weight_decay))


def _adam_tensor(p, st, g, lp=None):
    t = K.AdamTensor(p.data_ptr(), st["exp_avg"].data_ptr(), st["exp_avg_sq"].data_ptr(), g.data_ptr(), p.stride(0), g.stride(0),
                     g[0].numel())
    if lp is not None and lp.dtype != torch.float32:      # reduced-precision copy the kernels read: refreshed by the update
        if lp.shape != p.shape or lp.stride(0) != p.stride(0):
            raise ValueError("reduced-precision copy must mirror the master tensor's layout")
        t.param_lp, t.lp_dtype = lp.data_ptr(), _TORCH_DT[lp.dtype]
    return t


def adam_tensor_arrays(fc, params, state, grads, lp=None):
    """(mlp AdamTensor array in gradient-segment order, lattice AdamTensor or None) for ngm_render_bwd_adam.
    `params` = fp32 master weights; `lp` (optional dict) = their reduced-precision copies, kept in sync by the update."""
    names = [n for n in K.param_names(fc) if n not in K.NO_GRAD_PARAMS]
    mlp = [n for n in names if n not in ("_encoding.lattice_values", "_encoding.plane_coef")]   # these two: own reduction kernels
    arr = (K.AdamTensor * len(mlp))()
    for i, n in enumerate(mlp):
        arr[i] = _adam_tensor(params[n], state[n], grads[n], None if lp is None else lp[n])
    lat = None
    if "_encoding.lattice_values" in names:
        n = "_encoding.lattice_values"
        lat = (K.AdamTensor * 1)(_adam_tensor(params[n], state[n], grads[n], None if lp is None else lp[n]))
    return arr, len(mlp), lat


def adam_sparse_multi_(fc, params, state, grads, field_index, step, step_dev=None, lr=1e-3, betas=(0.9, 0.999),
                       eps=1e-15, weight_decay=1e-5, advance=False, philox_offset_dev=None, lp=None):
    """One launch for every parameter tensor of the field set (rows `field_index` updated in place)."""
    names = [n for n in K.param_names(fc) if n not in K.NO_GRAD_PARAMS]
    arr = (K.AdamTensor * len(names))()
    for i, n in enumerate(names):
        arr[i] = _adam_tensor(params[n], state[n], grads[n], None if lp is None else lp[n])
    K.check(K.lib().ngm_adam_sparse_multi(arr, len(names), _ptr(field_index), grads[names[0]].shape[0], int(step),
                                          _ptr(step_dev), lr, betas[0], betas[1], eps, weight_decay,
                                          int(bool(advance and step_dev is not None)),
                                          _ptr(philox_offset_dev) if advance else None, _stream()),
            "ngm_adam_sparse_multi")


# ------------------------------------------------------------------------------------------------
# training-target sampler, device part (rm.py:1321-1459)
# ------------------------------------------------------------------------------------------------
def keyframes_struct(c2ws, rgbd_store, frame_to_store, fx, fy, cx, cy) -> K.Keyframes:
    """c2ws (Nc,4,4), rgbd_store (N,H,W,4), frame_to_store (Nc,) int64; intrinsics at pixel centre 0."""
    _require_gpu(c2ws, rgbd_store, frame_to_store)
    assert c2ws.is_contiguous() and rgbd_store.is_contiguous() and frame_to_store.dtype == torch.int64
    kf = K.Keyframes()
    kf.num_frames, kf.height, kf.width = c2ws.shape[0], rgbd_store.shape[1], rgbd_store.shape[2]
    kf.c2ws = C.cast(c2ws.data_ptr(), K.f32p)
    kf.rgbd = C.cast(rgbd_store.data_ptr(), K.f32p)
    kf.frame_to_store = frame_to_store.data_ptr()
    kf.fx, kf.fy, kf.cx, kf.cy = float(fx), float(fy), float(cx), float(cy)
    return kf


def target_visibility(kf: K.Keyframes, field_pos, offsets, radius):
    """(kf_mask (F,Nc) bool, bbox (F,Nc,4) [min_x, min_y, max_x, max_y] clamped to the image)."""
    field_pos, offsets = _f32c(field_pos), _f32c(offsets)
    F, Nc = field_pos.shape[0], kf.num_frames
    mask = torch.empty(F, Nc, dtype=torch.uint8, device=field_pos.device)
    bbox = torch.empty(F, Nc, 4, device=field_pos.device)
    K.check(K.lib().ngm_target_visibility(C.byref(kf), F, _ptr(field_pos), offsets.shape[0], _ptr(offsets), float(radius),
                                          _ptr(mask), _ptr(bbox), _stream()), "ngm_target_visibility")
    return mask.bool(), bbox


def target_rays(kf: K.Keyframes, field_pos, radius, bbox, frame_cids, u_xy):
    """Per-ray targets (rm.py:1394-1459) as a dict keyed like the reference's Target record."""
    field_pos, bbox, u_xy = _f32c(field_pos), _f32c(bbox), _f32c(u_xy)
    frame_cids = frame_cids.contiguous()
    F, R = frame_cids.shape
    dev = field_pos.device
    o = dict(ijs=torch.empty(F, R, 2, dtype=torch.int64, device=dev), c2ws=torch.empty(F, R, 4, 4, device=dev),
             near=torch.empty(F, R, device=dev), far=torch.empty(F, R, device=dev), gt=torch.empty(F, R, device=dev),
             rgbds=torch.empty(F, R, 4, device=dev), rgb_mask=torch.empty(F, R, dtype=torch.uint8, device=dev),
             depth_mask=torch.empty(F, R, dtype=torch.uint8, device=dev), term_probs=torch.empty(F, R, device=dev),
             term_mask=torch.empty(F, R, dtype=torch.uint8, device=dev))
    out = K.TargetOut()
    out.ijs = o["ijs"].data_ptr()
    for k in ("c2ws", "near", "far", "gt", "rgbds", "term_probs"):
        setattr(out, k, C.cast(o[k].data_ptr(), K.f32p))
    for k in ("rgb_mask", "depth_mask", "term_mask"):
        setattr(out, k, o[k].data_ptr())
    K.check(K.lib().ngm_target_rays(C.byref(kf), F, R, _ptr(field_pos), float(radius), _ptr(bbox), _ptr(frame_cids),
                                    _ptr(u_xy), C.byref(out), _stream()), "ngm_target_rays")
    for k in ("rgb_mask", "depth_mask", "term_mask"):
        o[k] = o[k].bool()
    return o


def target_sv_intersect(pos_c, points, radius):
    """(F, N) bool: the segment camera origin -> point n passes through the sphere of field f (geometry.py:67-105)."""
    pos_c, points = _f32c(pos_c), _f32c(points)
    F, N = pos_c.shape[0], points.shape[0]
    hit = torch.empty(F, N, dtype=torch.uint8, device=points.device)
    K.check(K.lib().ngm_target_sv_intersect(F, N, _ptr(pos_c), _ptr(points), float(radius), _ptr(hit), _stream()),
            "ngm_target_sv_intersect")
    return hit.bool()


def target_sv_rays(pos_c, radius, pts_ijs, segments, image, fx, fy, cx, cy):
    """Per-ray targets of the single-view sampler (rm.py:1536-1561), keyed like the reference's Target record."""
    pos_c, image = _f32c(pos_c), _f32c(image)
    _require_gpu(pos_c, image, pts_ijs, segments)
    if pts_ijs.dtype != torch.int64 or segments.dtype != torch.int64:
        raise TypeError(f"target_sv_rays: pts_ijs / segments must be int64 (got {pts_ijs.dtype}, {segments.dtype})")
    pts_ijs, segments = pts_ijs.contiguous(), segments.contiguous()
    F, R = segments.shape
    H, W = image.shape[0], image.shape[1]
    if segments.numel() and (int(segments.min()) < 0 or int(segments.max()) >= pts_ijs.shape[0]):
        raise IndexError("target_sv_rays: segment index outside the point list")
    if pts_ijs.numel() and (int(pts_ijs[:, 0].min()) < 0 or int(pts_ijs[:, 0].max()) >= H or int(pts_ijs[:, 1].min()) < 0
                            or int(pts_ijs[:, 1].max()) >= W):
        raise IndexError("target_sv_rays: pixel index outside the frame")
    dev = pos_c.device
    o = dict(ijs=torch.empty(F, R, 2, dtype=torch.int64, device=dev), near=torch.empty(F, R, device=dev),
             far=torch.empty(F, R, device=dev), gt=torch.empty(F, R, device=dev), rgbds=torch.empty(F, R, 4, device=dev),
             rgb_mask=torch.empty(F, R, dtype=torch.uint8, device=dev), depth_mask=torch.empty(F, R, dtype=torch.uint8, device=dev),
             term_probs=torch.empty(F, R, device=dev), term_mask=torch.empty(F, R, dtype=torch.uint8, device=dev))
    out = K.TargetOut()
    out.ijs = o["ijs"].data_ptr()
    for k in ("near", "far", "gt", "rgbds", "term_probs"):
        setattr(out, k, C.cast(o[k].data_ptr(), K.f32p))
    for k in ("rgb_mask", "depth_mask", "term_mask"):
        setattr(out, k, o[k].data_ptr())
    K.check(K.lib().ngm_target_sv_rays(F, R, _ptr(pos_c), float(radius), _ptr(pts_ijs), _ptr(segments), _ptr(image), H, W,
                                       float(fx), float(fy), float(cx), float(cy), C.byref(out), _stream()), "ngm_target_sv_rays")
    for k in ("rgb_mask", "depth_mask", "term_mask"):
        o[k] = o[k].bool()
    return o
