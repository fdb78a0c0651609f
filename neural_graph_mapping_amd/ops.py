"""PyTorch-ROCm custom ops over the C ABI (include/ngm_hip.h).

torch is plumbing here (device memory, streams, autograd glue); every op below is a hand-written
gfx950 kernel reached through ``_capi``.  There is NO CPU / eager fallback: tensors must live on a
ROCm device ("cuda" under PyTorch-ROCm) and the HIP library must be built, otherwise the call raises.
"""
import ctypes as C
from typing import Dict, Optional

import torch

from . import _capi as K


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
def params_struct(fc: K.FieldCfg, params: Dict[str, torch.Tensor], field_index: Optional[torch.Tensor] = None):
    ptrs, strides = {}, {}
    for n, shp in K.param_shapes(fc).items():
        if n not in params:
            raise KeyError(f"missing parameter tensor '{n}'")
        t = params[n]
        _require_gpu(t)
        if t.dtype != torch.float32 or tuple(t.shape[1:]) != tuple(shp):
            raise ValueError(f"parameter '{n}' must be float32 (N,{shp}), got {t.dtype} {tuple(t.shape)}")
        inner = t[0] if t.shape[0] > 0 else t
        if t.shape[0] > 0 and not inner.is_contiguous():
            raise ValueError(f"parameter '{n}' rows must be contiguous")
        ptrs[n] = t.data_ptr()
        strides[n] = t.stride(0) if t.shape[0] > 1 else int(torch.tensor(shp).prod())
    fi = None
    if field_index is not None:
        _require_gpu(field_index)
        if field_index.dtype != torch.int64:
            raise TypeError("field_index must be int64")
        fi = field_index.contiguous().data_ptr()
    return K.params_struct(fc, ptrs, strides, fi)


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


# ------------------------------------------------------------------------------------------------
# K2+K3: NeuralFieldSet.forward(use_vmap=True) with autograd
# ------------------------------------------------------------------------------------------------
class _FieldEval(torch.autograd.Function):
    @staticmethod
    def forward(ctx, fc, names, points, pos, quat, *param_tensors):
        params = dict(zip(names, param_tensors))
        _require_gpu(points, pos, quat, *param_tensors)
        points = _f32c(points, "points")
        F, P, _ = points.shape
        out = torch.empty(F, P, 4, device=points.device, dtype=torch.float32)
        ps = params_struct(fc, params)
        K.check(K.lib().ngm_field_eval_fwd(C.byref(fc), C.byref(ps), F, P, _ptr(points), _ptr(_f32c(pos)),
                                           _ptr(_f32c(quat)), _ptr(out), _stream()), "ngm_field_eval_fwd")
        ctx.fc, ctx.names = fc, names
        ctx.save_for_backward(points, pos, quat, *param_tensors)
        return out

    @staticmethod
    def backward(ctx, d_out):
        points, pos, quat, *param_tensors = ctx.saved_tensors
        fc, names = ctx.fc, ctx.names
        params = dict(zip(names, param_tensors))
        F, P, _ = points.shape
        d_out = _f32c(d_out, "d_out")
        grads, gs, _ = alloc_grads(fc, F, points.device)
        ps = params_struct(fc, params)
        L = K.lib()
        wsb = L.ngm_field_eval_bwd_workspace(C.byref(fc), F, P)
        ws = torch.empty(wsb, device=points.device, dtype=torch.uint8)
        K.check(L.ngm_field_eval_bwd(C.byref(fc), C.byref(ps), F, P, _ptr(points), _ptr(_f32c(pos)), _ptr(_f32c(quat)),
                                     _ptr(d_out), C.byref(gs), _ptr(ws), wsb, _stream()), "ngm_field_eval_bwd")
        return (None, None, None, None, None) + tuple(grads[n] for n in names)


def field_eval(fc: K.FieldCfg, params: Dict[str, torch.Tensor], points, pos=None, quat=None):
    """(F,P,3) points -> (F,P,4); differentiable w.r.t. the parameters (not the points/poses)."""
    names = tuple(K.param_names(fc))
    return _FieldEval.apply(fc, names, points, pos, quat, *[params[n] for n in names])


def field_eval_knn(fc, params, points, pos, quat, num_knn=2, distance_factor=10.0, outside_value=1.0,
                   field_index=None):
    """models.py:347-405: kNN-blended evaluation of world points (P,3) over all fields -> (P,4)."""
    _require_gpu(points, pos, quat)
    points = _f32c(points.reshape(-1, 3))
    P, NF = points.shape[0], pos.shape[0]
    out = torch.empty(P, 4, device=points.device, dtype=torch.float32)
    if P == 0:
        return out
    ps = params_struct(fc, params, field_index)
    L = K.lib()
    wsb = L.ngm_field_eval_knn_workspace(NF, P, num_knn)
    ws = torch.empty(wsb, device=points.device, dtype=torch.uint8)
    K.check(L.ngm_field_eval_knn(C.byref(fc), C.byref(ps), NF, P, _ptr(points), _ptr(_f32c(pos)), _ptr(_f32c(quat)),
                                 num_knn, distance_factor, outside_value, _ptr(out), _ptr(ws), wsb, _stream()),
            "ngm_field_eval_knn")
    return out


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


def sample_rays(rc: K.RenderCfg, ijs, near, far, gt=None, u_coarse=None, u_guided=None, seed=0):
    """camera.py:215-292 + rm.py:521-545 -> (points_cam (F,R,S,3), distances (F,R,S), dirs (F,R,3))."""
    keep = []
    dev = ijs.device
    F, R = ijs.shape[0], ijs.shape[1]
    eye = torch.eye(4, device=dev)
    pos = torch.zeros(F, 3, device=dev)
    quat = torch.zeros(F, 4, device=dev)
    rays = make_rays(rc, ijs, eye, near, far, gt, pos, quat, u_coarse, u_guided, seed, keep=keep)
    S = rc.num_samples_coarse + (rc.num_samples_guided if gt is not None else 0)
    pts = torch.empty(F, R, S, 3, device=dev)
    dist = torch.empty(F, R, S, device=dev)
    dirs = torch.empty(F, R, 3, device=dev)
    K.check(K.lib().ngm_sample_rays(C.byref(rc), C.byref(rays), _ptr(pts), _ptr(dist), _ptr(dirs), _stream()),
            "ngm_sample_rays")
    return pts, dist, dirs


def sample_rays_world(rc: K.RenderCfg, ijs, c2ws, near=None, far=None, gt=None, u_coarse=None, u_guided=None, seed=0,
                      near_const=0.0, far_const=8.0):
    """sampler + utils.transform_points (rm.py:513-547): (points_cam, points_world, distances), each (F,R,S,·)."""
    keep = []
    dev = ijs.device
    if ijs.dim() == 2:
        ijs = ijs[None]
    F, R = ijs.shape[0], ijs.shape[1]
    pos = torch.zeros(F, 3, device=dev)
    quat = torch.zeros(F, 4, device=dev)
    rays = make_rays(rc, ijs, c2ws, near, far, gt, pos, quat, u_coarse, u_guided, seed, near_const=near_const,
                     far_const=far_const, keep=keep)
    S = rc.num_samples_coarse + (rc.num_samples_guided if gt is not None else 0)
    pc = torch.empty(F, R, S, 3, device=dev)
    pw = torch.empty(F, R, S, 3, device=dev)
    dist = torch.empty(F, R, S, device=dev)
    K.check(K.lib().ngm_sample_rays_world(C.byref(rc), C.byref(rays), _ptr(pc), _ptr(pw), _ptr(dist), None, _stream()),
            "ngm_sample_rays_world")
    return pc, pw, dist


def composite_packed(rc: K.RenderCfg, field_out4, dists, points_cam):
    """_quadrature on the raw (N,S,4) field outputs (eval path): -> rgbd (N,4), Cvar (N,3), Dvar (N), term (N)."""
    _require_gpu(field_out4, dists, points_cam)
    S = dists.shape[-1]
    N = dists.numel() // S
    dev = dists.device
    rgbd, cv, dv, term = (torch.empty(N, 4, device=dev), torch.empty(N, 3, device=dev), torch.empty(N, device=dev),
                          torch.empty(N, device=dev))
    K.check(K.lib().ngm_composite_fwd_packed(C.byref(rc), N, S, _ptr(_f32c(field_out4)), _ptr(_f32c(dists)),
                                             _ptr(_f32c(points_cam)), _ptr(rgbd), _ptr(cv), _ptr(dv), _ptr(term),
                                             _stream()), "ngm_composite_fwd_packed")
    return rgbd, cv, dv, term


# ------------------------------------------------------------------------------------------------
# K4: quadrature with autograd (rm.py:709-799)
# ------------------------------------------------------------------------------------------------
class _Quadrature(torch.autograd.Function):
    @staticmethod
    def forward(ctx, rc, colors, geoms, dists, depths, isds):
        _require_gpu(colors, geoms, dists, depths, isds)
        lead = geoms.shape[:-1]
        S = geoms.shape[-1]
        N = geoms.numel() // S
        colors, geoms, dists, depths = (_f32c(colors), _f32c(geoms), _f32c(dists), _f32c(depths))
        isd_flat = None
        if isds is not None:
            isd_flat = isds.expand(*lead, 1).reshape(N).contiguous().float()
        dev = geoms.device
        S_eff = S - 1 if rc.geometry_mode in (K.GEO["density"], K.GEO["neus"]) else S
        Cc, D, Cv, Dv, T = (torch.empty(N, 3, device=dev), torch.empty(N, device=dev), torch.empty(N, 3, device=dev),
                            torch.empty(N, device=dev), torch.empty(N, device=dev))
        W = torch.empty(N, S_eff, device=dev)
        K.check(K.lib().ngm_composite_fwd(C.byref(rc), N, S, _ptr(colors), _ptr(geoms), _ptr(dists), _ptr(depths),
                                          _ptr(isd_flat), _ptr(Cc), _ptr(D), _ptr(Cv), _ptr(Dv), _ptr(T), _ptr(W),
                                          _stream()), "ngm_composite_fwd")
        ctx.rc = rc
        ctx.isd_shape = None if isds is None else tuple(isds.shape)
        ctx.lead = tuple(lead)
        ctx.save_for_backward(colors, geoms, dists, depths, isd_flat)
        ctx.mark_non_differentiable(Cv, Dv, W)
        return (Cc.view(*lead, 3), D.view(*lead), Cv.view(*lead, 3), Dv.view(*lead), T.view(*lead),
                W.view(*lead, S_eff))

    @staticmethod
    def backward(ctx, dC, dD, dCv, dDv, dT, dW):
        colors, geoms, dists, depths, isd_flat = ctx.saved_tensors
        S = geoms.shape[-1]
        N = geoms.numel() // S
        d_colors = torch.empty_like(colors)
        d_geoms = torch.empty_like(geoms)
        want_isd = isd_flat is not None and ctx.needs_input_grad[5] and ctx.rc.geometry_mode == K.GEO["neus"]
        d_isd = torch.empty(N, device=geoms.device) if want_isd else None
        K.check(K.lib().ngm_composite_bwd(C.byref(ctx.rc), N, S, _ptr(colors), _ptr(geoms), _ptr(dists), _ptr(depths),
                                          _ptr(isd_flat), _ptr(_f32c(dC)), _ptr(_f32c(dD)), _ptr(_f32c(dT)),
                                          _ptr(d_colors), _ptr(d_geoms), _ptr(d_isd), _stream()), "ngm_composite_bwd")
        g_isd = None
        if want_isd:   # undo the broadcast of neus_isds to one value per ray
            g_isd = d_isd.view(*ctx.lead, 1).sum_to_size(ctx.isd_shape)
        return None, d_colors, d_geoms, None, None, g_isd


def quadrature(rc: K.RenderCfg, sample_colors, sample_geometries, sample_distances, sample_depths, neus_isds=None):
    """Returns (ray_colors, ray_depths, ray_color_vars, ray_depth_vars, ray_term_probs, sample_weights)."""
    return _Quadrature.apply(rc, sample_colors, sample_geometries, sample_distances, sample_depths, neus_isds)


# ------------------------------------------------------------------------------------------------
# sparse Adam
# ------------------------------------------------------------------------------------------------
def adam_sparse_(param, exp_avg, exp_avg_sq, grad, field_index, step, lr=1e-3, betas=(0.9, 0.999), eps=1e-15,
                 weight_decay=1e-5):
    """In-place Adam on rows `field_index` of stacked tensors (N, ...); grad is (F, ...)."""
    _require_gpu(param, exp_avg, exp_avg_sq, grad, field_index)
    F = grad.shape[0]
    numel = grad[0].numel()
    K.check(K.lib().ngm_adam_sparse(_ptr(param), _ptr(exp_avg), _ptr(exp_avg_sq), param.stride(0), _ptr(grad),
                                    grad.stride(0), _ptr(field_index), F, numel, int(step), lr, betas[0], betas[1], eps,
                                    weight_decay, _stream()), "ngm_adam_sparse")


def adam_tensor_arrays(fc, params, state, grads):
    """(mlp AdamTensor array in gradient-segment order, lattice AdamTensor or None) for ngm_render_bwd_adam."""
    names = [n for n in K.param_names(fc) if n not in K.NO_GRAD_PARAMS]
    mlp = [n for n in names if n != "_encoding.lattice_values"]
    arr = (K.AdamTensor * len(mlp))()
    for i, n in enumerate(mlp):
        p, g = params[n], grads[n]
        arr[i] = K.AdamTensor(p.data_ptr(), state[n]["exp_avg"].data_ptr(), state[n]["exp_avg_sq"].data_ptr(),
                              g.data_ptr(), p.stride(0), g.stride(0), g[0].numel())
    lat = None
    if "_encoding.lattice_values" in names:
        n = "_encoding.lattice_values"
        p, g = params[n], grads[n]
        lat = (K.AdamTensor * 1)(K.AdamTensor(p.data_ptr(), state[n]["exp_avg"].data_ptr(), state[n]["exp_avg_sq"].data_ptr(),
                                             g.data_ptr(), p.stride(0), g.stride(0), g[0].numel()))
    return arr, len(mlp), lat


def adam_sparse_multi_(fc, params, state, grads, field_index, step, step_dev=None, lr=1e-3, betas=(0.9, 0.999),
                       eps=1e-15, weight_decay=1e-5, advance=False, philox_offset_dev=None):
    """One launch for every parameter tensor of the field set (rows `field_index` updated in place)."""
    names = [n for n in K.param_names(fc) if n not in K.NO_GRAD_PARAMS]
    arr = (K.AdamTensor * len(names))()
    for i, n in enumerate(names):
        p, g = params[n], grads[n]
        arr[i] = K.AdamTensor(p.data_ptr(), state[n]["exp_avg"].data_ptr(), state[n]["exp_avg_sq"].data_ptr(),
                              g.data_ptr(), p.stride(0), g.stride(0), g[0].numel())
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
