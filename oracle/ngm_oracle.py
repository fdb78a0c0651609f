"""CPU oracle for the per-field NeRF render/train hot path of neural_graph_mapping.

*** TEST INFRASTRUCTURE -- NOT PRODUCT CODE ***
Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import this module.  The product path (``neural_graph_mapping_amd``)
never routes through it; it fails loudly when the HIP library is missing.

This is a from-scratch restatement (plain PyTorch on CPU, fp32 by default, fp64
on request) of the reference algorithm.  Every function cites the reference
file:line it follows (paths relative to /root/reference/src/neural_graph_mapping,
``rm.py`` = ``run_mapping.py``).  Randomness is an *input* (``u_coarse`` /
``u_guided`` are the ``torch.rand`` draws of camera.py:274) so that results can
be compared sample-for-sample.

Parity pin: ``tests/test_oracle_golden.py`` checks every function below against
fixtures generated from the real reference (``tests/golden/make_golden.py``);
``tests/test_oracle_vs_reference.py`` additionally compares against the live
reference whenever /root/reference is present.  Hash (permutohedral) encoding
is **parity unpinned**: its arithmetic lives in an un-vendored CUDA dependency
(permutohedral_encoding @ bf445adb, pyproject.toml:21).
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Dict, Optional, Tuple

import torch

# ----------------------------------------------------------------------------------------
# specs
# ----------------------------------------------------------------------------------------


@dataclass
class CameraSpec:
    """Pinhole intrinsics; principal point given for pixel_center 0 (camera.py:22-80,98-116).

    The reference stores cx at pixel centre 0.5 (camera.py:69-70) and converts back with
    get_pinhole_camera_parameters(0.0) inside ijs_to_directions (camera.py:188), i.e. the
    effective principal point is ``cx_cfg - pixel_center_cfg``.
    """
    width: int
    height: int
    fx: float
    fy: float
    cx: float
    cy: float

    @staticmethod
    def from_config(width, height, fx, fy, cx, cy, pixel_center=0.0, **_):
        return CameraSpec(width, height, fx, fy, cx - pixel_center, cy - pixel_center)


@dataclass
class FieldSpec:
    """Architecture of one NeuralField (models.py:69-128)."""
    encoding: str = "fourier"          # "fourier" | "nerf" | "none" | "permuto" | "triplane"
    dim_enc: int = 64                  # encoding width D
    raw_coords: bool = True            # Fourier: cat(x, sin(Wx)) (positional_encodings.py:208-212)
    num_octaves: int = 8               # NeRF octaves (positional_encodings.py:230)
    start_octave: int = 0
    # permutohedral hash encoding (positional_encodings.py:19-66; PARITY UNPINNED, see encode_permuto)
    nr_levels: int = 16
    nr_feat_per_level: int = 2
    log2_hashmap_size: int = 12
    coarsest_scale: float = 1.0
    finest_scale: float = 1e-4
    # triplane encoding (positional_encodings.py:69-161)
    resolution: int = 32
    num_components: int = 64
    tri_mode: str = "sum"              # "sum" | "product" | "concat"
    num_layers: int = 2                # hidden layers L
    dim_hidden: Optional[int] = None   # H; None -> D (models.py:99-100)
    dim_out: int = 4
    skip_mode: str = "no"              # "no" | "add" | "concat" | "rezero" (models.py:159-180)

    def __post_init__(self):
        if self.encoding == "nerf":
            self.dim_enc = 3 * self.num_octaves * 2
        if self.encoding == "permuto":
            self.dim_enc = self.nr_levels * self.nr_feat_per_level
        if self.encoding == "triplane":
            self.dim_enc = self.num_components * (3 if self.tri_mode == "concat" else 1)   # positional_encodings.py:119-126
        if self.dim_hidden is None:
            self.dim_hidden = self.dim_enc

    def layer_dims(self):
        mlp_in = self.dim_hidden + (self.dim_enc if self.skip_mode == "concat" else 0)   # models.py:105-110
        dims_in = [self.dim_enc] + [mlp_in] * self.num_layers
        dims_out = [self.dim_hidden] * self.num_layers + [self.dim_out]
        return list(zip(dims_in, dims_out))

    def param_shapes(self) -> Dict[str, Tuple[int, ...]]:
        """Names as in the reference's stacked state dict."""
        shapes = {}
        if self.encoding == "fourier":
            n = self.dim_enc - 3 if self.raw_coords else self.dim_enc
            shapes["_encoding._linear.weight"] = (n, 3)
        if self.encoding == "permuto":
            shapes["_encoding.lattice_values"] = (self.nr_levels, 2 ** self.log2_hashmap_size, self.nr_feat_per_level)
            shapes["_encoding.random_shift_per_level"] = (self.nr_levels, 3)
        if self.encoding == "triplane":
            shapes["_encoding.plane_coef"] = (3, self.num_components, self.resolution, self.resolution)
        if self.skip_mode == "rezero":
            shapes["_rezero"] = (self.num_layers,)                   # models.py:112-113
        for i, (di, do) in enumerate(self.layer_dims()):
            shapes[f"_linears.{i}.weight"] = (do, di)
            shapes[f"_linears.{i}.bias"] = (do,)
        return shapes


@dataclass
class RenderSpec:
    """Renderer constants (rm.py:116-220; defaults of config/neural_graph_map.yaml)."""
    geometry_mode: str = "nrgbd"
    geometry_factor: float = 20.0
    color_factor: float = 1.0
    truncation_distance: float = 0.1
    range_depth_guided: Optional[float] = None     # None -> truncation (rm.py:169-170)
    num_samples_coarse: int = 8
    num_samples_depth_guided: int = 16
    field_radius: float = 1.0
    scale_mode: str = "unit_cube"
    freespace_weight: float = 40.0
    tsdf_weight: float = 50.0
    termination_weight: float = 0.0
    photometric_weight: float = 1.0
    depth_weight: float = 1.0
    huber_delta: float = 0.05                       # losses.py:63
    photometric_loss: str = "l1"                    # losses.py:26-36: "l1" | "l2" | "gaussian_nll"
    depth_loss: str = "huber"                       # losses.py:60-75: "huber" | "gaussian_nll" | "laplacian_nll"

    def __post_init__(self):
        if self.range_depth_guided is None:
            self.range_depth_guided = self.truncation_distance


# ----------------------------------------------------------------------------------------
# ray sampler (camera.py:186-292, rm.py:513-547)
# ----------------------------------------------------------------------------------------


def ijs_to_directions(ijs: torch.Tensor, cam: CameraSpec, dtype=torch.float32) -> torch.Tensor:
    """Unit view directions, OpenGL convention (camera.py:186-203)."""
    dx = (ijs[..., 1] - cam.cx) / cam.fx
    dy = -((ijs[..., 0] - cam.cy) / cam.fy)
    dz = -torch.ones_like(dx)
    d = torch.stack([dx, dy, dz], -1).to(dtype)
    return torch.nn.functional.normalize(d, dim=-1)


def stratified_distances(near, far, n: int, u):
    """``t_k = (delta*u_k + lin_k*(far-near)) + near`` (camera.py:269-276)."""
    delta = (far - near) / n
    lin = torch.linspace(0.0, 1.0, steps=n + 1, dtype=near.dtype)
    bounds = lin[None] * (far - near)[..., None]
    return (delta[..., None] * u + bounds[..., :-1]) + near[..., None]


def sample_rays(ijs, cam: CameraSpec, near, far, gt, spec: RenderSpec, u_coarse, u_guided=None,
                num_samples=None, return_order=False):
    """Coarse stratum + depth-guided stratum, merged ascending (rm.py:513-545).

    Returns points_cam (...,S,3), distances (...,S) sorted, dirs (...,3)."""
    n_c = spec.num_samples_coarse if num_samples is None else num_samples
    dirs = ijs_to_directions(ijs, cam, near.dtype)
    t = stratified_distances(near, far, n_c, u_coarse)
    n_g = spec.num_samples_depth_guided
    if gt is not None and n_g > 0 and u_guided is not None:
        invalid = (gt == 0.0) | (near > gt) | (far < gt)            # rm.py:522-526
        g_near = torch.where(invalid, near, gt - spec.range_depth_guided)
        g_far = torch.where(invalid, far, gt + spec.range_depth_guided)
        t_g = stratified_distances(g_near, g_far, n_g, u_guided)
        t, order = torch.sort(torch.cat([t, t_g], -1), dim=-1)      # rm.py:538-545
    else:
        order = torch.arange(t.shape[-1]).expand(t.shape)
    pts = dirs.unsqueeze(-2) * t.unsqueeze(-1)                      # camera.py:291
    if return_order:       # test aid: sorted slot -> source element (< n_c: coarse draw, else guided draw - n_c)
        return pts, t, dirs, order
    return pts, t, dirs


def sample_rays_weighted(ijs, cam: CameraSpec, boundaries, weights, u_bin, u_off):
    """Camera.sample_ijs_uniform with weights / boundaries (camera.py:277-289): the bin of a sample is the first whose
    cumulative weight + 1e-3 reaches the first draw, its distance a uniform position inside that bin (second draw);
    no sorting.  Returns points_cam (...,S,3), distances (...,S), dirs (...,3)."""
    dirs = ijs_to_directions(ijs, cam, boundaries.dtype)
    cum = torch.cumsum(weights, dim=-1) + 1e-3
    bins = torch.searchsorted(cum, u_bin)
    deltas = boundaries[..., 1:] - boundaries[..., :-1]
    t = torch.gather(boundaries, -1, bins) + torch.gather(deltas, -1, bins) * u_off
    return dirs.unsqueeze(-2) * t.unsqueeze(-1), t, dirs


def transform_points(p, T, inv=False):
    """p_w = R p + t, or with inv: p_c = R^T (p - t) (utils.py:276-286)."""
    if inv:
        return torch.einsum("...kd,...k->...d", T[..., :3, :3], p - T[..., :3, 3])
    return torch.einsum("...dk,...k->...d", T[..., :3, :3], p) + T[..., :3, 3]


# ----------------------------------------------------------------------------------------
# world -> field-local (models.py:329-339, 278-285; pytorch3d quaternion math restated)
# ----------------------------------------------------------------------------------------


def quat_mul(a, b):
    aw, ax, ay, az = a.unbind(-1)
    bw, bx, by, bz = b.unbind(-1)
    return torch.stack((aw * bw - ax * bx - ay * by - az * bz,
                        aw * bx + ax * bw + ay * bz - az * by,
                        aw * by - ax * bz + ay * bw + az * bx,
                        aw * bz + ax * by - ay * bx + az * bw), -1)


def quat_invert(q):
    return q * q.new_tensor([1.0, -1.0, -1.0, -1.0])


def quat_apply(q, p):
    p4 = torch.cat((p.new_zeros(p.shape[:-1] + (1,)), p), -1)
    return quat_mul(quat_mul(q, p4), quat_invert(q))[..., 1:]


def complex_mul(a, b):
    """(models.py:27-45) raw complex product, real part first"""
    ar, ai = a.unbind(-1)
    br, bi = b.unbind(-1)
    return torch.stack((ar * br - ai * bi, ar * bi + br * ai), -1)


def orientation_apply_inverse(orient, p):
    """`_orientation_apply(_orientation_invert(o), p)` (models.py:236-243): quaternions for 3-D points, complex numbers
    (conjugate = models.py:12-24, `complex_apply` :48-62) for 2-D points"""
    if p.shape[-1] == 2:
        return complex_mul(orient * orient.new_tensor([1.0, -1.0]), p)
    return quat_apply(quat_invert(orient), p)


def scale_local(p, radius, scale_mode):
    if scale_mode == "unit_cube":
        return p / (2 * radius) + 0.5
    if scale_mode == "unit_ball":
        return p / radius
    if scale_mode == "no":
        return p
    raise NotImplementedError(scale_mode)


def world_to_field(p_world, pos, quat, radius, scale_mode):
    """p_world (F,P,3), pos (F,3), quat (F,4) real-first -> local scaled (F,P,3)  (2-D: (F,P,2), (F,2), complex (F,2))."""
    local = p_world - pos.unsqueeze(-2)
    local = orientation_apply_inverse(quat.unsqueeze(-2), local)
    return scale_local(local, radius, scale_mode)


# ----------------------------------------------------------------------------------------
# encodings + MLP (positional_encodings.py:164-276, models.py:143-182)
# ----------------------------------------------------------------------------------------


def encode(x, params, fs: FieldSpec):
    """x (F,P,3) -> (F,P,D)."""
    if fs.encoding == "fourier":
        W = params["_encoding._linear.weight"]                      # (F, D-3, 3)
        feat = torch.sin(torch.einsum("fpc,fdc->fpd", x, W))
        return torch.cat((x, feat), -1) if fs.raw_coords else feat
    if fs.encoding == "nerf":
        octs = torch.arange(fs.start_octave, fs.start_octave + fs.num_octaves, dtype=x.dtype)
        mult = 2 ** octs * math.pi
        sp = x.unsqueeze(-1) * mult                                 # (F,P,3,O)
        lead = x.shape[:-1]
        return torch.cat((torch.sin(sp).reshape(*lead, -1), torch.cos(sp).reshape(*lead, -1)), -1)
    if fs.encoding == "none":
        return x
    if fs.encoding == "permuto":
        return encode_permuto(x, params["_encoding.lattice_values"], params["_encoding.random_shift_per_level"], fs)
    if fs.encoding == "triplane":
        return torch.stack([encode_triplane(x[f], params["_encoding.plane_coef"][f], fs.tri_mode) for f in range(x.shape[0])])
    raise NotImplementedError(fs.encoding)


def encode_triplane(x, plane_coef, mode):
    """TriplaneEncoding.forward (positional_encodings.py:128-161) for one field: x (P,3) in [-1,1], plane_coef
    (3, C, res, res) -> (P, C | 3C).  grid_sample(align_corners=True, padding_mode="border") of the (x,y), (x,z), (y,z)
    projections, then sum / product over the planes or concatenation."""
    coord = torch.stack([x[..., [0, 1]], x[..., [0, 2]], x[..., [1, 2]]], 0).view(3, -1, 1, 2)
    feat = torch.nn.functional.grid_sample(plane_coef, coord, align_corners=True, padding_mode="border")   # (3, C, P, 1)
    if mode == "product":
        return feat.prod(0).squeeze(-1).T
    if mode == "sum":
        return feat.sum(0).squeeze(-1).T
    if mode == "concat":
        return feat.squeeze(-1).reshape(3 * plane_coef.shape[1], -1).T
    raise ValueError(mode)


def permuto_scale_factors(fs: FieldSpec, dtype=torch.float32):
    """(L,3) per-level, per-axis scale: 1 / (sqrt((i+1)(i+2)) * sigma_l), sigma = geomspace(coarsest, finest, L)
    (positional_encodings.py:50 for the sigmas; elevation scaling of Adams et al. 2010 / the
    permutohedral_encoding package)."""
    import numpy as np
    sig = torch.tensor(np.geomspace(fs.coarsest_scale, fs.finest_scale, num=fs.nr_levels), dtype=torch.float64)
    ax = torch.tensor([1.0 / math.sqrt((i + 1) * (i + 2)) for i in range(3)], dtype=torch.float64)
    return (ax[None, :] / sig[:, None]).to(dtype)


def encode_permuto(x, lattice, shift, fs: FieldSpec):
    """Multi-resolution permutohedral-lattice hash encoding; x (F,P,3) in the field frame, lattice
    (F,L,T,2), shift (F,L,3) -> (F,P,2L).

    *** PARITY UNPINNED *** The reference only wraps `permutohedral_encoding.PermutoEncoding`
    (roym899 fork @ bf445adb, pyproject.toml:21), a CUDA package that is not vendored and cannot be
    built or run here, and the reference has no tests for it.  This is a restatement of the published
    algorithm (Adams, Baek, Davis 2010 "Fast high-dimensional filtering using the permutohedral
    lattice"; Rosu & Behnke 2023 "PermutoSDF"): elevate the scaled point onto the hyperplane
    sum = 0 of R^4, round to the nearest remainder-0 lattice point, rank the residuals to find the
    enclosing simplex, barycentric weights, hash every simplex vertex key into a table of T entries per
    level, blend the F=2 features.  It is the definition the HIP kernels are tested against.
    Differentiable w.r.t. `lattice` only (as the CUDA package: no gradient to positions/shifts here)."""
    F, P, _ = x.shape
    L, T = lattice.shape[1], lattice.shape[2]
    d = 3
    scale = permuto_scale_factors(fs, x.dtype)                                   # (L,3)
    cf = (x[:, :, None, :] + shift[:, None, :, :]) * scale[None, None]           # (F,P,L,3)
    el = x.new_zeros(F, P, L, d + 1)
    sm = x.new_zeros(F, P, L)
    for i in range(d, 0, -1):
        el[..., i] = sm - i * cf[..., i - 1]
        sm = sm + cf[..., i - 1]
    el[..., 0] = sm
    v = el * (1.0 / (d + 1))
    up, down = torch.ceil(v) * (d + 1), torch.floor(v) * (d + 1)
    rem0 = torch.where(up - el < el - down, up, down)
    ssum = torch.div(rem0.sum(-1).to(torch.int64), d + 1, rounding_mode="trunc")
    diff = el - rem0
    rank = torch.zeros(F, P, L, d + 1, dtype=torch.int64)
    for i in range(d):
        for j in range(i + 1, d + 1):
            lt = diff[..., i] < diff[..., j]
            rank[..., i] += lt
            rank[..., j] += ~lt
    rank = rank + ssum[..., None]
    rem0 = rem0.to(torch.int64)
    low, high = rank < 0, rank > d
    rem0 = rem0 + low * (d + 1) - high * (d + 1)
    rank = rank + low * (d + 1) - high * (d + 1)
    delta = (el - rem0.to(x.dtype)) * (1.0 / (d + 1))
    bary = x.new_zeros(F, P, L, d + 2)
    bary.scatter_add_(-1, d - rank, delta)
    bary.scatter_add_(-1, d + 1 - rank, -delta)
    bary[..., 0] = bary[..., 0] + 1.0 + bary[..., d + 1]
    out = x.new_zeros(F, P, L, lattice.shape[-1])
    fi = torch.arange(F)[:, None, None].expand(F, P, L)
    li = torch.arange(L)[None, None, :].expand(F, P, L)
    for r in range(d + 1):
        key = rem0[..., :d] + r - (rank[..., :d] > d - r) * (d + 1)               # (F,P,L,3) int64
        h = torch.zeros(F, P, L, dtype=torch.int64)
        for i in range(d):
            h = ((h + key[..., i]) * 2531011) & 0xFFFFFFFF                        # uint32 wrap-around
        idx = h % T
        out = out + lattice[fi, li, idx] * bary[..., r:r + 1]
    return out.reshape(F, P, L * lattice.shape[-1])


def field_mlp(h, params, fs: FieldSpec, pre_out: Optional[list] = None):
    """NeuralField.forward (models.py:143-182): relu on all but the last layer, then the skip connection:
    concat appends the encoding, add adds it to the first D units, rezero scales by a learnt per-layer scalar and
    adds the layer's own input (the encoding for layer 0).  `pre_out` (test aid) collects the hidden layers'
    pre-activations, i.e. the arguments of the ReLUs."""
    n = fs.num_layers
    D = fs.dim_enc
    enc = h
    for i in range(n + 1):
        prev = h
        W = params[f"_linears.{i}.weight"]
        b = params[f"_linears.{i}.bias"]
        h = torch.einsum("fpi,foi->fpo", h, W) + b.unsqueeze(-2)
        if i == n:
            break
        if pre_out is not None:
            pre_out.append(h)
        h = torch.relu(h)
        if fs.skip_mode == "concat":
            h = torch.cat((h, enc), -1)
        elif fs.skip_mode == "add":
            h = torch.cat((h[..., :D] + enc, h[..., D:]), -1)
        elif fs.skip_mode == "rezero":
            rz = params["_rezero"][:, i].view(-1, 1, 1)
            if i == 0:
                h = torch.cat((rz * h[..., :D] + prev, rz * h[..., D:]), -1)
            else:
                h = rz * h + prev
    return h


def field_forward_local(x_local, params, fs: FieldSpec):
    return field_mlp(encode(x_local, params, fs), params, fs)


def field_set_forward_vmap(query_points, pos, quat, params, fs: FieldSpec, radius=1.0,
                           scale_mode="unit_cube"):
    """NeuralFieldSet.forward(use_vmap=True) (models.py:329-345)."""
    if pos is not None:
        x = world_to_field(query_points, pos, quat, radius, scale_mode)
    else:
        x = scale_local(query_points, radius, scale_mode)
    return field_forward_local(x, params, fs)


def field_set_forward_knn(points, pos, quat, params, fs: FieldSpec, radius=1.0,
                          scale_mode="unit_cube", num_knn=2, distance_factor=10.0,
                          outside_value=1.0, mask_radius=None):
    """NeuralFieldSet.forward(use_vmap=False) (models.py:347-405). points (P,3).  `mask_radius` = the forward's
    `field_radius` argument (models.py:293, 368: the inside test only); `radius` = the model's own (scaling, :278-285)."""
    P = points.shape[0]
    K = min(num_knn, pos.shape[0])
    d2 = ((points[:, None, :] - pos[None, :, :]) ** 2).sum(-1)
    d2k, idx = torch.topk(d2, K, dim=-1, largest=False, sorted=True)
    dist = torch.sqrt(d2k)
    inside = dist[:, 0] < (radius if mask_radius is None else mask_radius)
    out = torch.full((P, fs.dim_out), outside_value, dtype=points.dtype)
    if inside.any():
        pi, di, ii = points[inside], dist[inside], idx[inside]
        w = torch.softmax(-distance_factor * di, dim=-1)
        acc = torch.zeros(pi.shape[0], fs.dim_out, dtype=points.dtype)
        for k in range(K):
            fk = ii[:, k]
            loc = orientation_apply_inverse(quat[fk], pi - pos[fk])
            loc = scale_local(loc, radius, scale_mode)
            # evaluate each point with its own field's parameters
            pk = {n: v[fk] for n, v in params.items()}
            o = field_forward_local(loc.unsqueeze(1), pk, fs).squeeze(1)
            acc = acc + w[:, k:k + 1] * o
        out[inside] = acc
    return out


# ----------------------------------------------------------------------------------------
# volume renderer (rm.py:709-799) and render_ijs (rm.py:439-666)
# ----------------------------------------------------------------------------------------


def occupancy_probs(mode, geoms, dists, geometry_factor, neus_isds=None):
    if mode == "density":                                           # rm.py:746-749
        deltas = dists[..., 1:] - dists[..., :-1]
        return 1 - torch.exp(-deltas * torch.relu(geoms[..., :-1])), -1
    if mode == "occupancy":                                         # rm.py:750-752
        return torch.sigmoid(geometry_factor * geoms), None
    if mode == "neus":                                              # rm.py:753-758
        tno = torch.sigmoid(neus_isds * geometry_factor * geoms)
        return torch.clamp_min((tno[..., :-1] - tno[..., 1:]) / (tno[..., :-1] + 1e-5), 0), -1
    if mode == "nrgbd":                                             # rm.py:759-762
        x = geometry_factor * geoms
        return 4 * torch.sigmoid(x) * torch.sigmoid(-x), None
    raise NotImplementedError(mode)


def quadrature(mode, colors, geoms, dists, depths, geometry_factor=20.0, neus_isds=None):
    """Returns colors(...,3), depth, color_var(...,3), depth_var, term_prob, weights."""
    occ, last = occupancy_probs(mode, geoms, dists, geometry_factor, neus_isds)
    lead = geoms.shape[:-1]
    T = torch.cat([occ.new_ones(*lead, 1), torch.cumprod(1 - occ[..., :-1], -1)], -1)
    w = occ * T
    bg = 1 - w.sum(-1)
    c = colors[..., :last, :]
    d = depths[..., :last]
    C = (c * w[..., None]).sum(-2)
    D = (d * w).sum(-1)
    Cv = (w[..., None] * (C.unsqueeze(-2) - c) ** 2).sum(-2)
    Dv = (w * (D[..., None] - d) ** 2).sum(-1)
    return C, D, Cv, Dv, 1.0 - bg, w


def render_ijs(ijs, c2ws, cam: CameraSpec, pos, quat, params, fs: FieldSpec, rs: RenderSpec,
               near, far, gt=None, u_coarse=None, u_guided=None, num_samples=None,
               neus_isds=None, return_samples=False, overwrite_samples_behind_camera=True):
    """Training-style render (use_vmap=True) of rm.py:439-666.

    ijs (F,R,2), c2ws (F,R,4,4) or (4,4), pos (F,3), quat (F,4), near/far/gt (F,R).
    Returns a dict with the Prediction fields (rm.py:59-69)."""
    if c2ws.dim() == 2:
        c2ws = c2ws[None, None]
    pts_cam, t, dirs = sample_rays(ijs, cam, near, far, gt, rs, u_coarse, u_guided, num_samples)
    pts_w = transform_points(pts_cam, c2ws.unsqueeze(-3))
    F, R, S = t.shape
    out = field_set_forward_vmap(pts_w.reshape(F, R * S, 3), pos, quat, params, fs,
                                 rs.field_radius, rs.scale_mode).view(F, R, S, -1)
    colors = rs.color_factor * out[..., :3]
    geoms = out[..., 3]
    depths = -pts_cam[..., 2]
    if overwrite_samples_behind_camera and near is not None and not bool((near >= 0).all()):   # rm.py:494-495
        const = -100.0 if rs.geometry_mode in ("occupancy", "density") else 1.0                # rm.py:614-622
        geoms = torch.where(pts_cam[..., 2] > 0, torch.full_like(geoms, const), geoms)
    tau = rs.truncation_distance
    fs_vec = ts_vec = fs_mask = ts_mask = None
    if rs.freespace_weight != 0.0 and gt is not None:               # rm.py:624-630
        fs_mask = t < (gt[..., None] - tau) * (gt[..., None] != 0.0)
        fs_vec = geoms[fs_mask] * tau
    if rs.tsdf_weight != 0.0 and gt is not None:                    # rm.py:632-639
        deltas = gt[..., None] - t
        ts_mask = (deltas.abs() < tau) & (gt[..., None] != 0.0)
        ts_vec = geoms[ts_mask] * tau - deltas[ts_mask]
    C, D, Cv, Dv, term, w = quadrature(rs.geometry_mode, colors, geoms, t, depths,
                                       rs.geometry_factor, neus_isds)
    pred = dict(rgbds=torch.cat([C, D[..., None]], -1), color_vars=Cv, depth_vars=Dv,
                term_probs=term, freespace_geometry=fs_vec, tsdf_residuals=ts_vec)
    if return_samples:
        pred.update(sample_distances=t, sample_outs=out, sample_weights=w, points_world=pts_w,
                    freespace_mask=fs_mask, tsdf_mask=ts_mask)
    return pred


def render_ijs_knn(ijs, c2ws, cam: CameraSpec, pos, quat, params, fs: FieldSpec, rs: RenderSpec, num_samples: int,
                   near=None, far=None, gt=None, u_coarse=None, u_guided=None, field_ids=None, near_const=0.0,
                   far_const=8.0, num_knn=2, distance_factor=10.0, outside_value=1.0,
                   overwrite_samples_behind_camera=True):
    """`_render_ijs(use_vmap=False)` (rm.py:439-666; the default of the signature, :445): arbitrary rays, every field --
    or the subset `field_ids` (:502-508; the model then blends among THOSE centres and reads parameter rows
    `field_ids[k]`, models.py:390-394) --, `num_samples` = the map's current `_num_samples` (train / eval, :1966-1974),
    scalar near / far from the same mode where no per-ray tensors are given (:513-519), depth-guided second stratum when
    `gt` is given and the config has guided samples (:521-545; the reference's gather needs (F,R,2) rays there),
    kNN-blended evaluation (:586-595), behind-camera overwrite (:494-495, 614-622), free-space / TSDF vectors (:624-639),
    quadrature without neus_isds (:641-647: the neus mode has no kNN branch).

    ijs (...,2); c2ws (4,4) or (...,4,4); pos / quat / params: ALL fields of the map.  Returns the Prediction dict."""
    lead = ijs.shape[:-1]
    dtype = pos.dtype
    near_t = torch.full(lead, near_const, dtype=dtype) if near is None else near           # camera.py:264-267
    far_t = torch.full(lead, far_const, dtype=dtype) if far is None else far
    if near is None or bool((near >= 0).all()):                                            # rm.py:494-495
        overwrite_samples_behind_camera = False
    if field_ids is not None:                                                              # rm.py:502-508
        pos, quat = pos[field_ids], quat[field_ids]
        params = {n: v[field_ids] for n, v in params.items()}
    if c2ws.dim() == 2:
        c2ws = c2ws[None]
    guided = gt is not None and rs.num_samples_depth_guided > 0
    pts_cam, t, _ = sample_rays(ijs, cam, near_t, far_t, gt if guided else None, rs, u_coarse, u_guided if guided else None,
                                num_samples)
    pts_w = transform_points(pts_cam, c2ws.unsqueeze(-3))
    out = field_set_forward_knn(pts_w.reshape(-1, 3), pos, quat, params, fs, rs.field_radius, rs.scale_mode, num_knn,
                                distance_factor, outside_value).view(*t.shape, 4)
    colors = rs.color_factor * out[..., :3]
    geoms = out[..., 3]
    depths = -pts_cam[..., 2]
    if overwrite_samples_behind_camera:
        const = -100.0 if rs.geometry_mode in ("occupancy", "density") else 1.0
        geoms = torch.where(pts_cam[..., 2] > 0, torch.full_like(geoms, const), geoms)
    tau = rs.truncation_distance
    fs_vec = ts_vec = None
    if rs.freespace_weight != 0.0 and gt is not None:
        fs_vec = geoms[t < (gt[..., None] - tau) * (gt[..., None] != 0.0)] * tau
    if rs.tsdf_weight != 0.0 and gt is not None:
        deltas = gt[..., None] - t
        m = (deltas.abs() < tau) & (gt[..., None] != 0.0)
        ts_vec = geoms[m] * tau - deltas[m]
    C, D, Cv, Dv, term, _ = quadrature(rs.geometry_mode, colors, geoms, t, depths, rs.geometry_factor, None)
    return dict(rgbds=torch.cat([C, D[..., None]], -1), color_vars=Cv, depth_vars=Dv, term_probs=term,
                freespace_geometry=fs_vec, tsdf_residuals=ts_vec, sample_distances=t)


# ----------------------------------------------------------------------------------------
# training-target sampler (rm.py:1259-1459)
# ----------------------------------------------------------------------------------------


def project_points_opengl(points_cam, cam: CameraSpec, pixel_center: float = 0.5):
    """Camera.project_points(points, "opengl") (camera.py:119-154, matrix :176-180): continuous (x, y)
    image coordinates for the given pixel-centre convention (cam.cx/cy are stored for pixel centre 0)."""
    cx, cy = cam.cx + pixel_center, cam.cy + pixel_center
    M = torch.tensor([[cam.fx, 0, -cx], [0, -cam.fy, -cy], [0, 0, -1]], dtype=torch.float)
    h = torch.einsum("oi,...i->...o", M, points_cam)
    return h[..., :2] / h[..., 2].unsqueeze(-1)


def depth_to_distance(depths, ijs, cam: CameraSpec):
    """camera.py:319-340: depth along z / z-component of the unit OpenCV ray direction."""
    dx = (ijs[..., 1] - cam.cx) / cam.fx
    dy = (ijs[..., 0] - cam.cy) / cam.fy
    d = torch.nn.functional.normalize(torch.stack([dx, dy, torch.ones_like(dx)], -1), dim=-1)
    return depths / d[..., 2]


def sample_target_mv(cam: CameraSpec, c_c2w, nc_rgbd, frame_cid_to_ncid, positions, num_fields, current_field_ids,
                     num_train_fields, num_rays_per_field, field_radius, draws=None):
    """NeuralGraphMap._sample_target_mv (rm.py:1259-1459): pick fields, find the keyframes that see them,
    sample keyframes and pixels per field and collect the supervision targets.

    Randomness: with draws=None the same torch RNG calls as the reference are made in the same order (global
    generator), so after torch.manual_seed(s) the result equals the reference's; otherwise `draws` supplies
    them: dict(subset_observed, subset_random (or None), offsets (20,3) normalised, frame_cids (F,R), u_xy (F,R,2)).
    Returns (target dict, draws dict, aux dict with the visibility mask / boxes before filtering)."""
    radius = field_radius + 0.0
    num_field_samples = 20
    d = {} if draws is None else draws
    cur = current_field_ids
    n_obs = min(num_train_fields // 2, len(cur))
    sub_obs = d["subset_observed"] if draws is not None else torch.multinomial(torch.ones(len(cur)), n_obs)
    obs_ids = cur[sub_obs]
    n_rand = min(num_train_fields - len(obs_ids), num_fields - len(obs_ids))
    sub_rand = None
    if n_rand > 0:
        dist = torch.ones(num_fields)
        dist[obs_ids] = 0.0
        sub_rand = d["subset_random"] if draws is not None else torch.multinomial(dist, n_rand)
        field_ids = torch.unique(torch.cat((torch.arange(num_fields)[sub_rand], obs_ids)))
    else:
        field_ids = obs_ids
    pos_w = positions[field_ids]
    if draws is not None:
        offsets = d["offsets"]
    else:
        offsets = torch.randn((num_field_samples, 3))
        offsets = offsets / torch.linalg.norm(offsets, dim=-1, keepdim=True)
    samples_w = pos_w.unsqueeze(1) + offsets * radius * 1.0                           # (F,20,3)
    samples_c = transform_points(samples_w.unsqueeze(-2), c_c2w, inv=True)            # (F,20,Nc,3)
    depths = -samples_c[..., 2]
    xy = project_points_opengl(samples_c, cam)                                        # (F,20,Nc,2)
    xi = xy.int()
    valid = (xi[..., 0] >= 0) & (xi[..., 0] < cam.width) & (xi[..., 1] >= 0) & (xi[..., 1] < cam.height)
    F, Nc = len(field_ids), c_c2w.shape[0]
    cids = torch.arange(Nc).expand(F, num_field_samples, -1)
    kf_depths = torch.zeros_like(depths)
    kf_depths[valid] = nc_rgbd[frame_cid_to_ncid[cids[valid]], xi[..., 1][valid].long(), xi[..., 0][valid].long(), 3]
    kf_mask = (depths > 0).any(-2) & (depths < kf_depths).any(-2) & valid.any(-2)     # (F,Nc)
    fmask = kf_mask.any(-1)
    aux = dict(field_ids_all=field_ids, kf_mask_all=kf_mask, min_xy_all=xy.min(1)[0], max_xy_all=xy.max(1)[0])
    kf_mask, field_ids, pos_w, xy = kf_mask[fmask], field_ids[fmask], pos_w[fmask], xy[fmask]
    F, R = len(field_ids), num_rays_per_field
    frame_cids = d["frame_cids"] if draws is not None else torch.multinomial(kf_mask.float(), R, replacement=True)
    wh = torch.tensor((cam.width, cam.height), dtype=torch.float)
    min_xy = xy.min(1)[0].clamp_min(0.0)
    max_xy = torch.minimum(xy.max(1)[0], wh)
    idx = frame_cids[..., None].expand(-1, -1, 2)
    tmin, tmax = torch.gather(min_xy, 1, idx), torch.gather(max_xy, 1, idx)
    u_xy = d["u_xy"] if draws is not None else torch.rand(F, R, 2)
    jis = torch.minimum(((tmax - tmin) * u_xy + tmin).int(), torch.tensor((cam.width - 1, cam.height - 1), dtype=torch.int32))
    ijs = torch.stack((jis[..., 1], jis[..., 0]), -1)
    c2ws = c_c2w[frame_cids]
    pos_c = transform_points(pos_w.unsqueeze(1), c2ws, inv=True)
    dirs = ijs_to_directions(ijs, cam)
    center = (pos_c * dirs).sum(-1)
    near = (center - radius).clamp_min(0.0)
    far = (center + radius).clamp_min(0.0)
    rgbds = nc_rgbd[frame_cid_to_ncid[frame_cids], ijs[..., 0].long(), ijs[..., 1].long()]
    gt = depth_to_distance(rgbds[..., 3], ijs, cam)
    vd = gt != 0.0
    target = dict(ijs=ijs, c2ws=c2ws, near=near, far=far, gt=gt, field_ids=field_ids, rgbds=rgbds,
                  rgb_mask=(rgbds[..., :2] != 0.0).any(-1), depth_mask=(gt > near) & (gt < far) & vd,
                  term_probs=(gt < far).float(), term_mask=(gt > near) & vd)
    used = dict(subset_observed=sub_obs, subset_random=sub_rand, offsets=offsets, frame_cids=frame_cids, u_xy=u_xy)
    return target, used, aux


def sample_target_sv(cam: CameraSpec, rgbd_image, c2w, positions, active_field_ids, num_train_fields, num_rays_per_field,
                     field_radius, draws=None, num_points=50000):
    """NeuralGraphMap._sample_target_sv (rm.py:1461-1583, `update_mode: single_view`): fields and rays from one RGB-D
    frame.  Randomness as in sample_target_mv: with draws=None the reference's three torch.multinomial calls are made in
    its order; otherwise `draws` (subset_points, subset_fields, segments) replays recorded ones."""
    radius = field_radius + 0.0                                            # MARGIN = 0.0, rm.py:1499-1501
    d = draws or {}
    pos_w = positions[active_field_ids]
    pos_c = (pos_w - c2w[:3, 3]) @ c2w[:3, :3]                             # utils.transform_points(..., inv=True), utils.py:279-282
    depth = rgbd_image[..., 3]
    ijs = torch.nonzero(depth)                                             # Camera.depth_to_pointcloud, camera.py:371-386
    dv = depth[ijs[:, 0], ijs[:, 1]]
    points = torch.stack(((ijs[:, 1].float() - cam.cx) * dv / cam.fx, -(ijs[:, 0].float() - cam.cy) * dv / cam.fy, -dv), -1)
    sub = d["subset_points"] if draws else torch.multinomial(torch.ones(len(points)), num_points)     # rm.py:1507
    points, ijs = points[sub], ijs[sub]
    mins, maxs = points.min(0)[0], points.max(0)[0]                        # geometry.AABBs.intersects_aabbs, geometry.py:26-42
    aabb_mask = ((pos_c - radius) <= maxs).all(-1) & ((pos_c + radius) >= mins).all(-1)
    pos_in = pos_c[aabb_mask]
    # geometry.LineSegments(origin, points).intersects_spheres, geometry.py:67-105 (p1 = 0)
    sq = (points * points).sum(-1, keepdim=True)
    sq = torch.where(sq == 0, torch.ones_like(sq), sq)
    t = ((pos_in[:, None, :] * points).sum(-1, keepdim=True) / sq).clamp(0.0, 1.0)
    closest = points * t
    hit = ((pos_in[:, None, :] - closest) ** 2).sum(-1) <= radius ** 2     # (F', N)
    seg_mask = hit.sum(-1) >= num_rays_per_field                           # rm.py:1527-1529
    hit = hit[seg_mask]
    ids, pc = active_field_ids[aabb_mask][seg_mask], pos_in[seg_mask]
    sf = None
    if len(hit) > num_train_fields:                                        # rm.py:1532-1543
        sf = d["subset_fields"] if draws else torch.multinomial(torch.ones(len(hit)), num_train_fields)
        ids, pc, hit = ids[sf], pc[sf], hit[sf]
    segments = d["segments"] if draws else torch.multinomial(hit.float(), num_rays_per_field)   # rm.py:1546
    t_ijs = ijs[segments]
    dirs = ijs_to_directions(t_ijs, cam)
    center = (pc[:, None, :] * dirs).sum(-1)
    near, far = center - radius, center + radius                           # not clamped at 0 here (rm.py:1553-1554)
    rgbds = rgbd_image[t_ijs[..., 0], t_ijs[..., 1]]
    gt = depth_to_distance(rgbds[..., 3], t_ijs, cam)
    dm = gt < far
    target = dict(ijs=t_ijs, c2ws=c2w, near=near, far=far, gt=gt, field_ids=ids, rgbds=rgbds, rgb_mask=dm, depth_mask=dm,
                  term_probs=dm.float(), term_mask=torch.ones_like(dm))
    used = dict(subset_points=sub, subset_fields=sf, segments=segments)
    return target, used


# ----------------------------------------------------------------------------------------
# losses (rm.py:1769-1872, losses.py:10-75)
# ----------------------------------------------------------------------------------------


def compute_losses(pred, target_rgbds, depth_mask, term_mask, term_target, rs: RenderSpec):
    """Global masked means over all fields (rm.py:1787-1872); photometric l1 / l2 / gaussian_nll (losses.py:26-36), depth
    huber / gaussian_nll / laplacian_nll (losses.py:60-75); the loss dict keys carry the mode like rm.py:1827, 1837.
    The variance-weighted modes read the rendered variances (pred["color_vars"], pred["depth_vars"], rm.py:781-790) and
    differentiate through them."""
    m = depth_mask & (pred["term_probs"] > 0.8)                     # rm.py:1787-1788
    loss = {}
    loss["termination"] = ((pred["term_probs"][term_mask] - term_target[term_mask]) ** 2).mean()
    pk, dk = "photometric_" + rs.photometric_loss, "depth_" + rs.depth_loss
    # rm.py:1820-1825 hands the PREDICTION in as `measured_colors` and the target as `rendered_colors`
    p_rgb, t_rgb = pred["rgbds"][m][:, :3], target_rgbds[m][:, :3]
    if rs.photometric_loss == "l1":
        loss[pk] = (p_rgb - t_rgb).abs().mean()
    elif rs.photometric_loss == "l2":
        loss[pk] = ((p_rgb - t_rgb) ** 2).mean()
    elif rs.photometric_loss == "gaussian_nll":                     # losses.py:30-36 (no epsilon on the variance)
        cv = pred["color_vars"][m]
        nlls = 0.5 * (t_rgb - p_rgb) ** 2 / cv + torch.log(torch.sqrt(cv))
        loss[pk] = (p_rgb - t_rgb).abs().mean() if nlls.mean() > 2 else nlls.mean()      # data-dependent switch to L1
    else:
        raise NotImplementedError(rs.photometric_loss)
    p_d, t_d = pred["rgbds"][m][:, 3], target_rgbds[m][:, 3]
    if rs.depth_loss == "huber":
        loss[dk] = torch.nn.functional.huber_loss(p_d, t_d, delta=rs.huber_delta)
    elif rs.depth_loss == "gaussian_nll":                           # losses.py:64-69
        dv = pred["depth_vars"][m] + 1e-15
        loss[dk] = (0.5 * (p_d - t_d) ** 2 / dv + torch.log(torch.sqrt(dv))).mean()
    elif rs.depth_loss == "laplacian_nll":                          # losses.py:70-75
        dv = pred["depth_vars"][m]
        loss[dk] = ((t_d - p_d).abs() / torch.sqrt(0.5 * dv + 1e-6) + 0.5 * torch.log(2 * dv + 1e-6)).mean()
    else:
        raise NotImplementedError(rs.depth_loss)
    total = (rs.termination_weight * loss["termination"]
             + rs.photometric_weight * loss[pk]
             + rs.depth_weight * loss[dk])
    if pred["freespace_geometry"] is not None:
        loss["freespace"] = ((pred["freespace_geometry"] - rs.truncation_distance) ** 2).mean()
        total = total + rs.freespace_weight * loss["freespace"]
    if pred["tsdf_residuals"] is not None:
        loss["tsdf"] = (pred["tsdf_residuals"] ** 2).mean()
        total = total + rs.tsdf_weight * loss["tsdf"]
    loss["combined"] = total
    return loss


# ----------------------------------------------------------------------------------------
# Adam with L2 weight decay and one shared step counter (rm.py:347-389, 668-707, 1183-1221)
# ----------------------------------------------------------------------------------------


def adam_step(param, grad, exp_avg, exp_avg_sq, step: int, lr=1e-3, beta1=0.9, beta2=0.999,
              eps=1e-15, weight_decay=1e-5):
    """One torch.optim.Adam (non-amsgrad, L2-coupled wd) update; `step` is the new count."""
    g = grad + weight_decay * param
    exp_avg = beta1 * exp_avg + (1 - beta1) * g
    exp_avg_sq = beta2 * exp_avg_sq + (1 - beta2) * g * g
    bc1 = 1 - beta1 ** step
    bc2 = 1 - beta2 ** step
    denom = exp_avg_sq.sqrt() / math.sqrt(bc2) + eps
    param = param - (lr / bc1) * exp_avg / denom
    return param, exp_avg, exp_avg_sq


# ----------------------------------------------------------------------------------------
# PSNR (evaluation.py:46-56; torchmetrics PSNR with data_range=1 restated)
# ----------------------------------------------------------------------------------------


def psnr(pred_rgb, target_rgb, crop=0):
    if crop > 0:
        pred_rgb = pred_rgb[crop:-crop, crop:-crop]
        target_rgb = target_rgb[crop:-crop, crop:-crop]
    p = pred_rgb.clamp(0.0, 1.0)
    t = target_rgb.clamp(0.0, 1.0)
    mse = ((p - t) ** 2).mean()
    return float(10.0 * torch.log10(1.0 / mse))


# ----------------------------------------------------------------------------------------
# helpers used by tests / bench
# ----------------------------------------------------------------------------------------


def init_params(fs: FieldSpec, num_fields: int, seed=0, mu=0.0, sigma=4.0,
                identical=False, dtype=torch.float32):
    """Random parameters in the reference's stacked layout.

    ``identical=True`` mimics add_fields (clone of one prototype, models.py:254-257)."""
    g = torch.Generator().manual_seed(seed)
    n = 1 if identical else num_fields
    params = {}
    for name, shape in fs.param_shapes().items():
        if name == "_encoding._linear.weight":
            v = torch.randn(n, *shape, generator=g) * sigma + mu
        elif name == "_encoding.lattice_values":
            v = torch.randn(n, *shape, generator=g) * 0.1
        elif name == "_encoding.random_shift_per_level":
            v = torch.randn(n, *shape, generator=g) * 10.0
        elif name == "_encoding.plane_coef":
            v = torch.randn(n, *shape, generator=g) * 0.5
        elif name == "_rezero":
            v = 0.5 * torch.randn(n, *shape, generator=g)            # the reference initialises zeros (models.py:131-132)
        elif name.endswith("weight"):
            bound = 1.0 / math.sqrt(shape[1])
            v = (torch.rand(n, *shape, generator=g) * 2 - 1) * bound
        else:
            fan_in = dict(fs.param_shapes())[name.replace("bias", "weight")][1]
            bound = 1.0 / math.sqrt(fan_in)
            v = (torch.rand(n, *shape, generator=g) * 2 - 1) * bound
        if identical:
            v = v.expand(num_fields, *shape).clone()
        params[name] = v.to(dtype)
    return params
