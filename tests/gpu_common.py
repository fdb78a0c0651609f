"""Shared helpers of the -m gpu parity tests (renderer / target builders, tolerances)."""
import torch

from neural_graph_mapping_amd import _capi as K  # noqa: F401
from neural_graph_mapping_amd import models as M
from neural_graph_mapping_amd import renderer as Rr
from oracle import ngm_oracle as O

DEV = "cuda"
NRGBD_KW = dict(fx=554.2562584220408, fy=554.2562584220408, cx=319.5, cy=239.5)
NRGBD = O.CameraSpec(640, 480, **NRGBD_KW)


def cu(d):
    return {k: v.to(DEV) for k, v in d.items()}


def close(a, b, rtol=2e-4, atol=2e-5, equal_nan=False):
    torch.testing.assert_close(a.cpu(), b.cpu(), rtol=rtol, atol=atol, equal_nan=equal_nan)


# what the oracle comparisons actually measured, written by tests/conftest.py at the end of a -m gpu session to
# gpurun_out/parity_margins.txt (committed per round as profiles/rNN_parity_margins.txt): every gradient comparison's worst
# relative error next to its bar, and the share of rays kink_free_draws took out of a batch before the comparison
MARGINS = []
KINK = []
SWEEPS = {}        # test function -> dict(run, compared, empty): parametrised sweep instances that reached the loss + gradient comparison


def compare_losses(res, pred, t, rs, rtol=3e-4, atol=1e-6):
    """Loss dict of the fused iteration against `O.compute_losses` on the oracle's prediction, term by term, and the bookkeeping
    of the randomised sweeps.  A loss term whose selection is EMPTY is NaN in the reference (`.mean()` of an empty tensor,
    rm.py:1803-1835) and makes `combined` NaN -- here too, compared as such --, while its gradient is empty: the gradients of
    the other terms stay finite in the reference and are compared by the caller against the oracle's backward like in any other
    case (no early return: every instance of a sweep compares its gradients).  Returns (oracle loss dict, empty term names)."""
    n = dict(photometric=int((t["depth_mask"] & (pred["term_probs"] > 0.8)).sum()), termination=int(t["term_mask"].sum()))
    if pred["freespace_geometry"] is not None:
        n["freespace"] = pred["freespace_geometry"].numel()
    if pred["tsdf_residuals"] is not None:
        n["tsdf"] = pred["tsdf_residuals"].numel()
    empty = sorted(k for k, v in n.items() if v == 0)
    loss = O.compute_losses(pred, t["rgbds"], t["depth_mask"], t["term_mask"], t["term_probs"], rs)
    for k, v in loss.items():
        sel = "photometric" if k.startswith(("photometric_", "depth_")) else k       # both read the selection m (rm.py:1787-1788)
        assert bool(torch.isnan(v)) == (bool(empty) if k == "combined" else sel in empty), (k, empty)
        close(res[k], v.detach(), rtol=rtol, atol=atol, equal_nan=True)
    rec = SWEEPS.setdefault(_test_id().split("[")[0], dict(run=0, compared=0, empty=0))
    rec["run"] += 1
    rec["compared"] += 1
    rec["empty"] += bool(empty)
    return loss, empty



def _test_id():
    import os
    return os.environ.get("PYTEST_CURRENT_TEST", "?").split(" ")[0].split("::", 1)[-1]


def lattice_level_errors(a, b):
    """hash table gradient (F, L, T, 2): worst |a - b| per level as a fraction of that LEVEL's max |grad| -> list of L floats"""
    a, b = a.cpu(), b.cpu()
    L = b.shape[1]
    d = (a - b).abs().permute(1, 0, 2, 3).reshape(L, -1).max(-1).values
    return (d / b.abs().permute(1, 0, 2, 3).reshape(L, -1).max(-1).values.clamp_min(1e-30)).tolist()


def grad_close(a, b, tol=2e-3, name=""):
    scale = b.abs().max().clamp_min(1e-12)
    err = float((a.cpu() - b.cpu()).abs().max() / scale)
    rec = dict(test=_test_id(), name=name, err=err, tol=tol)
    if name == "_encoding.lattice_values" and b.dim() == 4:
        rec["per_level"] = lattice_level_errors(a, b)
    MARGINS.append(rec)
    assert err < tol, (name, err)


# Gradient bars of the HASH network, set from what the comparisons measured (profiles/r05_parity_margins.txt: the whole -m gpu
# suite, worst case per tensor) times a margin of two, instead of round 4's blanket 1e-2:
#   table gradient, against the GLOBAL max |grad|        measured 2.96e-3  -> 6e-3
#   table gradient of levels 0-4, against the LEVEL's max measured 2.8e-3   -> 6e-3
#   table gradient of levels 5-15, against the LEVEL's max measured 1.0e-1  -> 2e-1: sigma_l <= 6e-3 there, the fp32 position
#       error (~5e-7) is up to 5e-3 in lattice coordinates, i.e. in a barycentric weight, and an entry few samples touch with
#       opposite signs inherits it relative to ITS scale -- in the oracle as in the kernels (and in the reference's CUDA
#       package); these levels carry < 1 % of the table gradient's norm
#   first layer's weight (it multiplies the features)      measured 1.83e-3  -> 4e-3
#   every other MLP tensor                                  measured 6.6e-4   -> 2e-3 (the Fourier bar)
HASH_BARS = dict(lattice=6e-3, lattice_coarse_level=6e-3, lattice_fine_level=2e-1, first_weight=4e-3, other=2e-3)


def hash_grad_close(a, b, name, slack=1.0, sigmas=None):
    """`sigmas`: the levels' scales when they are not the default 16-level ladder (a level counts as coarse when its scale is
    >= 0.08: levels 0-4 of geomspace(1, 1e-4, 16)); `slack`: multiplies every bar (the randomised sweep over other ladders,
    table sizes and very short rays, whose statistics the bars were not measured on)"""
    if name == "_encoding.lattice_values" and b.dim() == 4:
        grad_close(a, b, slack * HASH_BARS["lattice"], name)
        for l, e in enumerate(lattice_level_errors(a, b)):
            coarse = (l < 5) if sigmas is None else (float(sigmas[l]) >= 0.08)
            bar = slack * (HASH_BARS["lattice_coarse_level"] if coarse else HASH_BARS["lattice_fine_level"])
            assert e < bar, (name, "level", l, e, bar)
    elif name == "_linears.0.weight":
        grad_close(a, b, slack * HASH_BARS["first_weight"], name)
    else:
        grad_close(a, b, slack * HASH_BARS["other"], name)


def away_from_relu_boundaries(q, pos, quat, params, fs, margin=1e-5, tries=20):
    """Resample query points whose fp64 pre-activations come within `margin` of a ReLU kink, where the
    derivative is discontinuous and fp32 implementations may legitimately disagree."""
    p64 = {k: v.double() for k, v in params.items()}
    g = torch.Generator().manual_seed(99)
    for _ in range(tries):
        x = O.world_to_field(q.double(), pos.double(), quat.double(), 1.0, "unit_cube")
        h = O.encode(x, p64, fs)
        bad = torch.zeros(q.shape[:2], dtype=torch.bool)
        for i in range(fs.num_layers):
            pre = torch.einsum("fpi,foi->fpo", h, p64[f"_linears.{i}.weight"]) + p64[f"_linears.{i}.bias"].unsqueeze(-2)
            bad |= (pre.abs() < margin).any(-1)
            h = torch.relu(pre)
        if not bad.any():
            return q
        q = q.clone()
        q[bad] = (pos[:, None] + 0.5 * torch.randn(q.shape, generator=g))[bad]
    return q


CASES = {
    "g6_train_cfg0": (dict(encoding="fourier", dim_enc=64, num_layers=2), dict(num_samples_coarse=16, num_samples_depth_guided=16)),
    "g6_train_3field": (dict(encoding="fourier", dim_enc=64, num_layers=2),
                        dict(num_samples_coarse=8, num_samples_depth_guided=16, termination_weight=0.5)),
    "g6_train_nerf_l1": (dict(encoding="nerf", num_octaves=8, num_layers=1), dict(num_samples_coarse=8, num_samples_depth_guided=8)),
    # photometric_loss: l2 (losses.py:28-29), fixture G18 from the real reference
    "g18_train_l2": (dict(encoding="fourier", dim_enc=64, num_layers=2),
                     dict(num_samples_coarse=8, num_samples_depth_guided=8, termination_weight=0.5, photometric_loss="l2")),
    # the variance-weighted loss modes (losses.py:30-36, 64-75), fixtures G20 from the real reference; the third sits on the
    # L1 branch of the photometric gaussian_nll's switch (mean NLL > 2)
    "g20_train_gnll_gnll": (dict(encoding="fourier", dim_enc=64, num_layers=2),
                            dict(num_samples_coarse=8, num_samples_depth_guided=8, termination_weight=0.5,
                                 photometric_loss="gaussian_nll", depth_loss="gaussian_nll")),
    "g20_train_l1_lnll": (dict(encoding="fourier", dim_enc=64, num_layers=2),
                          dict(num_samples_coarse=8, num_samples_depth_guided=8, termination_weight=0.5, depth_loss="laplacian_nll")),
    "g20_train_gnll_switch_l1": (dict(encoding="fourier", dim_enc=64, num_layers=2),
                                 dict(num_samples_coarse=8, num_samples_depth_guided=8, termination_weight=0.5,
                                      photometric_loss="gaussian_nll")),
    # cameras inside the field, near < 0: geometry of samples behind the camera overwritten (rm.py:614-622)
    "g10_train_behind_camera": (dict(encoding="fourier", dim_enc=64, num_layers=2),
                                dict(num_samples_coarse=12, num_samples_depth_guided=8, termination_weight=0.5)),
}


def make_renderer(fkw, ckw, num_fields, params=None):
    if fkw["encoding"] == "fourier":
        et = "neural_graph_mapping.positional_encodings.PositionalEncodingFourier"
        ek = dict(dim_in=3, dim_out=fkw["dim_enc"], mu=0.0, sigma=4.0, raw_coords=True)
    elif fkw["encoding"] == "permuto":
        et = "neural_graph_mapping.positional_encodings.PermutohedralEncoding"
        ek = dict(pos_dim=3, log2_hashmap_size=fkw.get("log2_hashmap_size", 12), nr_levels=fkw.get("nr_levels", 16),
                  nr_feat_per_level=2, coarsest_scale=fkw.get("coarsest_scale", 1.0),
                  finest_scale=fkw.get("finest_scale", 1e-4), init_scale=fkw.get("init_scale", 1e-5))
    elif fkw["encoding"] == "triplane":
        et = "neural_graph_mapping.positional_encodings.TriplaneEncoding"
        ek = dict(resolution=fkw.get("resolution", 32), num_components=fkw.get("num_components", 64), init_scale=0.5,
                  mode=fkw.get("tri_mode", "sum"))
    else:
        et = "neural_graph_mapping.positional_encodings.PositionalEncodingNeRF"
        ek = dict(dim_in=3, num_octaves=fkw["num_octaves"], start_octave=0)
    model = M.NeuralFieldSet(dim_points=3, field_type="neural_graph_mapping.models.NeuralField", field_kwargs=dict(
        encoding_type=et, encoding_kwargs=ek, num_layers=fkw["num_layers"], dim_out=4, neus_initial_sd=1.0,
        skip_mode=fkw.get("skip_mode", "no")), num_knn=2,
        distance_factor=10.0, outside_value=1.0, field_radius=float(ckw.get("field_radius", 1.0)), scale_mode="unit_cube",
        weight_dtype=fkw.get("weight_dtype")).to(DEV)                 # one radius for model and map, like the shipped YAML's anchor
    cfg = dict(geometry_mode="nrgbd", geometry_factor=20.0, color_factor=1.0, truncation_distance=0.1, field_radius=1.0,
               termination_weight=0.0, photometric_weight=1.0, photometric_loss="l1", depth_weight=1.0, depth_loss="huber",
               freespace_weight=40.0, tsdf_weight=50.0,
               learning_rate=1e-3, adam_eps=1e-15, adam_weight_decay=1e-5, near_distance=0.0, far_distance=8.0)
    cfg.update(ckw)
    cam = Rr.Camera(640, 480, NRGBD_KW["fx"], NRGBD_KW["fy"], 319.5, 239.5, pixel_center=0.0)
    r = Rr.NeuralGraphRenderer(model, cam, cfg, device=DEV)
    r.add_fields(num_fields)
    if params is not None:
        for k, v in params.items():
            model.all_fields_params[k].copy_(v.to(DEV))
        model.refresh_lp()
    return r


def make_target(t, ids):
    t = cu(t)
    return Rr.Target(ijs=t["ijs"], c2ws=t["c2ws"], near_distances=t["near"], far_distances=t["far"], gt_distances=t["gt"],
                     field_ids=ids.to(DEV), rgbds=t["rgbds"], rgb_mask=t["depth_mask"], depth_mask=t["depth_mask"],
                     term_probs=t["term_probs"], term_mask=t["term_mask"])


def synth_target(F, R, seed=0):
    gen = torch.Generator().manual_seed(seed)
    pos = 0.5 * torch.randn(F, 3, generator=gen)
    quat = torch.nn.functional.normalize(torch.randn(F, 4, generator=gen), dim=-1)
    ijs = torch.stack([torch.randint(0, 480, (F, R), generator=gen), torch.randint(0, 640, (F, R), generator=gen)], -1)
    eye = pos[:, None] + torch.nn.functional.normalize(torch.randn(F, R, 3, generator=gen), dim=-1) * (2 + torch.rand(F, R, 1, generator=gen))
    fwd = torch.nn.functional.normalize(pos[:, None] + 0.3 * torch.randn(F, R, 3, generator=gen) - eye, dim=-1)
    right = torch.nn.functional.normalize(torch.linalg.cross(fwd, torch.tensor([0.0, 1.0, 0.0]).expand_as(fwd)), dim=-1)
    c2w = torch.eye(4).repeat(F, R, 1, 1)
    c2w[..., :3, 0], c2w[..., :3, 1], c2w[..., :3, 2], c2w[..., :3, 3] = right, torch.linalg.cross(right, fwd), -fwd, eye
    d = O.ijs_to_directions(ijs, NRGBD)
    pos_c = torch.einsum("...kd,...k->...d", c2w[..., :3, :3], pos[:, None] - c2w[..., :3, 3])
    center = (pos_c * d).sum(-1)
    near, far = (center - 1).clamp_min(0), (center + 1).clamp_min(0)
    gt = near + (far - near) * (0.1 + 0.8 * torch.rand(F, R, generator=gen))
    gt[torch.rand(F, R, generator=gen) < 0.1] = 0.0
    rgbds = torch.cat([torch.rand(F, R, 3, generator=gen), (gt * d[..., 2].abs())[..., None]], -1)
    dm = (gt > near) & (gt < far) & (gt != 0)
    return pos, quat, dict(ijs=ijs, c2ws=c2w, near=near, far=far, gt=gt, rgbds=rgbds, depth_mask=dm,
                           term_probs=(gt < far).float(), term_mask=(gt > near) & (gt != 0))


def kink_free_draws(t, pos, quat, params, fs, rs, u_c, u_g, margin=5e-5, tries=40, seed=4321, max_neutralised=0.15,
                    neus_isds=None, geometry_margin=2e-7):
    """Redraw the jitter of every SAMPLE whose fp64 hidden pre-activations come within `margin` of a ReLU kink.
    The loss gradient is discontinuous there, so two correct fp32 implementations (different summation order) may put
    the sample on different sides and differ by that sample's whole contribution; away from the kinks the strict
    gradient bar (2e-3 of max |grad|) applies.
    A few samples cannot be moved out of the band (a hidden unit whose pre-activation stays near zero across a whole
    stratum): their rays first get a narrower band (down to margin / 8), and the rays that still hold such samples are taken
    out of the loss (gt = 0, masks false: no term of rm.py:1769-1872 sees them), so a flip there cannot matter.
    The geometry modes with a kink of their own are covered the same way (fp64, band `geometry_margin`): neus clamps
    (tno_k - tno_k+1) / (tno_k + 1e-5) at 0 (rm.py:753-758; pass `neus_isds`) -- a pair of neighbouring samples whose
    transformed geometries agree to within fp32 rounding sits on either side of the clamp depending on the last bit of a
    position (pairs deep in the sigmoid's saturation are left alone: the clamp switches nothing there) -- and density applies a ReLU to the geometry output (rm.py:746-749).
    Returns (u_coarse, u_guided, t) with the offending elements redrawn (t: copy with those rays neutralised).  The share of
    neutralised rays is recorded (KINK -> gpurun_out/parity_margins.txt) and bounded by `max_neutralised` (the metric-size
    comparisons pass 0.02: "M1-size vs oracle" must mean at least 98 % of the batch)."""
    gen = torch.Generator().manual_seed(seed)
    u_c = u_c.clone()
    u_g = None if u_g is None else u_g.clone()
    t = dict(t)
    for k in ("gt", "depth_mask", "term_mask"):
        t[k] = t[k].clone()
    p64 = {k: v.double() for k, v in params.items()}
    n_c = u_c.shape[-1]
    c2ws = t["c2ws"] if t["c2ws"].dim() == 4 else t["c2ws"][None, None]
    dead = torch.zeros(t["gt"].shape, dtype=torch.bool)
    mar = torch.full(t["gt"].shape, margin, dtype=torch.float64)            # per-ray band, see below
    for it in range(tries):
        pts_cam, ts, _, order = O.sample_rays(t["ijs"], NRGBD, t["near"], t["far"], t["gt"] if u_g is not None else None,
                                              rs, u_c, u_g, return_order=True)
        F, R, S = ts.shape
        pts_w = O.transform_points(pts_cam, c2ws.unsqueeze(-3)).double().reshape(F, R * S, 3)
        x = O.world_to_field(pts_w, pos.double(), quat.double(), rs.field_radius, rs.scale_mode)
        pres = []
        out64 = O.field_mlp(O.encode(x, p64, fs), p64, fs, pre_out=pres)
        near0 = torch.full((F, R * S), float("inf"), dtype=torch.float64)
        for pre in pres:
            near0 = torch.minimum(near0, pre.abs().min(-1).values)
        bad = (near0.view(F, R, S) < mar[..., None]) & ~dead[..., None]
        geo = out64[..., 3].view(F, R, S)
        if rs.geometry_mode == "neus" and neus_isds is not None and S > 1:
            k_ = neus_isds.double().view(F, 1, 1) * rs.geometry_factor
            tno = torch.sigmoid(k_ * geo)
            slope = k_ * tno * (1 - tno)          # what a flip of the clamp switches on or off; ~0 where the sigmoid is saturated
            smax = torch.maximum(slope[..., :-1], slope[..., 1:])
            # the band in which the clamp's side is undecided between two fp32 evaluations: an ulp or two of tno (6e-8) plus the
            # difference of the two geometry outputs' errors through the sigmoid's slope (neighbouring samples of one ray round
            # almost alike: measured, kernels and oracle agree on such pairs to ~1e-7)
            pair = ((tno[..., :-1] - tno[..., 1:]).abs() < geometry_margin * smax + 1.5e-7) & (smax > 1e-4)
            bad[..., :-1] |= pair & ~dead[..., None]
        elif rs.geometry_mode == "density":
            bad |= (geo.abs() < geometry_margin) & ~dead[..., None]
        if not bad.any():
            frac = float(dead.float().mean())
            KINK.append(dict(test=_test_id(), rays=int(dead.numel()), samples_per_ray=int(S), neutralised_rays=int(dead.sum()),
                             neutralised_frac=frac, narrowed_rays=int(((mar < margin) & ~dead).sum()),
                             min_margin=float(mar[~dead].min()) if (~dead).any() else margin, redraw_rounds=it))
            assert frac <= max_neutralised, f"{frac:.4f} of the rays neutralised (> {max_neutralised}): not a meaningful comparison"
            return u_c, u_g, t
        if it >= tries - 4:
            dead |= bad.any(-1)
            t["gt"][dead] = 0.0
            t["depth_mask"][dead] = False
            t["term_mask"][dead] = False
            continue
        if it >= 8:
            # dense strata (3 mm in the depth-guided interval) across a SLOWLY varying pre-activation: the band |pre| < margin is
            # wider than a stratum there, some sample always sits in it and no redraw can leave.  Such rays get a narrower band,
            # halved per round down to margin / 8 (6e-6: still several times the 1-2e-6 by which two fp32 evaluations of a
            # pre-activation differ); only what is stuck even then is neutralised.  Both counts are reported.
            stuck = bad.any(-1)
            mar[stuck] = torch.clamp(mar[stuck] * 0.5, min=margin / 8)
        f, r, k = bad.nonzero(as_tuple=True)
        src = order[f, r, k]
        is_c = src < n_c
        u_c[f[is_c], r[is_c], src[is_c]] = torch.rand(int(is_c.sum()), generator=gen)
        if u_g is not None and (~is_c).any():
            u_g[f[~is_c], r[~is_c], src[~is_c] - n_c] = torch.rand(int((~is_c).sum()), generator=gen)
    raise AssertionError("kink_free_draws did not converge")


def ragged_case(F, R, n_c, n_g, fkw, geometry_mode="nrgbd", geometry_factor=20.0, max_neutralised=0.15, fwd_tol=None, grad_tol=2e-3, **ckw_extra):
    torch.manual_seed(F * 1000 + R)
    ckw = dict(num_samples_coarse=n_c, num_samples_depth_guided=n_g, termination_weight=0.3, geometry_mode=geometry_mode,
               geometry_factor=geometry_factor, **ckw_extra)
    pos, quat, t = synth_target(F, R, seed=R)
    fs = O.FieldSpec(**fkw)
    rs = O.RenderSpec(num_samples_coarse=n_c, num_samples_depth_guided=n_g, termination_weight=0.3,
                      geometry_mode=geometry_mode, geometry_factor=geometry_factor,
                      **{k: v for k, v in ckw_extra.items() if k in ("photometric_loss", "depth_loss")})
    params = O.init_params(fs, F, seed=R, sigma=3.0)
    params[f"_linears.{fkw['num_layers']}.weight"] *= 2.0
    u_c, u_g = torch.rand(F, R, n_c), (torch.rand(F, R, n_g) if n_g else None)
    u_c, u_g, t = kink_free_draws(t, pos, quat, params, fs, rs, u_c, u_g, max_neutralised=max_neutralised)
    po = {k: v.clone().requires_grad_() for k, v in params.items()}
    pred = O.render_ijs(t["ijs"], t["c2ws"], NRGBD, pos, quat, po, fs, rs, t["near"], t["far"], t["gt"], u_c, u_g)
    r = make_renderer(fkw, ckw, F, params)
    r.set_field_poses(pos.to(DEV), quat.to(DEV))
    res = r.optimization_iteration(make_target(t, torch.arange(F)), u_c.to(DEV), u_g.to(DEV) if n_g else None,
                                   update=False)
    close(res["prediction"].rgbds, pred["rgbds"].detach(), **(fwd_tol or {}))
    close(res["prediction"].term_probs, pred["term_probs"].detach(), **(fwd_tol or {}))
    loss, _ = compare_losses(res, pred, t, rs, rtol=3e-4 if fwd_tol is None else 10 * fwd_tol["rtol"])
    loss["combined"].backward()        # NaN value or not: the oracle's backward is finite (an empty mean has an empty gradient)
    for k in po:
        grad_close(res["grads"][k], po[k].grad, grad_tol, k)


from _philox_host import host_philox_draws, host_philox_uniform  # noqa: E402,F401  (numpy-only: also imported by the CPU tests)
