"""Generate golden fixtures from the REAL reference (build container only).

    python tests/golden/make_golden.py

Imports /root/reference on CPU through ``_ref_import`` (third-party stubs only), runs the
reference's own functions on seeded inputs and stores inputs + expected outputs as small
``.npz`` files next to this script.  The random numbers the reference draws with
``torch.rand`` (camera.py:274) are recorded by re-seeding and re-drawing the same shapes in
the same order (coarse stratum first, then depth-guided).  Fixtures are data only.
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from _ref_import import NRGBD_CAMERA, build_map, import_reference, make_config  # noqa: E402
import scene  # noqa: E402

rm, models, camera, pe, losses, utils = import_reference()


def npy(d):
    out = {}
    for k, v in d.items():
        if v is None:
            continue
        if isinstance(v, torch.Tensor):
            v = v.detach().cpu().numpy()
        out[k] = np.asarray(v)
    return out


def save(name, **arrays):
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **npy(arrays))
    print(f"{name}: {os.path.getsize(path) / 1024:.1f} KiB")


def rand_quats(n, gen):
    return torch.nn.functional.normalize(torch.randn(n, 4, generator=gen), dim=-1)


def look_at_c2w(eye, target, gen):
    """OpenGL camera-to-world looking from eye to target (-z forward), random roll-free."""
    fwd = torch.nn.functional.normalize(target - eye, dim=-1)
    up = torch.tensor([0.0, 1.0, 0.0]).expand_as(fwd)
    right = torch.nn.functional.normalize(torch.linalg.cross(fwd, up), dim=-1)
    up2 = torch.linalg.cross(right, fwd)
    T = torch.eye(4).repeat(*eye.shape[:-1], 1, 1)
    T[..., :3, 0] = right
    T[..., :3, 1] = up2
    T[..., :3, 2] = -fwd
    T[..., :3, 3] = eye
    return T


def synth_target(F, R, cam, pos, gen, radius=1.0, invalid_frac=0.15, inside=False):
    """Synthetic Target in the spirit of _sample_target_mv (rm.py:1383-1459).  inside=True puts the cameras
    inside the field sphere and leaves near unclamped (negative): samples behind the camera (rm.py:614-622)."""
    ijs = torch.stack([torch.randint(0, cam.height, (F, R), generator=gen),
                       torch.randint(0, cam.width, (F, R), generator=gen)], -1)
    eye_dir = torch.nn.functional.normalize(torch.randn(F, R, 3, generator=gen), dim=-1)
    dist = (0.3 + 0.6 * torch.rand(F, R, 1, generator=gen)) if inside else (2.0 + torch.rand(F, R, 1, generator=gen))
    eye = pos[:, None, :] + eye_dir * dist
    tgt = pos[:, None, :] + 0.3 * torch.randn(F, R, 3, generator=gen)
    c2ws = look_at_c2w(eye, tgt, gen)
    dirs = cam.ijs_to_directions(ijs)
    pos_c = utils.transform_points(pos[:, None, :], c2ws, inv=True)
    center = (pos_c * dirs).sum(-1)
    near = (center - radius) if inside else (center - radius).clamp_min(0.0)
    far = (center + radius).clamp_min(0.0)
    gt_lo = near.clamp_min(0.05) if inside else near
    gt = gt_lo + (far - gt_lo) * (0.1 + 0.8 * torch.rand(F, R, generator=gen))
    sel = torch.rand(F, R, generator=gen)
    gt = torch.where(sel < invalid_frac, torch.zeros_like(gt), gt)           # missing depth
    gt = torch.where((sel >= invalid_frac) & (sel < invalid_frac + 0.05), far + 0.3, gt)
    gt = torch.where((sel >= invalid_frac + 0.05) & (sel < invalid_frac + 0.1),
                     (near - 0.2).clamp_min(0.01), gt)
    rgb = torch.rand(F, R, 3, generator=gen)
    depth = gt * dirs[..., 2].abs()
    rgbds = torch.cat([rgb, depth[..., None]], -1)
    valid = gt != 0.0
    depth_mask = (gt > near) & (gt < far) & valid
    term_probs = (gt < far).float()
    term_mask = (gt > near) & valid
    return dict(ijs=ijs, c2ws=c2ws, near=near, far=far, gt=gt, rgbds=rgbds,
                depth_mask=depth_mask, term_probs=term_probs, term_mask=term_mask)


def make_target(t, field_ids):
    return rm.Target(ijs=t["ijs"], c2ws=t["c2ws"], near_distances=t["near"],
                     far_distances=t["far"], gt_distances=t["gt"], field_ids=field_ids,
                     rgbds=t["rgbds"], rgb_mask=t["depth_mask"], depth_mask=t["depth_mask"],
                     term_probs=t["term_probs"], term_mask=t["term_mask"])


def draw_u(seed, F, R, n_c, n_g):
    torch.manual_seed(seed)
    u_c = torch.rand(F, R, n_c)
    u_g = torch.rand(F, R, n_g) if n_g > 0 else None
    return u_c, u_g


# ------------------------------------------------------------------------------------------
def g1_directions():
    cam = camera.Camera(**NRGBD_CAMERA)
    gen = torch.Generator().manual_seed(1)
    ijs = torch.stack([torch.randint(0, 480, (60,), generator=gen),
                       torch.randint(0, 640, (60,), generator=gen)], -1)
    corners = torch.tensor([[0, 0], [0, 639], [479, 0], [479, 639]])
    ijs = torch.cat([corners, ijs])
    save("g1_directions", ijs=ijs, dirs=cam.ijs_to_directions(ijs))


def g2_g3_sampling():
    cam = camera.Camera(**NRGBD_CAMERA)
    gen = torch.Generator().manual_seed(2)
    F, R, n_c, n_g = 2, 8, 4, 4
    ijs = torch.stack([torch.randint(0, 480, (F, R), generator=gen),
                       torch.randint(0, 640, (F, R), generator=gen)], -1)
    near = 1.0 + torch.rand(F, R, generator=gen)
    far = near + 2.0
    # G2: plain stratified sampling with recorded draws
    torch.manual_seed(20)
    pts, t = cam.sample_ijs_uniform(ijs, n_c, near, far, convention="opengl")
    torch.manual_seed(20)
    u_c = torch.rand(F, R, n_c)
    save("g2_sample_uniform", ijs=ijs, near=near, far=far, u=u_c, points=pts, distances=t)
    # G3: merged coarse + guided incl. gt=0, gt<near, gt>far.  Statements of rm.py:521-545
    # are methods of NeuralGraphMap._render_ijs; their effect is captured through the sorted
    # sample distances recovered from the rendered free-space mask in g6; here we pin the two
    # camera calls and torch.sort on the concatenation.
    gt = near + (far - near) * torch.rand(F, R, generator=gen)
    gt[0, 0] = 0.0
    gt[0, 1] = near[0, 1] - 0.5
    gt[1, 0] = far[1, 0] + 0.5
    rho = 0.1
    mask = (gt == 0.0) + (near > gt) + (far < gt)
    gn, gf = gt - rho, gt + rho
    gn[mask] = near[mask]
    gf[mask] = far[mask]
    torch.manual_seed(30)
    pts_c, t_c = cam.sample_ijs_uniform(ijs, n_c, near, far, convention="opengl")
    pts_g, t_g = cam.sample_ijs_uniform(ijs, n_g, gn, gf, convention="opengl")
    t_all, order = torch.sort(torch.cat([t_c, t_g], -1), dim=-1)
    pts_all = torch.gather(torch.cat([pts_c, pts_g], -2), -2,
                           order.unsqueeze(-1).expand(-1, -1, -1, 3))
    u_c, u_g = draw_u(30, F, R, n_c, n_g)
    save("g3_sample_merged", ijs=ijs, near=near, far=far, gt=gt, rho=np.float32(rho),
         u_coarse=u_c, u_guided=u_g, points=pts_all, distances=t_all)


def g4_field_forward():
    gen = torch.Generator().manual_seed(4)
    F, P = 3, 40
    pos = torch.randn(F, 3, generator=gen)
    quat = rand_quats(F, gen)
    q = pos[:, None, :] + 0.6 * torch.randn(F, P, 3, generator=gen)
    for enc in ("fourier", "nerf"):
        cfg = make_config(encoding=enc, num_octaves=8)
        ngm = build_map(rm, cfg, F, pos, quat, seed=40)
        model = ngm._model
        # de-correlate the fields (add_fields clones one prototype)
        for k, v in model.all_fields_params.items():
            if v.dim() > 1:
                v.add_(0.05 * torch.randn(v.shape, generator=gen))
        model.set_vmap_fields(torch.arange(F))
        with torch.no_grad():
            out = model(q, pos, quat, torch.arange(F), True)
        arrays = {"p::" + k: v for k, v in model.all_fields_params.items()}
        save(f"g4_field_forward_{enc}", query=q, pos=pos, quat=quat, out=out, **arrays)


def g5_quadrature():
    gen = torch.Generator().manual_seed(5)
    for mode in ("nrgbd", "occupancy", "density", "neus"):
        for S in (2, 24, 128):
            cfg = make_config(geometry_mode=mode)
            ngm = build_map(rm, cfg, 1, torch.zeros(1, 3), torch.tensor([[1.0, 0, 0, 0]]), seed=50)
            lead = (2, 5)
            colors = torch.rand(*lead, S, 3, generator=gen)
            geoms = 0.1 * torch.randn(*lead, S, generator=gen)
            geoms[0, 0, 0] = 0.0                       # occ = 1 for nrgbd
            geoms[0, 1] = 5.0                          # |g| large
            geoms[0, 2] = -5.0
            dists = torch.sort(torch.rand(*lead, S, generator=gen) * 3 + 0.5, -1)[0]
            depths = dists * 0.9
            isds = (1.0 / (0.5 + torch.rand(2, 1, 1, generator=gen))) if mode == "neus" else None
            C, D, Cv, Dv, term, w = ngm._quadrature(colors, geoms, dists, depths, isds)
            save(f"g5_quadrature_{mode}_S{S}", colors=colors, geoms=geoms, dists=dists,
                 depths=depths, isds=isds, C=C, D=D, Cv=Cv, Dv=Dv, term=term, w=w,
                 geometry_factor=np.float32(cfg["geometry_factor"]))


def _train_case(name, F, R, n_c, n_g, seed, encoding="fourier", num_layers=2, dim_enc=64,
                termination_weight=0.0, perturb=True, save_samples=False, inside=False, photometric_loss="l1",
                depth_loss="huber", color_shift=0.0):
    cam = camera.Camera(**NRGBD_CAMERA)
    gen = torch.Generator().manual_seed(seed)
    pos = 0.5 * torch.randn(F, 3, generator=gen)
    quat = rand_quats(F, gen)
    cfg = make_config(encoding=encoding, dim_enc=dim_enc, num_layers=num_layers,
                      num_samples_coarse=n_c, num_samples_depth_guided=n_g,
                      termination_weight=termination_weight, photometric_loss=photometric_loss, depth_loss=depth_loss)
    ngm = build_map(rm, cfg, F, pos, quat, seed=seed)
    ngm._camera = cam
    model = ngm._model
    if perturb:
        for k, v in model.all_fields_params.items():
            if v.dim() > 1:
                v.add_(0.05 * torch.randn(v.shape, generator=gen))
        # make geometry non-trivial: larger last-layer weights
        model.all_fields_params[f"_linears.{num_layers}.weight"].mul_(2.0)
    t = synth_target(F, R, cam, pos, gen, inside=inside)
    if color_shift:                      # pushes the photometric gaussian_nll mean over 2: the L1 branch of losses.py:34-35
        t["rgbds"][..., :3] += color_shift
    fids = torch.arange(F)
    target = make_target(t, fids)
    torch.manual_seed(seed + 1000)
    pred = ngm._render_ijs(t["ijs"], t["c2ws"], cam, field_ids=fids, use_vmap=True,
                           near_distances=t["near"], far_distances=t["far"],
                           gt_distances=t["gt"])
    u_c, u_g = draw_u(seed + 1000, F, R, n_c, n_g)
    loss = ngm._compute_losses(target, pred)
    loss["combined"].backward()
    vp = model.vmap_fields_params
    arrays = {"p::" + k: v for k, v in vp.items()}
    arrays.update({"g::" + k: v.grad for k, v in vp.items() if v.grad is not None})
    arrays.update({"t::" + k: v for k, v in t.items()})
    arrays.update({"loss::" + k: v for k, v in loss.items()})
    save(name, pos=pos, quat=quat, u_coarse=u_c, u_guided=u_g,
         pred_rgbds=pred.rgbds, pred_color_vars=pred.color_vars, pred_depth_vars=pred.depth_vars,
         pred_term_probs=pred.term_probs, pred_freespace=pred.freespace_geometry,
         pred_tsdf=pred.tsdf_residuals, **arrays)


def g6_train():
    _train_case("g6_train_cfg0", F=1, R=256, n_c=16, n_g=16, seed=60)
    _train_case("g6_train_3field", F=3, R=24, n_c=8, n_g=16, seed=61, termination_weight=0.5)
    _train_case("g6_train_nerf_l1", F=2, R=16, n_c=8, n_g=8, seed=62, encoding="nerf",
                num_layers=1)


def g18_train_l2():
    """photometric_loss: l2 (losses.py:28-29) -- the one non-default loss mode the fused kernels build"""
    _train_case("g18_train_l2", F=2, R=24, n_c=8, n_g=8, seed=180, termination_weight=0.5, photometric_loss="l2")


def g11_target_sampler():
    """_sample_target_mv (rm.py:1259-1459) on a synthetic keyframe store; the torch RNG draws made inside are
    recorded by wrapping torch.multinomial / randn / rand so that other implementations can replay them."""
    W, H = 64, 48
    cam = camera.Camera(width=W, height=H, fx=55.0, fy=55.0, cx=31.5, cy=23.5, pixel_center=0.0)
    gen = torch.Generator().manual_seed(110)
    NF, NC = 12, 6
    grid = torch.stack(torch.meshgrid(torch.arange(3.0), torch.arange(2.0), torch.arange(2.0), indexing="ij"), -1).reshape(-1, 3)
    pos = grid * 1.1 + 0.05 * torch.randn(NF, 3, generator=gen)
    quat = rand_quats(NF, gen)
    cfg = make_config(num_samples_coarse=4, num_samples_depth_guided=4)
    cfg["num_train_fields"], cfg["num_rays_per_field"] = 8, 16
    ngm = build_map(rm, cfg, NF, pos, quat, seed=110)
    ngm._camera = cam
    eye = pos.mean(0) + torch.nn.functional.normalize(torch.randn(NC, 3, generator=gen), dim=-1) * (3.0 + torch.rand(NC, 1, generator=gen))
    c2w = look_at_c2w(eye[None], (pos.mean(0) + 0.5 * torch.randn(NC, 3, generator=gen))[None], gen)[0]
    rgbd = torch.rand(NC + 2, H, W, 4, generator=gen)
    rgbd[..., 3] = 2.0 + 3.0 * rgbd[..., 3]                       # depth 2..5 m: some fields in front, some behind
    rgbd[:, :8, :10, 3] = 0.0                                     # missing depth
    rgbd[:, :4, :5, :3] = 0.0                                     # black corner: rgb_mask false
    ngm._c_c2w_tensor = c2w
    ngm._nc_rgbd_tensor = rgbd
    ngm._frame_cid_to_ncid = torch.tensor([0, 2, 3, 5, 6, 7])
    ngm._rerun_field_details = None
    cur = torch.tensor([1, 4, 5, 7, 10])
    rec = {}
    orig = dict(multinomial=torch.multinomial, randn=torch.randn, rand=torch.rand)
    count = dict(multinomial=0)

    def wrap(name):
        def f(*a, **k):
            out = orig[name](*a, **k)
            if name == "multinomial":
                rec[["subset_observed", "subset_random", "frame_cids"][count["multinomial"]]] = out
                count["multinomial"] += 1
            elif name == "randn":
                rec["offsets_raw"] = out.clone()          # the reference normalises `out` in place afterwards
            else:
                rec["u_xy"] = out
            return out
        return f
    torch.manual_seed(111)
    torch.multinomial, torch.randn, torch.rand = wrap("multinomial"), wrap("randn"), wrap("rand")
    try:
        t = ngm._sample_target_mv(cur)
    finally:
        torch.multinomial, torch.randn, torch.rand = orig["multinomial"], orig["randn"], orig["rand"]
    assert count["multinomial"] == 3
    off = rec["offsets_raw"] / torch.linalg.norm(rec["offsets_raw"], dim=-1, keepdim=True)
    save("g11_target_sampler", width=np.int64(W), height=np.int64(H), fx=np.float32(55.0), fy=np.float32(55.0),
         cx=np.float32(31.5), cy=np.float32(23.5), positions=pos, c_c2w=c2w, nc_rgbd=rgbd,
         frame_cid_to_ncid=ngm._frame_cid_to_ncid, current_field_ids=cur, num_fields=np.int64(NF),
         num_train_fields=np.int64(8), num_rays_per_field=np.int64(16), field_radius=np.float32(1.0), seed=np.int64(111),
         d_subset_observed=rec["subset_observed"], d_subset_random=rec["subset_random"], d_offsets=off,
         d_frame_cids=rec["frame_cids"], d_u_xy=rec["u_xy"],
         o_ijs=t.ijs, o_c2ws=t.c2ws, o_near=t.near_distances, o_far=t.far_distances, o_gt=t.gt_distances,
         o_field_ids=t.field_ids, o_rgbds=t.rgbds, o_rgb_mask=t.rgb_mask, o_depth_mask=t.depth_mask,
         o_term_probs=t.term_probs, o_term_mask=t.term_mask)


def g16_target_sampler_sv():
    """_sample_target_sv (rm.py:1461-1583, update_mode single_view) on a procedural RGB-D frame (tests/golden/scene.py:
    sv_frame, regenerated from its seed); the three torch.multinomial draws made inside are recorded."""
    cam = camera.Camera(width=scene.SV_W, height=scene.SV_H, fx=scene.SV_FX, fy=scene.SV_FY, cx=scene.SV_CX, cy=scene.SV_CY,
                        pixel_center=0.0)
    gen = torch.Generator().manual_seed(160)
    NF = 14
    img = scene.sv_frame(161)
    # fields scattered through the viewing frustum of an identity-ish camera pose, some outside it
    eye = torch.tensor([0.3, -0.2, 0.1])
    c2w = look_at_c2w(eye[None, None], (eye + torch.tensor([0.0, 0.0, -1.0]) + 0.05 * torch.randn(3, generator=gen))[None, None], gen)[0, 0]
    pts_c = torch.stack((2.0 * torch.rand(NF, generator=gen) - 1.0, 1.2 * torch.rand(NF, generator=gen) - 0.6,
                         -(1.0 + 3.0 * torch.rand(NF, generator=gen))), -1)
    pts_c[:3] *= torch.tensor([3.0, 3.0, 1.0])                            # three of them well outside the frustum
    pos = pts_c @ c2w[:3, :3].T + c2w[:3, 3]
    quat = rand_quats(NF, gen)
    cfg = make_config(num_samples_coarse=4, num_samples_depth_guided=4)
    cfg["num_train_fields"], cfg["num_rays_per_field"], cfg["field_radius"] = 4, 24, 0.35
    ngm = build_map(rm, cfg, NF, pos, quat, seed=160)
    ngm._camera = cam
    active = torch.tensor([0, 1, 2, 3, 4, 5, 6, 8, 9, 10, 12, 13])
    rec, count = {}, dict(n=0)
    orig = torch.multinomial

    def wrapped(*a, **k):
        out = orig(*a, **k)
        rec[count["n"]] = (out, tuple(a[0].shape))
        count["n"] += 1
        return out
    torch.manual_seed(162)
    torch.multinomial = wrapped
    try:
        t = ngm._sample_target_sv(img, c2w, active)
    finally:
        torch.multinomial = orig
    assert count["n"] in (2, 3), count
    draws = dict(d_subset_points=rec[0][0].to(torch.int32))
    if count["n"] == 3:
        draws["d_subset_fields"] = rec[1][0]
    draws["d_segments"] = rec[count["n"] - 1][0]
    assert len(t.field_ids) >= 2, "fixture should train several fields"
    save("g16_target_sampler_sv", frame_seed=np.int64(161), positions=pos, c2w=c2w, active_field_ids=active,
         num_train_fields=np.int64(4), num_rays_per_field=np.int64(24), field_radius=np.float32(0.35), seed=np.int64(162),
         num_candidates=np.int64(rec[count["n"] - 1][1][0]),
         o_ijs=t.ijs, o_c2ws=t.c2ws, o_near=t.near_distances, o_far=t.far_distances, o_gt=t.gt_distances,
         o_field_ids=t.field_ids, o_rgbds=t.rgbds, o_rgb_mask=t.rgb_mask, o_depth_mask=t.depth_mask,
         o_term_probs=t.term_probs, o_term_mask=t.term_mask, **draws)


def g12_skip_modes():
    """NeuralField skip connections (models.py:159-180): vmapped forward and d(sum out * seed)/d(params) for
    add / concat / rezero; H > D exercises the partial add (first D units only)."""
    # (rezero cannot be pinned: the reference's own constructor raises for it -- reset_parameters() calls
    #  self._rezero.zero_() on a leaf parameter outside no_grad, models.py:131-132)
    for mode, dim_enc, dim_mlp in (("add", 32, 48), ("concat", 32, 32), ("add", 64, None), ("concat", 64, None), ("add", 61, 64)):
        F, P = 3, 40
        gen = torch.Generator().manual_seed(120 + len(mode) + dim_enc)
        pos = 0.5 * torch.randn(F, 3, generator=gen)
        quat = rand_quats(F, gen)
        cfg = make_config(encoding="fourier", dim_enc=dim_enc, num_layers=2, dim_mlp_out=dim_mlp, skip_mode=mode)
        ngm = build_map(rm, cfg, F, pos, quat, seed=120)
        model = ngm._model
        for k, v in model.all_fields_params.items():
            if v.dim() > 1 and k != "_rezero":
                v.add_(0.05 * torch.randn(v.shape, generator=gen))
        q = pos[:, None, :] + 0.4 * torch.randn(F, P, 3, generator=gen)
        model.set_vmap_fields(torch.arange(F))
        vp = model.vmap_fields_params
        for v in vp.values():
            v.requires_grad_()
        out = model(q, pos, quat, field_ids=torch.arange(F), use_vmap=True)
        seed = torch.randn(out.shape, generator=gen)
        (out * seed).sum().backward()
        arrays = {"p::" + k: v.detach() for k, v in vp.items()}
        arrays.update({"g::" + k: v.grad for k, v in vp.items() if v.grad is not None})
        save(f"g12_skip_{mode}_D{dim_enc}", query=q, pos=pos, quat=quat, out=out.detach(), seed=seed,
             dim_hidden=np.int64(dim_mlp or dim_enc), **arrays)


def g19_skip_other_encodings():
    """skip connections with the encodings that are pure torch in the reference (models.py:159-169 is encoding-agnostic):
    NeRF octaves (D = 48) and the triplane encoding (sum, C = 32), add and concat, two hidden layers; vmapped forward
    and every parameter gradient of sum(out * seed)."""
    for enc in ("nerf", "triplane"):
        for mode in ("add", "concat"):
            gen = torch.Generator().manual_seed(190 + len(enc) + len(mode))
            F, P = 2, 48
            torch.manual_seed(19)
            if enc == "nerf":
                ek = dict(encoding_type="neural_graph_mapping.positional_encodings.PositionalEncodingNeRF",
                          encoding_kwargs=dict(dim_in=3, num_octaves=8, start_octave=0))
                scale_mode, spread = "unit_cube", 0.8
            else:
                ek = dict(encoding_type="neural_graph_mapping.positional_encodings.TriplaneEncoding",
                          encoding_kwargs=dict(resolution=12, num_components=32, init_scale=0.5, mode="sum"))
                scale_mode, spread = "unit_ball", 2.3
            fs = models.NeuralFieldSet(dim_points=3, field_type="neural_graph_mapping.models.NeuralField", field_kwargs=dict(
                **ek, num_layers=2, dim_out=4, dim_mlp_out=None, skip_mode=mode, initial_geometry_bias=0.0, neus_initial_sd=1.0),
                num_knn=2, distance_factor=10.0, field_radius=1.0, scale_mode=scale_mode, outside_value=1.0)
            fs.add_fields(F)
            for k, v in fs.all_fields_params.items():
                if v.dim() > 1:
                    v.add_(0.1 * torch.randn(v.shape, generator=gen))
            fs.set_vmap_fields(None)
            for v in fs.vmap_fields_params.values():
                v.requires_grad_()
            pos = 0.3 * torch.randn(F, 3, generator=gen)
            quat = rand_quats(F, gen)
            q = pos[:, None] + spread * (torch.rand(F, P, 3, generator=gen) - 0.5)
            seed = torch.randn(F, P, 4, generator=gen)
            out = fs(q, pos, quat, None, True)
            (out * seed).sum().backward()
            vp = fs.vmap_fields_params
            save(f"g19_skip_{mode}_{enc}", query=q, pos=pos, quat=quat, out=out, seed=seed,
                 **{"p::" + k: v for k, v in vp.items()}, **{"g::" + k: v.grad for k, v in vp.items() if v.grad is not None})


def g15_triplane():
    """TriplaneEncoding (positional_encodings.py:69-161) in all three modes: vmapped field forward and every parameter
    gradient of sum(out * seed), points in and slightly outside [-1,1]^3 (border padding), scale_mode unit_ball."""
    for mode, comps in (("sum", 32), ("product", 32), ("concat", 20), ("sum", 64)):
        gen = torch.Generator().manual_seed(150 + comps + len(mode))
        F, P, res = 2, 50, 12
        torch.manual_seed(15)
        fs = models.NeuralFieldSet(dim_points=3, field_type="neural_graph_mapping.models.NeuralField", field_kwargs=dict(
            encoding_type="neural_graph_mapping.positional_encodings.TriplaneEncoding",
            encoding_kwargs=dict(resolution=res, num_components=comps, init_scale=0.5, mode=mode), num_layers=1, dim_out=4,
            dim_mlp_out=None, skip_mode="no", initial_geometry_bias=0.0, neus_initial_sd=1.0), num_knn=2, distance_factor=10.0,
            field_radius=1.0, scale_mode="unit_ball", outside_value=1.0)
        fs.add_fields(F)
        for k, v in fs.all_fields_params.items():
            if v.dim() > 1:
                v.add_(0.1 * torch.randn(v.shape, generator=gen))
        fs.set_vmap_fields(None)
        for v in fs.vmap_fields_params.values():
            v.requires_grad_()
        pos = 0.3 * torch.randn(F, 3, generator=gen)
        quat = rand_quats(F, gen)
        q = pos[:, None] + 2.3 * (torch.rand(F, P, 3, generator=gen) - 0.5)
        seed = torch.randn(F, P, 4, generator=gen)
        out = fs(q, pos, quat, None, True)
        (out * seed).sum().backward()
        vp = fs.vmap_fields_params
        save(f"g15_triplane_{mode}_C{comps}", query=q, pos=pos, quat=quat, out=out, seed=seed, resolution=np.int64(res),
             **{"p::" + k: v for k, v in vp.items()}, **{"g::" + k: v.grad for k, v in vp.items() if v.grad is not None})


def g10_behind_camera():
    """Cameras inside the field sphere, near < 0: _render_ijs overwrites the geometry of the samples behind the
    camera (rm.py:494-495, 614-622) and no gradient flows through them."""
    _train_case("g10_train_behind_camera", F=2, R=48, n_c=12, n_g=8, seed=70, termination_weight=0.5, inside=True)


def g7_adam():
    """10 iterations of the reference's sparse-Adam plumbing with changing active sets."""
    cam = camera.Camera(**NRGBD_CAMERA)
    gen = torch.Generator().manual_seed(7)
    NF, R, n_c, n_g = 4, 16, 4, 8
    pos = 0.5 * torch.randn(NF, 3, generator=gen)
    quat = rand_quats(NF, gen)
    cfg = make_config(num_samples_coarse=n_c, num_samples_depth_guided=n_g, dim_enc=32)
    torch.manual_seed(70)
    ngm = rm.NeuralGraphMap(cfg)
    ngm._camera = cam
    ngm._optimizer = torch.optim.Adam([torch.zeros((), requires_grad=True)],
                                      lr=cfg["learning_rate"], eps=cfg["adam_eps"],
                                      weight_decay=cfg["adam_weight_decay"])
    ngm._global_map_dict["num"] = NF
    ngm._global_map_dict["positions"][:NF] = pos
    ngm._global_map_dict["orientations"][:NF] = quat
    ngm._add_fields(NF)
    with torch.no_grad():
        for k, v in ngm._model.all_fields_params.items():
            if v.dim() > 1:
                v.add_(0.05 * torch.randn(v.shape, generator=gen))
    p0 = {"p0::" + k: v.clone() for k, v in ngm._model.all_fields_params.items()}
    active_sets = [[0, 1], [1, 2], [0, 1], [3, 2], [0, 3], [1, 2], [2, 3], [0, 1], [1, 3], [0, 2]]
    steps = {}
    for it, ids in enumerate(active_sets):
        fids = torch.tensor(ids)
        t = synth_target(len(ids), R, cam, pos[fids], gen)
        target = make_target(t, fids)
        torch.manual_seed(700 + it)
        pred = ngm._render_ijs(t["ijs"], t["c2ws"], cam, field_ids=fids, use_vmap=True,
                               near_distances=t["near"], far_distances=t["far"],
                               gt_distances=t["gt"])
        u_c, u_g = draw_u(700 + it, len(ids), R, n_c, n_g)
        loss = ngm._compute_losses(target, pred)
        ngm._update_step(loss, fids)
        steps.update({f"it{it}::" + k: v for k, v in t.items()})
        steps[f"it{it}::u_coarse"] = u_c
        steps[f"it{it}::u_guided"] = u_g
        steps[f"it{it}::field_ids"] = fids
        steps[f"it{it}::loss"] = loss["combined"]
    p1 = {"p1::" + k: v for k, v in ngm._model.all_fields_params.items()}
    m1 = {"m1::" + k: ngm._optim_state[k]["exp_avg"] for k in ngm._optim_state}
    v1 = {"v1::" + k: ngm._optim_state[k]["exp_avg_sq"] for k in ngm._optim_state}
    save("g7_adam", pos=pos, quat=quat, num_iters=np.int64(len(active_sets)),
         **p0, **p1, **m1, **v1, **steps)


def _reference_trainer(F, pos, quat, n_c, n_g, seed, perturb_seed=None, scale=1.0):
    """A reference NeuralGraphMap with F fields and the optimizer plumbing of g7 (rm.py:347-362)."""
    cfg = make_config(num_samples_coarse=n_c, num_samples_depth_guided=n_g, dim_enc=64)
    torch.manual_seed(seed)
    ngm = rm.NeuralGraphMap(cfg)
    ngm._camera = camera.Camera(**NRGBD_CAMERA)
    ngm._optimizer = torch.optim.Adam([torch.zeros((), requires_grad=True)], lr=cfg["learning_rate"], eps=cfg["adam_eps"],
                                      weight_decay=cfg["adam_weight_decay"])
    ngm._global_map_dict["num"] = F
    if ngm._global_map_dict["positions"].shape[0] < F:
        ngm._extend_map_dict(F)
    ngm._global_map_dict["positions"][:F] = pos
    ngm._global_map_dict["orientations"][:F] = quat
    ngm._add_fields(F)
    proto = {k: v[0].clone() for k, v in ngm._model.all_fields_params.items()}
    with torch.no_grad():
        if perturb_seed is not None:                       # fields that differ: the recipe the tests replay (scene.py)
            for k, v in ngm._model.all_fields_params.items():
                v.copy_(scene.perturbed_init(proto[k], F, k, perturb_seed))
        if scale != 1.0:
            for v in ngm._model.all_fields_params.values():
                v.mul_(scale)
    return ngm, proto


def _reference_iteration(ngm, t, fids, seed_u):
    target = make_target(t, fids)
    torch.manual_seed(seed_u)                               # the torch.rand draws of camera.py:274: coarse, then guided
    pred = ngm._render_ijs(t["ijs"], t["c2ws"], ngm._camera, field_ids=fids, use_vmap=True, near_distances=t["near"],
                           far_distances=t["far"], gt_distances=t["gt"])
    loss = ngm._compute_losses(target, pred)
    ngm._update_step(loss, fids)
    return float(loss["combined"].detach())


def g13_training_run():
    """Long training runs of the REAL reference on the learnable sphere scene (tests/golden/scene.py regenerates every
    batch from its seed, so the fixture holds no per-iteration data).
      A  cfg0 (1 field, 256 rays x (16+16) samples, Fourier 2x64): 100 iterations; parameters after 10 / 30 / 100
         iterations, moments after 100, every loss -- and the SAME run started from parameters scaled by (1 + 1e-7):
         the reference's own sensitivity to a one-ulp change, which is the yardstick for "same trajectory".
      B  an ensemble of fields trained together (one batch, rm.py:1164-1186): held-out PSNR of every field at
         checkpoints over the second half of the run -> the ensemble-mean PSNR the kernels must reach within 0.1 dB.
    """
    n_c = n_g = 16
    R = 256
    # ---------------------------------------------------------------- A
    gen = torch.Generator().manual_seed(13)
    pos = 0.5 * torch.randn(1, 3, generator=gen)
    quat = rand_quats(1, gen)
    fids = torch.arange(1)
    out = {}
    for tag, scale in (("a", 1.0), ("b", 1.0 + 1e-7)):
        ngm, proto = _reference_trainer(1, pos, quat, n_c, n_g, seed=130, perturb_seed=131, scale=scale)
        if tag == "a":
            out.update({"A::p0::" + k: v.clone() for k, v in ngm._model.all_fields_params.items()})
        losses = []
        for it in range(scene.A_ITERS):
            t = scene.sphere_scene_batch(1, R, pos, scene.A_BATCH_SEED + it)
            losses.append(_reference_iteration(ngm, t, fids, scene.A_U_SEED + it))
            if it + 1 in scene.A_CHECKPOINTS:
                out.update({f"A::{tag}{it + 1}::" + k: v.clone() for k, v in ngm._model.all_fields_params.items()})
        out[f"A::{tag}::losses"] = np.array(losses, dtype=np.float32)
        if tag == "a":
            out.update({"A::m::" + k: ngm._optim_state[k]["exp_avg"].clone() for k in ngm._optim_state})
            out.update({"A::v::" + k: ngm._optim_state[k]["exp_avg_sq"].clone() for k in ngm._optim_state})
    save("g13_train_cfg0", pos=pos, quat=quat, **out)
    # ---------------------------------------------------------------- B
    F, iters = scene.B_FIELDS, scene.B_ITERS
    gen = torch.Generator().manual_seed(14)
    pos = 3.0 * torch.randn(F, 3, generator=gen)
    quat = rand_quats(F, gen)
    phase = 6.28 * torch.rand(F, 3, generator=gen)
    fids = torch.arange(F)
    ngm, proto = _reference_trainer(F, pos, quat, n_c, n_g, seed=140, perturb_seed=141)
    init_chk = scene.checksum(ngm._model.all_fields_params)
    th = scene.sphere_scene_batch(F, scene.B_HELD_OUT_RAYS, pos, scene.B_HELD_OUT_SEED, phase=phase)
    losses, psnrs, derrs = [], [], []
    import time
    t0 = time.time()
    for it in range(iters):
        t = scene.sphere_scene_batch(F, R, pos, scene.B_BATCH_SEED + it, phase=phase)
        losses.append(_reference_iteration(ngm, t, fids, scene.B_U_SEED + it))
        if it + 1 >= scene.B_EVAL_FROM and (it + 1) % scene.B_EVAL_EVERY == 0:
            with torch.no_grad():
                torch.manual_seed(scene.B_HELD_OUT_U_SEED)
                pred = ngm._render_ijs(th["ijs"], th["c2ws"], ngm._camera, field_ids=fids, use_vmap=True,
                                       near_distances=th["near"], far_distances=th["far"], gt_distances=th["gt"])
            ps, de = scene.held_out_scores(pred.rgbds, th)
            psnrs.append(ps)
            derrs.append(de)
        if (it + 1) % 50 == 0:
            print(f"  g13 B: {it + 1}/{iters}  {time.time() - t0:.0f} s  loss {losses[-1]:.4f}"
                  + (f"  psnr {torch.stack(psnrs).mean():.3f}" if psnrs else ""), flush=True)
    final = {"B::p1::" + k: v[:2].clone() for k, v in ngm._model.all_fields_params.items()}      # two fields, for the record
    save("g13_train_ensemble", pos=pos, quat=quat, phase=phase, losses=np.array(losses, dtype=np.float32),
         psnr=torch.stack(psnrs), depth_err=torch.stack(derrs), init_checksum=np.float64(init_chk),
         **{"proto::" + k: v for k, v in proto.items()}, **final)


def g8_knn():
    gen = torch.Generator().manual_seed(8)
    NF, P = 3, 200
    pos = torch.tensor([[0.0, 0.0, 0.0], [0.9, 0.1, 0.0], [0.2, 1.0, -0.3]])
    quat = rand_quats(NF, gen)
    cfg = make_config()
    ngm = build_map(rm, cfg, NF, pos, quat, seed=80)
    model = ngm._model
    for k, v in model.all_fields_params.items():
        if v.dim() > 1:
            v.add_(0.05 * torch.randn(v.shape, generator=gen))
    pts = 2.2 * (torch.rand(P, 3, generator=gen) - 0.5) + torch.tensor([0.4, 0.4, -0.1])
    pts[:5] += 10.0                                  # far outside every field
    with torch.no_grad():
        out = model(pts, pos, quat, None, False)
    arrays = {"p::" + k: v for k, v in model.all_fields_params.items()}
    save("g8_knn", points=pts, pos=pos, quat=quat, out=out, **arrays)


def g9_render_image():
    gen = torch.Generator().manual_seed(9)
    cam_kw = dict(width=32, height=24, fx=27.7, fy=27.7, cx=15.5, cy=11.5, pixel_center=0.0)
    cam = camera.Camera(**cam_kw)
    NF = 3
    pos = torch.tensor([[0.0, 0.0, -2.5], [0.9, 0.1, -2.8], [-0.7, 0.3, -2.2]])
    quat = rand_quats(NF, gen)
    cfg = make_config(eval_far_distance=5.0, eval_num_samples=48)
    ngm = build_map(rm, cfg, NF, pos, quat, seed=90)
    model = ngm._model
    for k, v in model.all_fields_params.items():
        if v.dim() > 1:
            v.add_(0.05 * torch.randn(v.shape, generator=gen))
    model.all_fields_params["_linears.2.weight"].mul_(3.0)
    c2w = torch.eye(4)
    ngm.eval()
    # G14: the same map as a checkpoint in the reference's own layout -- the dict NeuralGraphMap.save_model hands to
    # torch.save (rm.py:2147-2156; its yoco config dump is control plane and not part of the interchange)
    torch.save({"map_dict": ngm._global_map_dict, "all_fields_params": ngm._model.all_fields_params,
                "state_dict": ngm._model.state_dict()}, os.path.join(HERE, "g14_checkpoint_reference_layout.pt"))
    torch.manual_seed(900)
    rgbd, dvar = ngm.render_image(c2w, cam)
    torch.manual_seed(900)
    u = torch.rand(24 * 32, 48)
    target = torch.rand(24, 32, 3, generator=gen)
    arrays = {"p::" + k: v for k, v in model.all_fields_params.items()}
    save("g9_render_image", pos=pos, quat=quat, c2w=c2w, u=u, rgbd=rgbd, dvar=dvar,
         target_rgb=target, cam=np.array([cam_kw[k] for k in
                                         ("width", "height", "fx", "fy", "cx", "cy")]),
         eval_far=np.float32(5.0), eval_num_samples=np.int64(48), **arrays)


def g17_extract_mesh():
    """NeuralGraphMap._extract_mesh (rm.py:2186-2384) run for real, with the two un-vendored pytorch3d calls replaced:
    `marching_cubes` by oracle/mesh_oracle.py behind pytorch3d's documented interface (volume (N, D, H, W) -> per batch
    element vertices in LOCAL coordinates, x <-> W, y <-> H, z <-> D, normalised to [-1, 1]; faces), `save_ply` by a
    recorder.  What the fixture pins is everything the reference does itself: bounding box and grid axes, the block
    loop (BLOCK_SIZE = 200 with one shared plane, two blocks along x here), the volume handed to marching cubes (sign:
    low_is_inside), the isolevel, the affine map of the local vertices back to the world, the second evaluation with
    radius + 0.1 for the colours, the concatenation / face offsets, and what is handed to save_ply / np.savetxt.
    Stored subsampled (the volume has 2.4 M entries): seeded index sets + values, and totals."""
    import pathlib
    import tempfile
    sys.path.insert(0, os.path.dirname(HERE))
    sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
    from oracle import mesh_oracle as MO
    gen = torch.Generator().manual_seed(17)
    NF, radius, res = 4, 0.5, 0.016
    pos = torch.tensor([[0.0, 0.0, 0.0], [0.8, 0.05, -0.04], [1.6, -0.06, 0.03], [2.4, 0.02, 0.05]])
    quat = rand_quats(NF, gen)
    cfg = make_config(field_radius=radius)
    ngm = build_map(rm, cfg, NF, pos, quat, seed=170)
    model = ngm._model
    for k, v in model.all_fields_params.items():
        if v.dim() > 1:
            v.add_(0.05 * torch.randn(v.shape, generator=gen))
    model.all_fields_params["_linears.2.weight"].mul_(3.0)
    model.all_fields_params["_linears.2.bias"][:, 3] = -0.02            # geometry crosses zero inside the fields
    ngm.eval()
    rec = dict(volumes=[], isolevels=[], raw=[], ply=None)

    def mc_stub(volume, isolevel):
        vol = volume[0].numpy()
        v, f = MO.marching_cubes(vol, float(isolevel))
        n1 = np.array(vol.shape, np.float32) - 1.0
        loc = v / n1 * 2.0 - 1.0                                                   # (ix, iy, iz) -> [-1, 1] per axis
        p3d = np.stack((loc[:, 2], loc[:, 1], loc[:, 0]), -1).astype(np.float32)    # pytorch3d: x <-> last volume axis
        rec["volumes"].append(volume[0].clone()); rec["isolevels"].append(float(isolevel))
        rec["raw"].append(torch.from_numpy(v.copy()))
        return [torch.from_numpy(p3d)], [torch.from_numpy(f)]

    def ply_stub(fp, verts, faces, verts_colors, ascii, colors_as_uint8, verts_normals):
        rec["ply"] = dict(verts=verts.clone(), faces=faces.clone(), colors=verts_colors.clone(), ascii=ascii,
                          colors_as_uint8=colors_as_uint8, normals=verts_normals)

    old_mc, old_ply = rm.marching_cubes, rm.save_ply
    rm.marching_cubes, rm.save_ply = mc_stub, ply_stub
    try:
        with tempfile.TemporaryDirectory() as d:
            path = pathlib.Path(d) / "mesh.ply"
            ngm._extract_mesh(path, resolution=res)
            fields_txt = np.loadtxt(str(pathlib.Path(d) / "mesh_fields.txt")).reshape(-1, 3)
    finally:
        rm.marching_cubes, rm.save_ply = old_mc, old_ply
    assert rec["ply"] is not None and len(rec["volumes"]) >= 2, "expected a surface and at least two blocks"
    out = {}
    rng = np.random.default_rng(170)
    v_all, c_all = rec["ply"]["verts"], rec["ply"]["colors"]
    off = 0
    for b, (vol, raw) in enumerate(zip(rec["volumes"], rec["raw"])):
        flat = vol.reshape(-1)
        sel = np.sort(rng.choice(flat.numel(), size=4096, replace=False))
        out[f"b{b}_shape"] = np.array(vol.shape, np.int64)
        out[f"b{b}_vol_idx"] = sel.astype(np.int64)
        out[f"b{b}_vol_val"] = flat[sel]
        out[f"b{b}_vol_sum"] = np.float64(flat.double().sum())
        out[f"b{b}_vol_inside"] = np.int64((flat > rec["isolevels"][b]).sum())
        nvb = raw.shape[0]
        vs = np.sort(rng.choice(nvb, size=min(2048, nvb), replace=False))
        out[f"b{b}_num_verts"] = np.int64(nvb)
        out[f"b{b}_vert_idx"] = vs.astype(np.int64)
        out[f"b{b}_vert_grid"] = raw[vs]                       # marching-cubes output: grid-index coordinates (ix, iy, iz)
        out[f"b{b}_vert_world"] = v_all[off + vs]              # what the reference made of them (xyz, world)
        out[f"b{b}_vert_color"] = c_all[off + vs]
        off += nvb
    assert off == v_all.shape[0]
    faces = rec["ply"]["faces"]
    save("g17_extract_mesh", pos=pos, quat=quat, field_radius=np.float32(radius), resolution=np.float64(res),
         isolevel=np.float32(rec["isolevels"][0]), num_blocks=np.int64(len(rec["volumes"])),
         num_verts=np.int64(v_all.shape[0]), num_faces=np.int64(faces.shape[0]), faces_max=np.int64(faces.max()),
         verts_min=v_all.min(0)[0], verts_max=v_all.max(0)[0], verts_sum=v_all.double().sum(0),
         colors_sum=c_all.double().sum(0), fields_txt=fields_txt.astype(np.float64),
         ply_ascii=np.int64(bool(rec["ply"]["ascii"])), ply_colors_as_uint8=np.int64(bool(rec["ply"]["colors_as_uint8"])),
         **{"p::" + k: v for k, v in model.all_fields_params.items()}, **out)


def g20_train_nll():
    """the variance-weighted loss modes (losses.py:30-36, 64-75): gradients flow through the rendered colour / depth variances
    (rm.py:781-790).  Three configurations: gaussian_nll for both losses; l1 + laplacian_nll; and gaussian_nll photometric
    with targets far off, where the reference's data-dependent switch (mean NLL > 2) returns the L1 loss instead."""
    _train_case("g20_train_gnll_gnll", F=2, R=24, n_c=8, n_g=8, seed=200, termination_weight=0.5,
                photometric_loss="gaussian_nll", depth_loss="gaussian_nll")
    _train_case("g20_train_l1_lnll", F=2, R=24, n_c=8, n_g=8, seed=201, termination_weight=0.5,
                photometric_loss="l1", depth_loss="laplacian_nll")
    _train_case("g20_train_gnll_switch_l1", F=2, R=24, n_c=8, n_g=8, seed=202, termination_weight=0.5,
                photometric_loss="gaussian_nll", depth_loss="huber", color_shift=40.0)


def g21_field_radius_override():
    """`NeuralFieldSet.forward(..., field_radius=r + 0.1)` exactly as `_extract_mesh` calls it for the vertex colours
    (run_mapping.py:2320-2332): the argument widens only the inside test (models.py:333-334, 368); the local coordinates
    keep the set's own scaling (models.py:278-285).  Both branches, radius 0.8 (so that 1 / (2 r) is not a power of two),
    a third of the points placed in the shell r <= |p - c| < r + 0.1 of their nearest centre, some beyond it."""
    gen = torch.Generator().manual_seed(21)
    NF, P, r = 4, 360, 0.8
    pos = torch.tensor([[0.0, 0.0, 0.0], [0.9, 0.1, 0.0], [0.2, 1.0, -0.3], [3.0, 3.0, 3.0]])
    quat = rand_quats(NF, gen)
    cfg = make_config(field_radius=r)
    ngm = build_map(rm, cfg, NF, pos, quat, seed=210)
    model = ngm._model
    assert model._field_radius == r and model._scale_mode == "unit_cube"
    for k, v in model.all_fields_params.items():
        if v.dim() > 1:
            v.add_(0.05 * torch.randn(v.shape, generator=gen))
    # points on shells around randomly chosen centres: inside, in the widened shell, outside both
    c = torch.randint(0, NF, (P,), generator=gen)
    d = torch.nn.functional.normalize(torch.randn(P, 3, generator=gen), dim=-1)
    rad = torch.cat((r * torch.rand(P // 3, generator=gen), r + 0.1 * torch.rand(P // 3, generator=gen),
                     r + 0.1 + 0.3 * torch.rand(P - 2 * (P // 3), generator=gen)))
    pts = pos[c] + rad[:, None] * d
    with torch.no_grad():
        out_knn = model(pts, pos, quat, None, use_vmap=False, field_radius=r + 0.1)
        out_knn_default = model(pts, pos, quat, None, use_vmap=False)
        ids = torch.tensor([2, 0, 3])
        model.set_vmap_fields(ids)
        q = pos[ids][:, None, :] + 0.5 * torch.randn(3, 50, 3, generator=gen)
        out_vmap = model(q, pos[ids], quat[ids], ids, use_vmap=True, field_radius=r + 0.1)
        out_vmap_default = model(q, pos[ids], quat[ids], ids, use_vmap=True)
    assert torch.equal(out_vmap, out_vmap_default)        # the vmap branch never reads the argument
    n_shell = int(((out_knn != out_knn_default).any(-1)).sum())
    assert n_shell > P // 6, n_shell                      # the shell points really differ between the two calls
    arrays = {"p::" + k: v for k, v in model.all_fields_params.items()}
    save("g21_field_radius_override", points=pts, pos=pos, quat=quat, radius=np.float32(r), mask_radius=np.float32(r + 0.1),
         out_knn=out_knn, out_knn_default=out_knn_default, vmap_ids=ids, query=q, out_vmap=out_vmap, **arrays)


def g22_fields_2d():
    """`NeuralFieldSet(dim_points=2)` (models.py:236-238): complex orientations (`complex_apply` :48-62), 2-D Fourier
    (raw coordinates) and NeRF-octave encodings, both branches of `forward`.  Orientations are NOT all of unit modulus
    (the reference multiplies raw complex numbers: a modulus scales the local coordinates)."""
    gen = torch.Generator().manual_seed(22)
    NF, P, r = 5, 300, 0.8
    pos = torch.tensor([[0.0, 0.0], [0.9, 0.1], [0.2, 1.0], [-0.7, 0.6], [3.0, 3.0]])
    ang = 2 * torch.pi * torch.rand(NF, generator=gen)
    mod = torch.tensor([1.0, 1.0, 0.9, 1.1, 1.0])
    comp = torch.stack((mod * torch.cos(ang), mod * torch.sin(ang)), -1)
    comp[1] = torch.tensor([-1.0, 0.0])                      # the branch cut of the complex square root
    for enc in ("fourier", "nerf"):
        if enc == "fourier":
            et, ek = "neural_graph_mapping.positional_encodings.PositionalEncodingFourier", dict(
                dim_in=2, dim_out=40, mu=0.0, sigma=4.0, raw_coords=True)
        else:
            et, ek = "neural_graph_mapping.positional_encodings.PositionalEncodingNeRF", dict(dim_in=2, num_octaves=6, start_octave=0)
        torch.manual_seed(220)
        model = models.NeuralFieldSet(
            dim_points=2, field_type="neural_graph_mapping.models.NeuralField",
            field_kwargs=dict(encoding_type=et, encoding_kwargs=ek, num_layers=2, dim_out=4, dim_mlp_out=64, skip_mode="no",
                              initial_geometry_bias=0.0, neus_initial_sd=None),
            num_knn=2, distance_factor=10.0, outside_value=1.0, field_radius=r, scale_mode="unit_cube")
        model.add_fields(NF)
        for k, v in model.all_fields_params.items():
            if v.dim() > 1:
                v.add_(0.05 * torch.randn(v.shape, generator=gen))
        c = torch.randint(0, NF, (P,), generator=gen)
        d = torch.nn.functional.normalize(torch.randn(P, 2, generator=gen), dim=-1)
        rad = torch.cat((r * torch.rand(2 * P // 3, generator=gen), r + 0.4 * torch.rand(P - 2 * P // 3, generator=gen)))
        pts = pos[c] + rad[:, None] * d
        with torch.no_grad():
            out_knn = model(pts, pos, comp, None, use_vmap=False)
            ids = torch.tensor([3, 0, 2])
            model.set_vmap_fields(ids)
            q = pos[ids][:, None, :] + 0.5 * torch.randn(3, 50, 2, generator=gen)
            out_vmap = model(q, pos[ids], comp[ids], ids, use_vmap=True)
            out_local = model(q, None, None, ids, use_vmap=True)       # points already local (models.py:344-345)
        assert int((out_knn[:, 3] == 1.0).sum()) > P // 8              # some points are outside every field
        arrays = {"p::" + k: v for k, v in model.all_fields_params.items()}
        save(f"g22_fields_2d_{enc}", points=pts, pos=pos, comp=comp, radius=np.float32(r), out_knn=out_knn, vmap_ids=ids, query=q,
             out_vmap=out_vmap, out_local=out_local, **arrays)


def g23_weighted_bins():
    """`Camera.sample_ijs_uniform(ijs, S, weights=..., boundaries=...)` (camera.py:277-289): the weighted-bin branch, with the
    two `torch.rand` draws recorded in the reference's order (bins first, offsets second).  Bin weights like a coarse pass
    produces them (a few dominant bins, many near-empty ones, exact zeros), uneven boundaries."""
    gen = torch.Generator().manual_seed(23)
    cam = camera.Camera(**NRGBD_CAMERA)
    F, R, S, B = 2, 37, 24, 15
    ijs = torch.stack((torch.randint(0, 480, (F, R), generator=gen), torch.randint(0, 640, (F, R), generator=gen)), -1)
    edges = torch.sort(torch.rand(F, R, B + 1, generator=gen) * 6.0 + 0.2, dim=-1).values
    w = torch.rand(F, R, B, generator=gen) ** 6
    w[torch.rand(F, R, B, generator=gen) < 0.2] = 0.0
    w[..., 7] += 0.05                                          # never all zero
    w = w / w.sum(-1, keepdim=True)
    torch.manual_seed(2300)
    u_bin = torch.rand(F, R, S)
    u_off = torch.rand(F, R, S)
    torch.manual_seed(2300)
    pts, dist = cam.sample_ijs_uniform(ijs, S, weights=w, boundaries=edges)
    save("g23_weighted_bins", ijs=ijs, boundaries=edges, weights=w, u_bin=u_bin, u_off=u_off, points=pts, distances=dist,
         cam=np.array([NRGBD_CAMERA[k] for k in ("width", "height", "fx", "fy", "cx", "cy")], dtype=np.float64))


def g24_render_ijs_knn():
    """`NeuralGraphMap._render_ijs` on its DEFAULT branch `use_vmap=False` (rm.py:440-451, 502-545, 586-595; the call
    `vis_blender.py:236-238` makes): arbitrary pixels, a camera that is not the training one, a `field_ids` subset, and the
    optional per-ray near / far / gt.  Three calls of the real method on one 6-field map:
      a  ijs (500,2), one c2w, camera = a preview-style camera, field_ids = [0,2,3,5], train mode (16 samples in [0, 5])
      b  ijs (3,40,2), per-ray c2ws, per-ray near / far / gt -> depth-guided second stratum + free-space / TSDF vectors
      c  eval mode (48 samples), field_ids=None, per-ray near with negative entries -> behind-camera overwrite"""
    gen = torch.Generator().manual_seed(24)
    cam_kw = dict(width=96, height=72, fx=83.1, fy=81.7, cx=47.2, cy=36.4, pixel_center=0.0)
    cam = camera.Camera(**cam_kw)
    NF = 6
    pos = torch.tensor([[0.0, 0.0, -2.5], [0.9, 0.1, -2.8], [-0.7, 0.3, -2.2], [0.3, -0.8, -3.1], [1.5, 0.9, -3.6],
                        [-0.2, 0.5, -3.3]])
    quat = rand_quats(NF, gen)
    cfg = make_config(far_distance=5.0, eval_far_distance=4.5, eval_num_samples=48, num_samples_coarse=16,
                      num_samples_depth_guided=12)
    ngm = build_map(rm, cfg, NF, pos, quat, seed=240)
    model = ngm._model
    for k, v in model.all_fields_params.items():
        if v.dim() > 1:
            v.add_(0.05 * torch.randn(v.shape, generator=gen))
    model.all_fields_params["_linears.2.weight"].mul_(3.0)
    arrays = {"p::" + k: v for k, v in model.all_fields_params.items()}
    out = dict(pos=pos, quat=quat, cam=np.array([cam_kw[k] for k in ("width", "height", "fx", "fy", "cx", "cy")]),
               train_far=np.float32(5.0), eval_far=np.float32(4.5), eval_num_samples=np.int64(48),
               num_samples_coarse=np.int64(16), num_samples_depth_guided=np.int64(12))

    def rec(tag, pred, **inputs):
        for k, v in inputs.items():
            out[f"{tag}::{k}"] = v
        for k, v in pred._asdict().items():
            if v is not None:
                out[f"{tag}::pred::{k}"] = v

    with torch.no_grad():
        # a: the vis_blender call
        N = 500
        ijs = torch.stack([torch.randint(0, cam.height, (N,), generator=gen), torch.randint(0, cam.width, (N,), generator=gen)], -1)
        c2w = look_at_c2w(torch.tensor([0.4, 0.2, 0.3]), torch.tensor([0.2, 0.0, -2.8]), gen)
        fids = torch.tensor([0, 2, 3, 5])
        torch.manual_seed(2400)
        u = torch.rand(N, 16)
        torch.manual_seed(2400)
        pred = ngm._render_ijs(ijs=ijs, c2ws=c2w, camera=cam, field_ids=fids)
        rec("a", pred, ijs=ijs, c2w=c2w, field_ids=fids, u=u)
        # b: (F,R,2) rays with per-ray poses and bounds, depth-guided samples on the kNN branch
        F, R = 3, 40
        t = synth_target(F, R, cam, pos[:F] * 0.0 + pos[[0, 1, 3]], gen, radius=1.0)
        fids_b = torch.tensor([0, 1, 3, 5])
        torch.manual_seed(2401)
        u_c = torch.rand(F, R, 16)
        u_g = torch.rand(F, R, 12)
        torch.manual_seed(2401)
        pred = ngm._render_ijs(ijs=t["ijs"], c2ws=t["c2ws"], camera=cam, field_ids=fids_b, use_vmap=False,
                               near_distances=t["near"], far_distances=t["far"], gt_distances=t["gt"].clone())
        rec("b", pred, ijs=t["ijs"], c2ws=t["c2ws"], near=t["near"], far=t["far"], gt=t["gt"], field_ids=fids_b, u_c=u_c, u_g=u_g)
        # c: eval mode, all fields, cameras inside the map with negative near distances
        ngm.eval()
        ti = synth_target(1, 300, cam, pos[:1], gen, radius=1.0, inside=True)
        ijs_c, c2ws_c, near_c, far_c = ti["ijs"][0], ti["c2ws"][0], ti["near"][0], ti["far"][0]
        assert bool((near_c < 0).any())
        torch.manual_seed(2402)
        u_e = torch.rand(300, 48)
        torch.manual_seed(2402)
        pred = ngm._render_ijs(ijs=ijs_c, c2ws=c2ws_c, camera=cam, near_distances=near_c, far_distances=far_c)
        rec("c", pred, ijs=ijs_c, c2ws=c2ws_c, near=near_c, far=far_c, u=u_e)
        ngm.train()
    save("g24_render_ijs_knn", **out, **arrays)


if __name__ == "__main__":
    import sys
    cases = [g1_directions, g2_g3_sampling, g4_field_forward, g5_quadrature, g6_train, g7_adam, g8_knn, g9_render_image,
             g10_behind_camera, g11_target_sampler, g12_skip_modes, g13_training_run, g15_triplane, g16_target_sampler_sv, g17_extract_mesh, g18_train_l2, g19_skip_other_encodings, g20_train_nll,
             g21_field_radius_override, g22_fields_2d, g23_weighted_bins, g24_render_ijs_knn]
    only = set(sys.argv[1:])            # e.g. `python make_golden.py g10_behind_camera` regenerates one group
    for fn in cases:
        if not only or fn.__name__ in only:
            fn()
