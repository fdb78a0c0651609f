"""End-to-end smoke of the drop-in path on one MI355X: keyframe store -> training-target sampler -> fused
training step (forward, losses, backward, sparse Adam) -> kNN-blended render -> PSNR.

    python examples/fit_synthetic.py [--iters 300]

A synthetic RGB-D "scan" of a textured wall with a sphere in front of it is observed from a few keyframes; fields
on a grid in front of the cameras are trained exactly as NeuralGraphMap._optimization_iteration would
(rm.py:1123-1221), with every tensor operation of the hot path running in the HIP kernels.
"""
import argparse
import math
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from neural_graph_mapping_amd import models as M  # noqa: E402
from neural_graph_mapping_amd import renderer as Rr  # noqa: E402

W, H, FOC = 160, 120, 140.0


def look_at(eye, target):
    f = torch.nn.functional.normalize(target - eye, dim=-1)
    up = torch.tensor([0.0, 1.0, 0.0])
    s = torch.nn.functional.normalize(torch.linalg.cross(f, up), dim=-1)
    u = torch.linalg.cross(s, f)
    T = torch.eye(4)
    T[:3, 0], T[:3, 1], T[:3, 2], T[:3, 3] = s, u, -f, eye          # OpenGL camera: x right, y up, z back
    return T


def synth_keyframe(c2w):
    """Ray-cast an analytic scene: wall z = -3 with a checker texture, unit-radius sphere at (0, 0, -2.2)."""
    ii, jj = torch.meshgrid(torch.arange(H, dtype=torch.float32), torch.arange(W, dtype=torch.float32), indexing="ij")
    d = torch.stack([(jj - (W - 1) / 2) / FOC, -(ii - (H - 1) / 2) / FOC, -torch.ones_like(ii)], -1)
    d = torch.nn.functional.normalize(d, dim=-1)
    dw = d @ c2w[:3, :3].T
    o = c2w[:3, 3]
    t_wall = (-3.0 - o[2]) / dw[..., 2]
    c = torch.tensor([0.0, 0.0, -2.2])
    oc = o - c
    b = (dw * oc).sum(-1)
    disc = b * b - (oc @ oc - 0.6 ** 2)
    t_sph = torch.where(disc > 0, -b - torch.sqrt(disc.clamp_min(0)), torch.full_like(b, float("inf")))
    t = torch.minimum(torch.where(t_wall > 0, t_wall, torch.full_like(t_wall, float("inf"))),
                      torch.where(t_sph > 0, t_sph, torch.full_like(t_sph, float("inf"))))
    p = o + dw * t[..., None]
    wall = ((torch.floor(p[..., 0] * 2) + torch.floor(p[..., 1] * 2)) % 2)[..., None] * torch.tensor([0.6, 0.5, 0.2]) + 0.2
    sph = torch.tensor([0.8, 0.2, 0.2]).expand_as(wall) * (0.5 + 0.5 * torch.nn.functional.normalize(p - c, dim=-1)[..., 1:2])
    rgb = torch.where((t_sph < t_wall)[..., None], sph, wall)
    depth = t * d[..., 2].abs()                                     # depth along the optical axis
    return torch.cat([rgb, depth[..., None]], -1)


def main(iters=300, device="cuda:0", quiet=False):
    torch.manual_seed(0)
    dev = torch.device(device)
    cam = Rr.Camera(W, H, FOC, FOC, (W - 1) / 2, (H - 1) / 2, pixel_center=0.0)
    radius = 0.5
    # fields on a grid covering the visible surfaces
    gx, gy, gz = torch.meshgrid(torch.arange(-2.0, 2.01, 0.5), torch.arange(-1.5, 1.51, 0.5), torch.tensor([-2.9]), indexing="ij")
    sx, sy, sz = torch.meshgrid(torch.arange(-0.5, 0.51, 0.5), torch.arange(-0.5, 0.51, 0.5), torch.tensor([-2.4, -1.9]),
                                indexing="ij")
    pos = torch.cat([torch.stack([gx, gy, gz], -1).reshape(-1, 3), torch.stack([sx, sy, sz], -1).reshape(-1, 3)])
    NF = pos.shape[0]
    quat = torch.zeros(NF, 4)
    quat[:, 0] = 1
    model = M.NeuralFieldSet(dim_points=3, field_type="neural_graph_mapping.models.NeuralField", field_kwargs=dict(
        encoding_type="neural_graph_mapping.positional_encodings.PositionalEncodingFourier",
        encoding_kwargs=dict(dim_in=3, dim_out=64, mu=0.0, sigma=4.0, raw_coords=True), num_layers=2, dim_out=4),
        num_knn=2, distance_factor=10.0, outside_value=1.0, field_radius=radius, scale_mode="unit_cube").to(dev)
    cfg = dict(geometry_mode="nrgbd", geometry_factor=20.0, color_factor=1.0, truncation_distance=0.1, field_radius=radius,
               termination_weight=0.0, photometric_weight=1.0, photometric_loss="l1", depth_weight=1.0, depth_loss="huber",
               freespace_weight=40.0, tsdf_weight=50.0,
               learning_rate=2e-3, adam_eps=1e-15, adam_weight_decay=1e-5, num_samples_coarse=8, num_samples_depth_guided=16,
               near_distance=0.0, far_distance=5.0, eval_near_distance=0.5, eval_far_distance=4.5, eval_num_samples=160)
    r = Rr.NeuralGraphRenderer(model, cam, cfg, device=dev)
    r.add_fields(NF)
    r.set_field_poses(pos.to(dev), quat.to(dev))
    eyes = torch.tensor([[0.0, 0.0, 0.0], [0.7, 0.2, 0.1], [-0.7, -0.1, 0.2], [0.2, 0.5, 0.3]])
    c2ws = torch.stack([look_at(e, torch.tensor([0.0, 0.0, -2.6])) for e in eyes])
    store = torch.stack([synth_keyframe(T) for T in c2ws]).contiguous().to(dev)
    c2ws = c2ws.to(dev)
    cid = torch.arange(len(eyes), device=dev)
    cur = torch.arange(NF, device=dev)

    losses = []
    for it in range(iters):
        tgt = r.sample_target_mv(cur, c2ws, store, cid, num_train_fields=32, num_rays_per_field=256, camera=cam)
        out = r.optimization_iteration(tgt, seed=it)
        if it % 25 == 0 or it == iters - 1:
            losses.append(float(out["combined"]))
            if not quiet:
                print(f"iter {it:4d}  loss {losses[-1]:.4f}  (photo {float(out['photometric_l1']):.4f} depth {float(out['depth_huber']):.4f} "
                      f"fs {float(out['freespace']):.4f} tsdf {float(out['tsdf']):.4f})")
    r.eval()                                                        # as rm.py:1978 before an evaluation render
    rgbd, _ = r.render_image(c2ws[0], camera=cam)
    r.train()
    ref = store[0]
    valid = torch.isfinite(ref[..., 3]) & (rgbd[..., 3] > 0.1)      # pixels whose ray meets a trained field
    mse = ((rgbd[..., :3].clamp(0, 1) - ref[..., :3]) ** 2)[valid].mean()
    psnr = float(10 * torch.log10(1.0 / mse))
    derr = float((rgbd[..., 3] - ref[..., 3]).abs()[valid].median())
    if not quiet:
        print(f"keyframe 0 re-rendered: PSNR {psnr:.2f} dB, median |depth error| {derr:.3f} m over "
              f"{100 * float(valid.float().mean()):.0f} % of the pixels ({NF} fields)")
    return losses, psnr, derr


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=300)
    a = ap.parse_args()
    main(a.iters)
