"""Synthetic, LEARNABLE supervision shared by the fixture generator (make_golden.py, runs the real reference) and the
tests that replay the same batches: a sphere of radius `radius` around every field centre, colour = smooth function of
the hit point in the field frame.  Pure torch on the CPU, seeded: both sides regenerate identical batches from
(F, R, pos, seed), so a long training run needs no per-iteration data in the fixture.  Test infrastructure only."""
import torch

FX = FY = 554.2562584220408          # NRGBD intrinsics (config/nrgbd_dataset.yaml:18-25), pixel centre 0
CX, CY, W, H = 319.5, 239.5, 640, 480


def pixel_dirs(ijs):
    d = torch.stack([(ijs[..., 1] - CX) / FX, -(ijs[..., 0] - CY) / FY, -torch.ones(ijs.shape[:-1])], -1)
    return torch.nn.functional.normalize(d, dim=-1)


def sphere_scene_batch(F, R, pos, seed, radius=0.6, phase=None):
    """Rays from cameras 2-3 m away looking at the field centres (as bench.py's synth_target); gt = distance to the
    sphere (0.0 = miss -> no depth / colour supervision, termination target 0)."""
    g = torch.Generator().manual_seed(seed)
    ijs = torch.stack([torch.randint(0, H, (F, R), generator=g), torch.randint(0, W, (F, R), generator=g)], -1)
    eye = pos[:, None] + torch.nn.functional.normalize(torch.randn(F, R, 3, generator=g), dim=-1) * (
        2 + torch.rand(F, R, 1, generator=g))
    fwd = torch.nn.functional.normalize(pos[:, None] + 0.3 * torch.randn(F, R, 3, generator=g) - eye, dim=-1)
    right = torch.nn.functional.normalize(torch.linalg.cross(fwd, torch.tensor([0.0, 1.0, 0.0]).expand_as(fwd)), dim=-1)
    c2w = torch.eye(4).repeat(F, R, 1, 1)
    c2w[..., :3, 0], c2w[..., :3, 1], c2w[..., :3, 2], c2w[..., :3, 3] = right, torch.linalg.cross(right, fwd), -fwd, eye
    d = pixel_dirs(ijs)
    dw = torch.einsum("...ij,...j->...i", c2w[..., :3, :3], d)
    oc = eye - pos[:, None]
    b = (oc * dw).sum(-1)
    disc = b * b - ((oc * oc).sum(-1) - radius * radius)
    hit_ok = disc > 0
    gt = torch.where(hit_ok, -b - disc.clamp_min(0).sqrt(), torch.zeros_like(b))
    near, far = (-b - 1).clamp_min(0), (-b + 1).clamp_min(0)
    hit = oc + gt[..., None] * dw                                       # hit point relative to the field centre
    ph = torch.tensor([0.0, 1.0, 2.0]) if phase is None else phase[:, None]
    rgb = torch.where(hit_ok[..., None], 0.5 + 0.4 * torch.sin(3.0 * hit + ph), torch.zeros_like(hit))
    return dict(ijs=ijs, c2ws=c2w, near=near, far=far, gt=gt, rgbds=torch.cat([rgb, (gt * d[..., 2].abs())[..., None]], -1),
                depth_mask=hit_ok, term_probs=hit_ok.float(), term_mask=torch.ones(F, R, dtype=torch.bool))
