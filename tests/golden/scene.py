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


# ---- the long training runs of fixture G13 (make_golden.g13_training_run <-> tests/test_gpu_training_run.py) --------
A_ITERS, A_CHECKPOINTS, A_BATCH_SEED, A_U_SEED = 100, (10, 30, 100), 20000, 30000
B_FIELDS, B_ITERS, B_BATCH_SEED, B_U_SEED = 96, 1200, 40000, 50000
B_EVAL_FROM, B_EVAL_EVERY = 400, 20
B_HELD_OUT_RAYS, B_HELD_OUT_SEED, B_HELD_OUT_U_SEED = 512, 99, 77


def perturbed_init(proto: torch.Tensor, F: int, name: str, seed: int) -> torch.Tensor:
    """F copies of the prototype tensor (models.py:254-257 clones it for every new field) + 0.05 N(0,1) on weights and
    biases so that the fields differ; seeded per tensor name, regenerated identically by the tests."""
    import zlib
    out = proto.unsqueeze(0).repeat(F, *([1] * proto.dim())).clone()
    if proto.dim() >= 1:
        g = torch.Generator().manual_seed(seed + zlib.crc32(name.encode()) % 100000)
        out += 0.05 * torch.randn(out.shape, generator=g)
    return out


def checksum(params: dict) -> float:
    return float(sum(v.double().abs().sum() for v in params.values()))


def held_out_scores(rgbds: torch.Tensor, th: dict):
    """per field: PSNR of the colours (evaluation.py:46-56: clamp to [0,1], data range 1) and mean |depth error| over
    the held-out rays that hit the sphere"""
    m = th["depth_mask"].to(rgbds.device)
    tgt = th["rgbds"].to(rgbds.device)
    n = m.sum(-1).clamp_min(1)
    mse = (((rgbds[..., :3].clamp(0, 1) - tgt[..., :3]) ** 2).mean(-1) * m).sum(-1) / n
    derr = ((rgbds[..., 3] - tgt[..., 3]).abs() * m).sum(-1) / n
    return (10.0 * torch.log10(1.0 / mse)).cpu(), derr.cpu()


# ---- one RGB-D frame for the single-view target sampler (fixture G16): regenerated from its seed, not stored ----
SV_H, SV_W = 200, 320
SV_FX = SV_FY = 260.0
SV_CX, SV_CY = 159.5, 99.5


def sv_frame(seed: int) -> torch.Tensor:
    """(H, W, 4) RGB-D: a smooth depth field between 1.5 and 4 m with a band of missing depth, colours from the pixel
    position.  Only exactly reproducible torch-CPU arithmetic (no transcendental of a device)."""
    g = torch.Generator().manual_seed(seed)
    i = torch.arange(SV_H, dtype=torch.float32)[:, None].expand(SV_H, SV_W)
    j = torch.arange(SV_W, dtype=torch.float32)[None, :].expand(SV_H, SV_W)
    a = torch.rand(4, generator=g)
    u, v = i / SV_H, j / SV_W
    depth = 1.5 + 2.5 * ((a[0] * u + a[1] * v + a[2] * u * v + 0.25 * a[3] * (u - v) ** 2) % 1.0)
    depth[40:48, 30:200] = 0.0                                            # missing depth
    rgb = torch.stack((u, v, 0.5 * (u + v)), -1)
    return torch.cat((rgb, depth[..., None]), -1).contiguous()
