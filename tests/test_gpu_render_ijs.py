"""`NeuralGraphRenderer.render_ijs` on the reference's DEFAULT branch, `use_vmap=False` (rm.py:440-451, 502-545, 586-595;
caller vis_blender.py:236-238): arbitrary rays through the kNN-blended map, a `field_ids` subset, a camera that is not the
constructor's, optional per-ray near / far / gt.  Against fixture G24 (three calls of the real `_render_ijs`) and against
the oracle restatement (`O.render_ijs_knn`, itself pinned by G24 on the CPU) on random shapes.  All through the C ABI."""
import os
import sys

import pytest
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from conftest import load_golden, split_prefix  # noqa: E402
from gpu_common import DEV, close, make_renderer  # noqa: E402
from oracle import ngm_oracle as O  # noqa: E402
from test_oracle_golden import _g24_case  # noqa: E402
from neural_graph_mapping_amd import renderer as Rr  # noqa: E402

pytestmark = pytest.mark.gpu

FOURIER = dict(encoding="fourier", dim_enc=64, num_layers=2)
TOL = dict(rtol=5e-4, atol=5e-5)


def _dev(t):
    return None if t is None else t.to(DEV)


def _check(pred: Rr.Prediction, exp: dict, tag=""):
    got = {k: v for k, v in pred._asdict().items() if v is not None}
    assert set(got) == set(exp), (tag, sorted(got), sorted(exp))
    for k, v in exp.items():
        assert got[k].shape == v.shape, (tag, k, got[k].shape, v.shape)
        close(got[k], v, **TOL)


def _g24_renderer(g):
    ckw = dict(num_samples_coarse=int(g["num_samples_coarse"]), num_samples_depth_guided=int(g["num_samples_depth_guided"]),
               far_distance=float(g["train_far"]), eval_far_distance=float(g["eval_far"]),
               eval_num_samples=int(g["eval_num_samples"]))
    NF = g["pos"].shape[0]
    r = make_renderer(FOURIER, ckw, NF, split_prefix(g, "p::"))            # constructor camera: NRGBD 640 x 480
    r.set_field_poses(g["pos"].to(DEV), g["quat"].to(DEV))
    w, h, fx, fy, cx, cy = [float(x) for x in g["cam"]]
    return r, Rr.Camera(int(w), int(h), fx, fy, cx, cy, pixel_center=0.0)    # the camera the reference was handed


@pytest.mark.parametrize("fused", [True, False])
def test_render_ijs_knn_branch_golden(fused):
    """G24's three calls of the real `_render_ijs`, replayed with the recorded torch.rand draws"""
    g = load_golden("g24_render_ijs_knn")
    r, cam = _g24_renderer(g)
    r.eval_fused = fused
    a, ea = _g24_case(g, "a")
    # a: positional like vis_blender.py:236-238 -- use_vmap is NOT passed: the default must be the kNN branch
    pred = r.render_ijs(ijs=_dev(a["ijs"]), c2ws=_dev(a["c2w"]), camera=cam, field_ids=_dev(a["field_ids"]), u_coarse=_dev(a["u"]))
    _check(pred, ea, "a")
    assert r.last_eval_path.startswith("fused" if fused else "staged")
    b, eb = _g24_case(g, "b")
    pred = r.render_ijs(_dev(b["ijs"]), _dev(b["c2ws"]), cam, _dev(b["field_ids"]), False, _dev(b["near"]), _dev(b["far"]),
                        _dev(b["gt"]), u_coarse=_dev(b["u_c"]), u_guided=_dev(b["u_g"]))
    _check(pred, eb, "b")
    r.eval()
    c, ec = _g24_case(g, "c")
    pred = r.render_ijs(_dev(c["ijs"]), _dev(c["c2ws"]), cam, near_distances=_dev(c["near"]), far_distances=_dev(c["far"]),
                        u_coarse=_dev(c["u"]))
    _check(pred, ec, "c")
    # the camera argument is honoured: the constructor's camera gives another image
    wrong = r.render_ijs(_dev(c["ijs"]), _dev(c["c2ws"]), None, near_distances=_dev(c["near"]), far_distances=_dev(c["far"]),
                         u_coarse=_dev(c["u"]))
    assert float((wrong.rgbds.cpu() - ec["rgbds"]).abs().max()) > 1e-2
    r.train()


def _random_case(seed, lead, NF, K, mode, with_bounds, with_gt, per_ray_pose, subset, S_c=12, S_g=8):
    gen = torch.Generator().manual_seed(seed)
    grid = torch.stack(torch.meshgrid(torch.arange(3.0), torch.arange(3.0), torch.arange(2.0), indexing="ij"), -1).reshape(-1, 3)
    pos = (grid[torch.randperm(len(grid), generator=gen)[:NF]] * 0.9 + 0.05 * torch.randn(NF, 3, generator=gen)
           + torch.tensor([-0.9, -0.9, -3.2]))
    quat = torch.nn.functional.normalize(torch.randn(NF, 4, generator=gen), dim=-1)
    cam_kw = dict(width=80, height=60, fx=70.0 + seed, fy=68.0, cx=39.5 + 0.25 * seed, cy=29.5)
    ijs = torch.stack([torch.randint(0, 60, lead, generator=gen), torch.randint(0, 80, lead, generator=gen)], -1)
    c2w = torch.eye(4).repeat(*lead, 1, 1) if per_ray_pose else torch.eye(4)
    if per_ray_pose:
        c2w[..., :3, 3] = 0.3 * torch.randn(*lead, 3, generator=gen)
        ang = 0.2 * torch.randn(*lead, generator=gen)
        c2w[..., 0, 0], c2w[..., 0, 2], c2w[..., 2, 0], c2w[..., 2, 2] = ang.cos(), ang.sin(), -ang.sin(), ang.cos()
    else:
        c2w[:3, 3] = torch.tensor([0.1, -0.1, 0.2])
    near = far = gt = None
    if with_bounds:
        near = 0.8 + 0.8 * torch.rand(*lead, generator=gen)
        far = near + 2.0 + torch.rand(*lead, generator=gen)
        if seed % 2:
            near[..., ::3] -= 1.5                      # negative entries: cameras "inside", behind-camera overwrite
    if with_gt:
        gt = near + (far - near) * torch.rand(*lead, generator=gen)
        sel = torch.rand(*lead, generator=gen)
        gt = torch.where(sel < 0.15, torch.zeros_like(gt), gt)
        gt = torch.where((sel > 0.9), far + 0.2, gt)
    fids = None
    if subset:
        fids = torch.randperm(NF, generator=gen)[:max(1, NF - 2)]
    return dict(pos=pos, quat=quat, cam_kw=cam_kw, ijs=ijs, c2w=c2w, near=near, far=far, gt=gt, fids=fids,
                u_c=torch.rand(*lead, S_c, generator=gen), u_g=torch.rand(*lead, S_g, generator=gen), S_c=S_c, S_g=S_g,
                K=K, mode=mode, NF=NF)


CASES = [
    # seed, lead, NF, K, geometry mode, bounds, gt, per-ray pose, subset
    (1, (333,), 7, 2, "nrgbd", False, False, False, True),
    (2, (4, 50), 9, 3, "nrgbd", True, True, True, True),
    (3, (257,), 5, 1, "occupancy", True, False, True, False),
    (4, (2, 65), 12, 2, "density", True, True, False, True),
    (5, (3, 31), 6, 4, "occupancy", True, True, True, False),
    (6, (1,), 3, 2, "nrgbd", True, False, False, True),
    (7, (64,), 2, 2, "density", False, False, True, True),            # subset of ... NF - 2 -> max(1, 0) = 1 field, K > fields
]


@pytest.mark.parametrize("case", CASES, ids=[f"s{c[0]}" for c in CASES])
def test_render_ijs_knn_random_shapes_vs_oracle(case):
    c = _random_case(*case)
    fs = O.FieldSpec(**FOURIER)
    rs = O.RenderSpec(num_samples_coarse=c["S_c"], num_samples_depth_guided=c["S_g"], geometry_mode=c["mode"])
    params = O.init_params(fs, c["NF"], seed=case[0], sigma=3.0)
    params["_linears.2.weight"] *= 2.0
    cam_o = O.CameraSpec(**{k: c["cam_kw"][k] for k in ("width", "height", "fx", "fy", "cx", "cy")})
    exp = O.render_ijs_knn(c["ijs"], c["c2w"], cam_o, c["pos"], c["quat"], params, fs, rs, c["S_c"], near=c["near"], far=c["far"],
                           gt=c["gt"], u_coarse=c["u_c"], u_guided=c["u_g"], field_ids=c["fids"], near_const=0.25, far_const=4.5,
                           num_knn=c["K"])
    exp = {k: v for k, v in exp.items() if v is not None and k != "sample_distances"}
    r = make_renderer(FOURIER, dict(num_samples_coarse=c["S_c"], num_samples_depth_guided=c["S_g"], geometry_mode=c["mode"],
                                    near_distance=0.25, far_distance=4.5), c["NF"], params)
    r._model._num_knn = c["K"]
    r.set_field_poses(c["pos"].to(DEV), c["quat"].to(DEV))
    cam = Rr.Camera(pixel_center=0.0, **c["cam_kw"])
    pred = r.render_ijs(_dev(c["ijs"]), _dev(c["c2w"]), cam, _dev(c["fids"]), near_distances=_dev(c["near"]),
                        far_distances=_dev(c["far"]), gt_distances=_dev(c["gt"]), u_coarse=_dev(c["u_c"]),
                        u_guided=_dev(c["u_g"]) if c["gt"] is not None else None)
    _check(pred, exp, str(case))
    if c["gt"] is None:
        # the one-call path and the staged entry points agree bit for bit
        assert r.last_eval_path.startswith("fused")
        r.eval_fused = False
        again = r.render_ijs(_dev(c["ijs"]), _dev(c["c2w"]), cam, _dev(c["fids"]), near_distances=_dev(c["near"]),
                             far_distances=_dev(c["far"]), u_coarse=_dev(c["u_c"]))
        assert r.last_eval_path == "staged"
        for k in ("rgbds", "color_vars", "depth_vars", "term_probs"):
            assert torch.equal(getattr(pred, k), getattr(again, k)), k


def test_render_ijs_knn_branch_semantics():
    """What the reference's branch does and does not accept: errors, empty input, in-kernel draws, block boundaries."""
    c = _random_case(11, (300,), 6, 2, "nrgbd", True, True, False, True)
    fs = O.FieldSpec(**FOURIER)
    params = O.init_params(fs, 6, seed=3, sigma=3.0)
    r = make_renderer(FOURIER, dict(num_samples_coarse=12, num_samples_depth_guided=8, block_size=20 * 77), 6, params)
    r.set_field_poses(c["pos"].to(DEV), c["quat"].to(DEV))
    ijs, c2w = _dev(c["ijs"]), _dev(c["c2w"])
    with pytest.raises(ValueError, match="field_ids=None only supported for use_vmap=False"):        # rm.py:497-498
        r.render_ijs(ijs[None], c2w, None, None, True)
    with pytest.raises(TypeError):                                        # rm.py:522-526: gt compared with near=None
        r.render_ijs(ijs, c2w, gt_distances=_dev(c["gt"]))
    # blocks of the staged path (block_size = 77 rays of 20 samples) do not change the result
    kw = dict(near_distances=_dev(c["near"]), far_distances=_dev(c["far"]), gt_distances=_dev(c["gt"]), u_coarse=_dev(c["u_c"]),
              u_guided=_dev(c["u_g"]))
    small = r.render_ijs(ijs, c2w, **kw)
    r._config["block_size"] = 3000000
    whole = r.render_ijs(ijs, c2w, **kw)
    for a, b in zip(small, whole):
        assert torch.equal(a, b)
    assert small.freespace_geometry.numel() > 0 and small.tsdf_residuals.numel() > 0
    # no rays
    e = r.render_ijs(ijs[:0], c2w)
    assert e.rgbds.shape == (0, 4) and e.term_probs.shape == (0,) and e.freespace_geometry is None
    # in-kernel Philox draws: finite, reproducible per seed, different across seeds
    p1 = r.render_ijs(ijs, c2w, seed=5)
    p2 = r.render_ijs(ijs, c2w, seed=5)
    p3 = r.render_ijs(ijs, c2w, seed=6)
    assert torch.isfinite(p1.rgbds).all() and torch.equal(p1.rgbds, p2.rgbds) and not torch.equal(p1.rgbds, p3.rgbds)
    # neus has no kNN branch in the reference (neus_isds=None reaches `None * float`, rm.py:641-647, 753-754)
    rn = make_renderer(FOURIER, dict(num_samples_coarse=12, num_samples_depth_guided=8, geometry_mode="neus"), 6, params)
    rn.set_field_poses(c["pos"].to(DEV), c["quat"].to(DEV))
    with pytest.raises(TypeError, match="neus"):
        rn.render_ijs(ijs, c2w)
