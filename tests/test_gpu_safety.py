"""Memory-safety / race pass of round 6 (SURVEY 5 "race detection / sanitizers").

GPU AddressSanitizer and xnack are not available on this pool, so the pass is built from what can run here:

(a) GUARD BANDS: every buffer the host layer hands to the C ABI -- outputs, gradients, loss vectors, and above all the
    caller-owned workspaces whose layout the library plans itself -- is allocated with 4 KiB of pattern on either side
    (torch.empty / zeros / empty_like / zeros_like are intercepted for device tensors while a scenario runs); the scenarios
    are the ragged shapes of the parity tests (rays of 1..130 samples against 16 / 32 / 64-sample tiles, fields that start
    in the middle of a tile, partial last blocks) over every entry point; after a device synchronisation every band must
    still hold its pattern.  A kernel that writes one element before or after its buffer fails here.
(b) TWO STREAMS: two renderers with DIFFERENT configurations (network, stash mode, sample counts) of one process, driven
    concurrently on two HIP streams, must produce bitwise what they produce alone -- exercises the process-wide
    bookkeeping of the library (the seed-provenance map keyed by workspace, the profile hooks, the last-launch records)."""
import os
import sys
from contextlib import contextmanager

import pytest
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from conftest import load_golden, split_prefix  # noqa: E402
from gpu_common import DEV, NRGBD, make_renderer, make_target, ragged_case, synth_target  # noqa: E402
from neural_graph_mapping_amd import _capi as K  # noqa: E402
from neural_graph_mapping_amd import ops  # noqa: E402
from neural_graph_mapping_amd import renderer as Rr  # noqa: E402
from oracle import ngm_oracle as O  # noqa: E402

pytestmark = pytest.mark.gpu

FOURIER = dict(encoding="fourier", dim_enc=64, num_layers=2)
HASH = dict(encoding="permuto", num_layers=1)
GUARD = 4096
PATTERN = 0xA5


class Bands:
    def __init__(self):
        self.raw = []            # (raw uint8 tensor, payload bytes)
        self.bytes = 0

    def alloc(self, shape, dtype, device, zero):
        n = 1
        for d in shape:
            n *= int(d)
        nbytes = n * torch.empty((), dtype=dtype).element_size()
        pad = (-nbytes) % 256                       # the payload keeps the allocator's 256-byte alignment on both ends
        raw = _REAL["empty"](GUARD + nbytes + pad + GUARD, dtype=torch.uint8, device=device)
        raw.fill_(PATTERN)
        self.raw.append((raw, nbytes))
        self.bytes += nbytes
        out = raw[GUARD:GUARD + nbytes].view(dtype).view(*shape) if n else _REAL["empty"](*shape, dtype=dtype, device=device)
        if zero and n:
            out.zero_()
        return out

    def check(self):
        torch.cuda.synchronize()
        bad = []
        for i, (raw, nbytes) in enumerate(self.raw):
            lo, hi = raw[:GUARD], raw[GUARD + nbytes:]
            if not bool((lo == PATTERN).all()):
                bad.append((i, nbytes, "before", int((lo != PATTERN).nonzero()[-1]) - GUARD))
            if not bool((hi == PATTERN).all()):
                bad.append((i, nbytes, "after", int((hi != PATTERN).nonzero()[0])))
        assert not bad, f"writes outside caller-owned buffers (index, payload bytes, side, byte offset): {bad[:8]}"
        return len(self.raw)


_REAL = {}


def _is_dev(device):
    return device is not None and torch.device(device).type == "cuda"


@contextmanager
def guard_bands():
    """intercept the host layer's device allocations (it allocates with exactly these four torch calls + new_empty)"""
    bands = Bands()
    for name in ("empty", "zeros", "empty_like", "zeros_like"):
        _REAL[name] = getattr(torch, name)
    real_new_empty = torch.Tensor.new_empty

    def shape_of(args):
        return tuple(args[0]) if len(args) == 1 and isinstance(args[0], (tuple, list, torch.Size)) else tuple(args)

    def empty(*args, dtype=None, device=None, **kw):
        if not _is_dev(device) or kw:
            return _REAL["empty"](*args, dtype=dtype, device=device, **kw)
        return bands.alloc(shape_of(args), dtype or torch.get_default_dtype(), device, False)

    def zeros(*args, dtype=None, device=None, **kw):
        if not _is_dev(device) or kw:
            return _REAL["zeros"](*args, dtype=dtype, device=device, **kw)
        return bands.alloc(shape_of(args), dtype or torch.get_default_dtype(), device, True)

    def empty_like(t, dtype=None, **kw):
        if not t.is_cuda or kw:
            return _REAL["empty_like"](t, dtype=dtype, **kw)
        return bands.alloc(tuple(t.shape), dtype or t.dtype, t.device, False)

    def zeros_like(t, dtype=None, **kw):
        if not t.is_cuda or kw:
            return _REAL["zeros_like"](t, dtype=dtype, **kw)
        return bands.alloc(tuple(t.shape), dtype or t.dtype, t.device, True)

    def new_empty(self, *args, dtype=None, device=None, **kw):
        if not self.is_cuda or kw or (device is not None and not _is_dev(device)):
            return real_new_empty(self, *args, dtype=dtype, device=device, **kw)
        return bands.alloc(shape_of(args), dtype or self.dtype, self.device, False)

    torch.empty, torch.zeros, torch.empty_like, torch.zeros_like = empty, zeros, empty_like, zeros_like
    torch.Tensor.new_empty = new_empty
    try:
        yield bands
    finally:
        torch.empty, torch.zeros, torch.empty_like, torch.zeros_like = (_REAL[k] for k in ("empty", "zeros", "empty_like", "zeros_like"))
        torch.Tensor.new_empty = real_new_empty


def _perturb(r, seed=0):
    g = torch.Generator(device=DEV).manual_seed(seed)
    with torch.no_grad():
        for k, v in r._model.all_fields_params.items():
            if v.dim() > 1 and k != "_encoding.random_shift_per_level":
                v.add_(0.05 * torch.randn(v.shape, device=DEV, generator=g))
    r._model.refresh_lp()


# ------------------------------------------------------------------------------------------------ (a) guard bands
TRAIN_SHAPES = [(1, 256, 16, 16), (3, 41, 9, 5), (2, 96, 8, 16), (1, 7, 64, 64), (4, 130, 2, 5), (5, 1, 1, 1), (2, 33, 3, 0)]


@pytest.mark.parametrize("net", ["fourier", "fourier_half", "fourier_f32", "hash", "nerf", "triplane", "skip_concat"])
def test_guard_bands_training_step(net):
    """fused forward + backward + Adam (every MLP backward kernel family), ragged shapes, all allocations banded"""
    fkw, ckw = {
        "fourier": (FOURIER, {}), "fourier_half": (FOURIER, dict(activation_stash="half")),
        "fourier_f32": (FOURIER, dict(mlp_matmul="f32")), "hash": (HASH, {}),
        "nerf": (dict(encoding="nerf", num_octaves=8, num_layers=1), {}),
        "triplane": (dict(encoding="triplane", num_layers=1, resolution=16, num_components=32), {}),
        "skip_concat": ({**FOURIER, "skip_mode": "concat"}, {}),
    }[net]
    total = 0
    for F, R, n_c, n_g in TRAIN_SHAPES:
        with guard_bands() as bands:
            r = make_renderer(fkw, dict(num_samples_coarse=n_c, num_samples_depth_guided=n_g, termination_weight=0.3, **ckw), F)
            _perturb(r, seed=R)
            pos, quat, t = synth_target(F, R, seed=R)
            r.set_field_poses(pos.to(DEV), quat.to(DEV))
            tgt = make_target(t, torch.arange(F))
            r.optimization_iteration(tgt, seed=3, update=False)
            r.optimization_iteration(tgt, seed=4, update=True)
            r.optimization_iteration(tgt, seed=5, update=True)               # cached workspace, second Adam step
            total += bands.check()
    assert total > 12 * len(TRAIN_SHAPES)                # every scenario banded at least its outputs, gradients and workspace


@pytest.mark.parametrize("geo", ["nrgbd", "occupancy", "density", "neus"])
def test_guard_bands_render_ijs_autograd_and_stages(geo):
    """render_ijs(use_vmap=True) under autograd (fused or staged per geometry mode), the standalone stages and their backward"""
    F, R, n_c, n_g = 3, 37, 7, 5
    with guard_bands() as bands:
        r = make_renderer(FOURIER, dict(num_samples_coarse=n_c, num_samples_depth_guided=n_g, geometry_mode=geo,
                                        geometry_factor=5.0 if geo == "neus" else 20.0), F)
        _perturb(r)
        pos, quat, t = synth_target(F, R, seed=2)
        r.set_field_poses(pos.to(DEV), quat.to(DEV))
        tgt = make_target(t, torch.arange(F))
        ids = torch.arange(F, device=DEV)
        pred = r.render_ijs(tgt.ijs, tgt.c2ws, None, ids, True, tgt.near_distances, tgt.far_distances, tgt.gt_distances, seed=7)
        loss = r.compute_losses(tgt, pred)
        loss["combined"].backward()
        # stages: sampler (+ weighted bins), per-field evaluation fwd / bwd, encode fwd / bwd, quadrature fwd / bwd
        rc = r._rc_train
        pc, pw, dist = ops.sample_rays_world(rc, tgt.ijs, tgt.c2ws, tgt.near_distances, tgt.far_distances, tgt.gt_distances, seed=1)
        B = 9
        edges = torch.sort(torch.rand(F, R, B + 1, device=DEV) * 4 + 0.3, -1).values
        w = torch.rand(F, R, B, device=DEV)
        ops.sample_rays_weighted(rc, tgt.ijs, edges, w / w.sum(-1, keepdim=True), seed=2)
        params = {k: v.detach().clone().requires_grad_() for k, v in r._model.all_fields_params.items() if k != "_neus_sd"}
        out = ops.field_eval(r._fc, params, pw.reshape(F, -1, 3), pos.to(DEV), quat.to(DEV))
        out.sum().backward()
        enc = ops.encode(r._fc, params, pw.reshape(F, -1, 3), pos.to(DEV), quat.to(DEV))
        ops.encode_bwd(r._fc, params, pw.reshape(F, -1, 3), torch.ones_like(enc), pos.to(DEV), quat.to(DEV))
        S = dist.shape[-1]
        cols = torch.rand(F, R, S, 3, device=DEV, requires_grad=True)
        geoms = torch.randn(F, R, S, device=DEV, requires_grad=True)
        isd = torch.full((F, 1, 1), 1.3, device=DEV) if geo == "neus" else None
        q = ops.quadrature(rc, cols, geoms, dist, -pc[..., 2].contiguous(), isd)
        (q[0].sum() + q[1].sum() + q[4].sum() + q[2].sum() + q[3].sum()).backward()
        n = bands.check()
    assert n > 30


@pytest.mark.parametrize("net,K_,S", [("fourier", 2, 48), ("fourier", 5, 33), ("hash", 2, 640), ("fourier_f32", 3, 17)])
def test_guard_bands_knn_paths(net, K_, S):
    """render_ijs(use_vmap=False) fused + staged (+ gt: free-space / TSDF vectors), render_pixels on unaligned ranges, evaluate_points"""
    gen = torch.Generator().manual_seed(S)
    g = torch.arange(-1.0, 1.01, 1.0)
    pos = torch.stack(torch.meshgrid(g, g, torch.tensor([-2.0, -1.4]), indexing="ij"), -1).reshape(-1, 3)
    pos = pos + 0.01 * torch.randn(pos.shape, generator=gen)
    NF = pos.shape[0]
    quat = torch.nn.functional.normalize(torch.randn(NF, 4, generator=gen), dim=-1)
    with guard_bands() as bands:
        ckw = dict(num_samples_coarse=11, num_samples_depth_guided=6, eval_num_samples=S, eval_far_distance=4.0, pixel_block_size=777,
                   eval_ray_block=777, block_size=5000)
        if net == "fourier_f32":
            ckw["mlp_matmul"] = "f32"
        r = make_renderer(HASH if net == "hash" else FOURIER, ckw, NF)
        r._model._num_knn = K_
        _perturb(r)
        r.set_field_poses(pos.to(DEV), quat.to(DEV))
        N = 1234
        ijs = torch.stack([torch.randint(0, 480, (N,), generator=gen), torch.randint(0, 640, (N,), generator=gen)], -1).to(DEV)
        c2w = torch.eye(4, device=DEV)
        near = (0.5 + torch.rand(N, generator=gen)).to(DEV)
        far = near + 2.5
        gt = near + 1.0
        sub = torch.tensor([0, 2, 5, 7, 11, 13], device=DEV)
        for fused in (True, False):
            r.eval_fused = fused
            r.train()
            r.render_ijs(ijs, c2w, field_ids=sub, near_distances=near, far_distances=far, seed=1)
            r.render_ijs(ijs.view(2, N // 2, 2), c2w, None, None, False, near.view(2, -1), far.view(2, -1), gt.view(2, -1), seed=2)
            r.eval()
            r.render_ijs(ijs, c2w.expand(N, 4, 4), seed=3)
            r.render_pixels(c2w, 640 * 100 + 13, 640 * 100 + 13 + 2001, seed=4)
        r.evaluate_points(torch.randn(3001, 3, device=DEV) + torch.tensor([0.0, 0.0, -1.7], device=DEV), block_size=1000)
        n = bands.check()
    assert n > 20


def test_guard_bands_mesh_target_sampler_and_adam():
    """dense-grid evaluation + marching cubes (classify / scan / emit) on the map of G17, the training-target sampler
    (multi-view + single-view geometry kernels), standalone sparse Adam"""
    g = load_golden("g17_extract_mesh")
    with guard_bands() as bands:
        NF = g["pos"].shape[0]
        r = make_renderer(FOURIER, dict(field_radius=float(g["field_radius"]), num_samples_coarse=8, num_samples_depth_guided=16), NF,
                          split_prefix(g, "p::"))
        r.set_field_poses(g["pos"].to(DEV), g["quat"].to(DEV))
        v, f, c = r.extract_mesh(None, resolution=float(g["resolution"]))
        v2, _, _ = r.extract_mesh(None, resolution=float(g["resolution"]), block=16)          # many small blocks
        assert v.shape[0] > 100 and f.shape[0] > 100 and v2.shape[0] > 100
        n = bands.check()
    assert n > 5
    g = load_golden("g11_target_sampler")
    with guard_bands() as bands:
        r = make_renderer(dict(encoding="fourier", dim_enc=32, num_layers=1), dict(num_samples_coarse=4, num_samples_depth_guided=4),
                          int(g["num_fields"]))
        r.set_field_poses(g["positions"].to(DEV), torch.zeros(int(g["num_fields"]), 4, device=DEV))
        cam = Rr.Camera(int(g["width"]), int(g["height"]), float(g["fx"]), float(g["fy"]), float(g["cx"]), float(g["cy"]), pixel_center=0.0)
        draws = dict(subset_observed=g["d_subset_observed"], subset_random=g["d_subset_random"], offsets=g["d_offsets"],
                     frame_cids=g["d_frame_cids"], u_xy=g["d_u_xy"])
        t = r.sample_target_mv(g["current_field_ids"], g["c_c2w"].to(DEV), g["nc_rgbd"].to(DEV).contiguous(),
                               g["frame_cid_to_ncid"].to(DEV), int(g["num_train_fields"]), int(g["num_rays_per_field"]), camera=cam,
                               draws=draws)
        assert torch.equal(t.ijs.cpu(), g["o_ijs"].long())
        n = bands.check()
    assert n > 5
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
    import scene
    g = load_golden("g16_target_sampler_sv")
    with guard_bands() as bands:
        NF = g["positions"].shape[0]
        r = make_renderer(dict(encoding="fourier", dim_enc=32, num_layers=1),
                          dict(num_samples_coarse=4, num_samples_depth_guided=4, field_radius=float(g["field_radius"])), NF)
        r.set_field_poses(g["positions"].to(DEV), torch.zeros(NF, 4, device=DEV))
        cam = Rr.Camera(scene.SV_W, scene.SV_H, scene.SV_FX, scene.SV_FY, scene.SV_CX, scene.SV_CY, pixel_center=0.0)
        draws = dict(subset_points=g["d_subset_points"].long(), segments=g["d_segments"],
                     subset_fields=g["d_subset_fields"] if "d_subset_fields" in g else None)
        t = r.sample_target_sv(scene.sv_frame(int(g["frame_seed"])), g["c2w"], g["active_field_ids"], int(g["num_train_fields"]),
                               int(g["num_rays_per_field"]), camera=cam, draws=draws)
        assert torch.equal(t.ijs.cpu(), g["o_ijs"].long())
        n = bands.check()
    assert n > 5
    with guard_bands() as bands:
        r = make_renderer(FOURIER, dict(num_samples_coarse=4, num_samples_depth_guided=4), 7)
        ids = torch.tensor([5, 0, 3], device=DEV)
        for k, v in r._model.all_fields_params.items():
            st = r._optim_state[k]
            ops.adam_sparse_(v, st["exp_avg"], st["exp_avg_sq"], torch.randn(3, *v.shape[1:], device=DEV), ids, 3)
        n = bands.check()
    assert n > 5


# ------------------------------------------------------------------------------------------------ (b) two streams
def _snapshot(res):
    out = {k: v.detach().clone() for k, v in res["grads"].items()}
    out["__combined"] = res["combined"].detach().clone()
    out["__rgbds"] = res["prediction"].rgbds.detach().clone()
    return out


def test_two_renderers_on_two_streams_equal_serial_results():
    cases = [
        (FOURIER, dict(num_samples_coarse=16, num_samples_depth_guided=16, activation_stash="half"), 4, 150, 11),
        (HASH, dict(num_samples_coarse=8, num_samples_depth_guided=16), 6, 96, 12),
        (FOURIER, dict(num_samples_coarse=9, num_samples_depth_guided=5, mlp_matmul="f32", depth_loss="laplacian_nll"), 3, 200, 13),
    ]
    rs, tgts = [], []
    for fkw, ckw, F, R, seed in cases:
        r = make_renderer(fkw, {**ckw, "termination_weight": 0.3}, F)
        _perturb(r, seed)
        pos, quat, t = synth_target(F, R, seed=seed)
        r.set_field_poses(pos.to(DEV), quat.to(DEV))
        rs.append(r)
        tgts.append(make_target(t, torch.arange(F)))
    serial = []
    for r, tgt in zip(rs, tgts):
        serial.append(_snapshot(r.optimization_iteration(tgt, seed=21, update=False)))
        torch.cuda.synchronize()
    streams = [torch.cuda.Stream() for _ in rs]
    for rep in range(12):
        got = [None] * len(rs)
        order = list(range(len(rs))) if rep % 2 == 0 else list(reversed(range(len(rs))))
        for s in streams:
            s.wait_stream(torch.cuda.current_stream())
        for i in order:                                          # launches are asynchronous: the three iterations overlap on the GPU
            with torch.cuda.stream(streams[i]):
                got[i] = rs[i].optimization_iteration(tgts[i], seed=21, update=False)
        for i in order:
            streams[i].synchronize()
            snap = _snapshot(got[i])
            for k, v in serial[i].items():
                assert torch.equal(torch.isnan(v), torch.isnan(snap[k])) and torch.equal(v.nan_to_num(), snap[k].nan_to_num()), (rep, i, k)
    # the evaluation path on two streams: two images of two maps
    g = load_golden("g9_render_image")
    imgs = []
    rr = []
    for mm in ("auto", "f32"):
        r = make_renderer(FOURIER, dict(num_samples_coarse=8, num_samples_depth_guided=16, eval_far_distance=float(g["eval_far"]),
                                        eval_num_samples=int(g["eval_num_samples"]), mlp_matmul=mm), 3, split_prefix(g, "p::"))
        r.set_field_poses(g["pos"].to(DEV), g["quat"].to(DEV))
        r.eval()
        rr.append(r)
    w, h, fx, fy, cx, cy = [float(x) for x in g["cam"]]
    cam = Rr.Camera(int(w), int(h), fx, fy, cx, cy, pixel_center=0.0)
    c2w, u = g["c2w"].to(DEV), g["u"].to(DEV)
    for r in rr:
        imgs.append(r.render_image(c2w, cam, u=u)[0].clone())
    torch.cuda.synchronize()
    for rep in range(6):
        outs = []
        for s in streams[:2]:
            s.wait_stream(torch.cuda.current_stream())
        for r, s in zip(rr, streams[:2]):
            with torch.cuda.stream(s):
                outs.append(r.render_image(c2w, cam, u=u)[0])
        torch.cuda.synchronize()
        for a, b in zip(outs, imgs):
            assert torch.equal(a, b), rep
