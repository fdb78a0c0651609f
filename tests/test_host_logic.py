"""Host-side mirror of the reference interface (models / renderer / distributed helpers) on CPU."""
import torch

from conftest import load_golden, split_prefix
import pytest

from neural_graph_mapping_amd import _capi as K
from neural_graph_mapping_amd import distributed as D
from neural_graph_mapping_amd import models as M
from neural_graph_mapping_amd import renderer as Rr
from oracle import ngm_oracle as O

FIELD_KW = dict(
    encoding_type="neural_graph_mapping.positional_encodings.PositionalEncodingFourier",
    encoding_kwargs=dict(dim_in=3, dim_out=64, mu=0.0, sigma=4.0, raw_coords=True),
    num_layers=2, dim_out=4, dim_mlp_out=None, skip_mode="no", initial_geometry_bias=0.0, neus_initial_sd=1.0)
SET_KW = dict(dim_points=3, field_type="neural_graph_mapping.models.NeuralField", field_kwargs=FIELD_KW, num_knn=2,
              distance_factor=10.0, outside_value=1.0, field_radius=1.0, scale_mode="unit_cube")


def test_reference_class_names_resolve_to_local_counterparts():
    assert M.str_to_object("neural_graph_mapping.models.NeuralField") is M.NeuralField
    assert M.str_to_object("neural_graph_mapping.positional_encodings.PositionalEncodingNeRF") is M.PositionalEncodingNeRF
    assert M.str_to_object("neural_graph_mapping.models.NeuralFieldSet") is M.NeuralFieldSet


def test_field_set_parameter_store_matches_reference_layout():
    fs = M.NeuralFieldSet(**SET_KW)
    fs.add_fields(3)
    fs.add_fields(2)
    g = load_golden("g6_train_cfg0")
    ref_shapes = {k: tuple(v.shape[1:]) for k, v in split_prefix(g, "p::").items()}
    assert {k: tuple(v.shape[1:]) for k, v in fs.all_fields_params.items()} == ref_shapes
    assert all(v.shape[0] == 5 for v in fs.all_fields_params.values())
    # add_fields clones ONE prototype (models.py:254-257): all fields start identical
    w = fs.all_fields_params["_linears.0.weight"]
    assert torch.equal(w[0], w[4])
    fs.set_vmap_fields(torch.tensor([4, 1]))
    assert fs.vmap_fields_params["_linears.1.bias"].shape == (2, 64)
    fs.set_vmap_fields(None)
    assert fs.vmap_fields_params is fs.all_fields_params
    assert fs.numel() == fs._prototype_field.numel() * len(fs.all_fields_params)   # reference quirk kept
    fc = fs.field_cfg()
    assert (fc.dim_enc, fc.dim_hidden, fc.num_layers, fc.scale_mode) == (64, 64, 2, 2)


def test_unsupported_variants_raise():
    import pytest
    with pytest.raises(NotImplementedError):
        M.NeuralField(**{**FIELD_KW, "skip_mode": "rezero"})      # the reference's own constructor raises too
    cat = M.NeuralField(**{**FIELD_KW, "skip_mode": "concat"})    # models.py:105-119: layers after the first read H + D
    assert cat._linears[1].weight.shape == (64, 128) and cat._linears[2].weight.shape == (4, 128)
    with pytest.raises(ValueError):
        M.NeuralFieldSet(**{**SET_KW, "field_radius": None})
    M.NeuralFieldSet(**{**SET_KW, "num_knn": 8})                  # K = 1..8: unrolled neighbour lists
    M.NeuralFieldSet(**{**SET_KW, "num_knn": 16})                 # K = 9..16: the 16-slot instance (round 6)
    with pytest.raises(NotImplementedError, match="1 <= K <= 16"):  # the reference takes any K; fail at construction, citing the limit
        M.NeuralFieldSet(**{**SET_KW, "num_knn": 17})


def test_planar_field_sets_wiring(monkeypatch):
    """dim_points = 2 (models.py:236-238): what the kernels are handed is the z = 0 embedding; parameter shapes stay the
    reference's 2-D ones; what the embedding cannot express raises at construction (arithmetic: fixture G22, CPU + GPU)"""
    enc2 = dict(dim_in=2, dim_out=40, mu=0.0, sigma=4.0, raw_coords=True)
    kw = {**SET_KW, "dim_points": 2, "field_kwargs": {**FIELD_KW, "encoding_kwargs": enc2, "dim_mlp_out": 64, "neus_initial_sd": None}}
    fs = M.NeuralFieldSet(**kw)
    fs.add_fields(3)
    fs.set_vmap_fields(None)
    assert fs.all_fields_params["_encoding._linear.weight"].shape == (3, 38, 2)
    assert fs.all_fields_params["_linears.0.weight"].shape == (3, 64, 40)
    fc = fs.field_cfg()
    assert (fc.dim_enc, fc.dim_hidden, fc.raw_coords) == (41, 64, 1)
    seen = {}

    def vmap(fc, params, q, pos=None, quat=None):
        seen.update(q=q, pos=pos, quat=quat, w0=params["_linears.0.weight"], we=params["_encoding._linear.weight"])
        return torch.zeros(*q.shape[:-1], 4)
    monkeypatch.setattr(M.ops, "field_eval", vmap)
    comp = torch.tensor([[0.0, 1.0], [-1.0, 0.0], [0.6, -0.8]])
    fs(torch.ones(3, 5, 2), torch.ones(3, 2), comp, None, use_vmap=True)
    assert seen["q"].shape == (3, 5, 3) and float(seen["q"][..., 2].abs().max()) == 0.0 and seen["pos"].shape == (3, 3)
    q = seen["quat"]
    assert q.shape == (3, 4) and float(q[:, 1:3].abs().max()) == 0.0
    torch.testing.assert_close(torch.stack((q[:, 0] ** 2 - q[:, 3] ** 2, 2 * q[:, 0] * q[:, 3]), -1), comp, rtol=1e-6, atol=1e-7)
    assert seen["w0"].shape == (3, 64, 41) and float(seen["w0"][..., 2].abs().max()) == 0.0
    assert torch.equal(seen["w0"][..., 3:], fs.all_fields_params["_linears.0.weight"][..., 2:])
    assert seen["we"].shape == (3, 38, 3) and float(seen["we"][..., 2].abs().max()) == 0.0
    with pytest.raises(NotImplementedError, match="skip"):
        M.NeuralFieldSet(**{**kw, "field_kwargs": {**kw["field_kwargs"], "skip_mode": "concat"}})
    with pytest.raises(NotImplementedError, match="2-D Fourier or NeRF"):
        M.NeuralFieldSet(**{**kw, "field_kwargs": FIELD_KW})          # a 3-D encoding in a planar set
    with pytest.raises(NotImplementedError):
        M.NeuralFieldSet(**{**SET_KW, "dim_points": 4})


def test_forward_field_radius_argument_reaches_the_kernel_as_mask_radius_only(monkeypatch):
    """models.py:333-337, 367-378: the argument is the inside test's radius; scaling keeps the constructor's (host wiring only:
    the ops are replaced by recorders, the arithmetic is tested on the GPU against fixture G21)"""
    fs = M.NeuralFieldSet(**{**SET_KW, "field_radius": 0.8})
    fs.add_fields(2)
    fs.set_vmap_fields(None)
    seen = {}

    def knn(fc, params, pts, pos, quat, num_knn, dfac, outside, field_index=None, mask_radius=None):
        seen["knn"] = (fc.field_radius, mask_radius, num_knn, dfac, outside, field_index)
        return torch.zeros(pts.shape[0], 4)

    def vmap(fc, params, q, pos=None, quat=None):
        seen["vmap"] = fc.field_radius
        return torch.zeros(*q.shape[:-1], 4)
    monkeypatch.setattr(M.ops, "field_eval_knn", knn)
    monkeypatch.setattr(M.ops, "field_eval", vmap)
    pts, pos, quat = torch.zeros(2, 5, 3), torch.zeros(2, 3), torch.tensor([[1.0, 0, 0, 0]] * 2)
    assert fs(pts, pos, quat, None, use_vmap=False, field_radius=0.9).shape == (2, 5, 4)
    assert seen["knn"][:2] == (pytest.approx(0.8), pytest.approx(0.9)) and seen["knn"][2:5] == (2, 10.0, 1.0)
    fs(pts, pos, quat, None, use_vmap=False)
    assert seen["knn"][:2] == (pytest.approx(0.8), pytest.approx(0.8))
    fs(pts, pos, quat, None, use_vmap=True, field_radius=0.9)
    assert seen["vmap"] == pytest.approx(0.8)
    free = M.NeuralFieldSet(**{**SET_KW, "field_radius": None, "scale_mode": "no"})
    free.add_fields(2)
    with pytest.raises(TypeError):              # the reference evaluates `dists < None` (models.py:368)
        free(pts, pos, quat, None, use_vmap=False)


def test_fused_image_call_falls_back_instead_of_failing(monkeypatch):
    """a fused evaluation block that does not fit (torch OOM / NGM_E_WORKSPACE) is halved down to pixel_block_size,
    then the staged per-block loop runs; which path ran is readable (host wiring only: the ops are recorders)."""
    model = M.NeuralFieldSet(**SET_KW)
    cam = Rr.Camera(64, 48, 50.0, 50.0, 31.5, 23.5, pixel_center=0.0)
    r = Rr.NeuralGraphRenderer(model, cam, Rr.shipped_config(eval_num_samples=16, pixel_block_size=1024), device="cpu")
    r.add_fields(2)
    r.set_field_poses(torch.zeros(2, 3), torch.tensor([[1.0, 0, 0, 0]] * 2))
    r.eval()
    calls = []

    def fused(fc, rc, params, ijs, c2w, pos, quat, *a, ray_block=0, **kw):
        calls.append(ray_block)
        if ray_block > 8192:
            raise torch.cuda.OutOfMemoryError("fake")
        if ray_block > 1024:
            raise K.NgmError("ngm_render_eval_knn failed with status -3: ngm_render_eval_knn: workspace too small", code=K.NGM_E_WORKSPACE)
        n = ijs.shape[0]
        return torch.zeros(n, 4), None, torch.zeros(n), None
    monkeypatch.setattr(Rr.ops, "render_eval_knn", fused)
    monkeypatch.setattr(torch.cuda, "empty_cache", lambda: None)
    rgbd, dv = r.render_image(torch.eye(4))
    assert calls == [32768, 16384, 8192, 4096, 2048, 1024] and rgbd.shape == (48, 64, 4)
    assert r.last_eval_path == "fused, ray_block 1024" and len(r.eval_fallbacks) == 5

    staged = []
    monkeypatch.setattr(Rr.ops, "render_eval_knn", lambda *a, **k: (_ for _ in ()).throw(K.NgmError("status -2: unsupported shape", code=K.NGM_E_UNSUPPORTED)))
    monkeypatch.setattr(Rr.ops, "sample_rays_world", lambda rc, ij, *a, **k: (staged.append(ij.shape[0]),
                        (torch.zeros(ij.shape[0], 16, 3), torch.zeros(ij.shape[0], 16, 3), torch.zeros(ij.shape[0], 16)))[1])
    monkeypatch.setattr(Rr.ops, "field_eval_knn", lambda fc, p, pts, *a, **k: torch.zeros(pts.shape[0], 4))
    monkeypatch.setattr(Rr.ops, "composite_packed", lambda rc, o, d, pc: (torch.zeros(d.shape[0], 4), None, torch.zeros(d.shape[0]), None))
    rgbd, _ = r.render_image(torch.eye(4))
    assert staged == [1024, 1024, 1024] and r.last_eval_path.startswith("staged") and rgbd.shape == (48, 64, 4)


def test_camera_effective_principal_point():
    cam = Rr.Camera(640, 480, 554.25, 554.25, 319.5, 239.5, pixel_center=0.0)
    fx, fy, cx, cy, _ = cam.get_pinhole_camera_parameters(0.0)
    assert (cx, cy) == (319.5, 239.5)
    rc = Rr.make_render_cfg(cam, Rr.shipped_config())
    assert rc.num_samples_guided == 16 and abs(rc.range_depth_guided - 0.1) < 1e-7 and rc.w_tsdf == 50.0


def test_field_sharding_partitions_targets():
    ids = torch.tensor([0, 3, 4, 7, 9])
    T = Rr.Target(ijs=torch.zeros(5, 6, 2, dtype=torch.long), c2ws=torch.zeros(5, 6, 4, 4), near_distances=torch.zeros(5, 6),
                  far_distances=torch.ones(5, 6), gt_distances=torch.ones(5, 6), field_ids=ids, rgbds=torch.zeros(5, 6, 4),
                  rgb_mask=torch.ones(5, 6, dtype=torch.bool), depth_mask=torch.ones(5, 6, dtype=torch.bool),
                  term_probs=torch.ones(5, 6), term_mask=torch.ones(5, 6, dtype=torch.bool))
    seen = []
    for r in range(4):
        sh = D.shard_target(T, r, 4)
        assert (sh.field_ids % 4 == r).all() and sh.ijs.shape[0] == sh.field_ids.shape[0]
        seen += sh.field_ids.tolist()
    assert sorted(seen) == ids.tolist()
    assert D.local_field_slots(10, 1, 4).tolist() == [1, 5, 9]
    assert D.global_to_local(torch.tensor([1, 5, 9]), 4).tolist() == [0, 1, 2]


def test_loss_values_from_global_sums_match_oracle():
    g = load_golden("g6_train_3field")
    fs = O.FieldSpec(encoding="fourier", dim_enc=64, num_layers=2)
    rs = O.RenderSpec(num_samples_coarse=8, num_samples_depth_guided=16, termination_weight=0.5)
    params = {k: v for k, v in split_prefix(g, "p::").items() if k != "_neus_sd"}
    t = split_prefix(g, "t::")
    nrgbd = O.CameraSpec(640, 480, 554.2562584220408, 554.2562584220408, 319.5, 239.5)
    pred = O.render_ijs(t["ijs"], t["c2ws"], nrgbd, g["pos"], g["quat"], params, fs, rs, t["near"], t["far"], t["gt"],
                        g["u_coarse"], g["u_guided"])
    m = t["depth_mask"] & (pred["term_probs"] > 0.8)
    e = pred["rgbds"][m][:, 3] - t["rgbds"][m][:, 3]
    hub = torch.where(e.abs() < 0.05, 0.5 * e * e, 0.05 * (e.abs() - 0.025))
    tm = t["term_mask"]
    s = torch.zeros(16)
    s[0] = (t["rgbds"][m][:, :3] - pred["rgbds"][m][:, :3]).abs().sum(); s[1] = m.sum()
    s[2] = hub.sum(); s[3] = m.sum()
    s[4] = ((pred["freespace_geometry"] - 0.1) ** 2).sum(); s[5] = pred["freespace_geometry"].numel()
    s[6] = (pred["tsdf_residuals"] ** 2).sum(); s[7] = pred["tsdf_residuals"].numel()
    s[8] = ((pred["term_probs"][tm] - t["term_probs"][tm]) ** 2).sum(); s[9] = tm.sum()
    vals = D.loss_values_from_sums(s, 0.5, 1.0, 1.0, 40.0, 50.0)
    ref = split_prefix(g, "loss::")
    for k in ref:
        torch.testing.assert_close(vals[k], ref[k], rtol=2e-4, atol=1e-6)


# ------------------------------------------------------------------ checkpoint interchange (rm.py:2147-2173)
def _cpu_renderer(num_fields):
    from neural_graph_mapping_amd import models as M
    from neural_graph_mapping_amd import renderer as Rr
    model = M.NeuralFieldSet(dim_points=3, field_type="neural_graph_mapping.models.NeuralField", field_kwargs=dict(
        encoding_type="neural_graph_mapping.positional_encodings.PositionalEncodingFourier",
        encoding_kwargs=dict(dim_in=3, dim_out=64, mu=0.0, sigma=4.0, raw_coords=True), num_layers=2, dim_out=4,
        neus_initial_sd=1.0), num_knn=2, distance_factor=10.0, outside_value=1.0, field_radius=1.0, scale_mode="unit_cube")
    r = Rr.NeuralGraphRenderer(model, Rr.Camera(32, 24, 27.7, 27.7, 15.5, 11.5), Rr.shipped_config(), device="cpu")
    if num_fields:
        r.add_fields(num_fields)
    return r


def test_checkpoint_layout_matches_the_reference_and_round_trips(tmp_path):
    """G14 = the dict the real reference's save_model hands to torch.save.  Loading it must fill the stacked parameters,
    the prototype state and the map dict; what save_model writes must have the same keys, dtypes and shapes, so the
    reference's load_model (rm.py:2165-2172) can read it back."""
    import os
    import torch
    from conftest import GOLDEN
    path = os.path.join(GOLDEN, "g14_checkpoint_reference_layout.pt")
    ref = torch.load(path)
    r = _cpu_renderer(0)
    r.load_model(path)
    assert r._global_map_dict["num"] == 3 and r._global_map_dict["positions"].shape == (32, 3)
    for k, v in ref["all_fields_params"].items():
        assert torch.equal(r._model.all_fields_params[k], v), k
    for k, v in ref["state_dict"].items():
        assert torch.equal(r._model.state_dict()[k], v), k
    assert set(r._optim_state) == set(ref["all_fields_params"]) and all(
        float(s["exp_avg"].abs().max()) == 0 for s in r._optim_state.values())       # moments are not checkpointed
    out = tmp_path / "mine.pt"
    r.save_model(str(out))
    mine = torch.load(str(out))
    assert list(mine) == list(ref) and list(mine["state_dict"]) == list(ref["state_dict"])
    assert list(mine["all_fields_params"]) == list(ref["all_fields_params"]) and set(mine["map_dict"]) == set(ref["map_dict"])
    for k in ref["all_fields_params"]:
        a, b = mine["all_fields_params"][k], ref["all_fields_params"][k]
        assert a.dtype == b.dtype and torch.equal(a, b), k
    r2 = _cpu_renderer(1)                                    # a map that already holds fields is replaced, not merged
    r2.load_model(str(out))
    assert r2._model.all_fields_params["_linears.0.weight"].shape[0] == 3


def test_config_fails_loudly_like_the_reference():
    """_read_config reads the loss weights and modes unconditionally (rm.py:116-220): a config it would refuse with a
    KeyError is refused here; loss modes of losses.py that the fused kernels do not build raise instead of silently
    training with l1 / huber."""
    cam = Rr.Camera(640, 480, 554.25, 554.25, 319.5, 239.5)
    good = Rr.shipped_config()
    rc = Rr.make_render_cfg(cam, good)
    assert rc.photometric_mode == K.PHOTO["l1"] and rc.depth_mode == K.DEPTH["huber"] and rc.w_freespace == 40.0
    assert Rr.make_render_cfg(cam, {**good, "photometric_loss": "l2"}).photometric_mode == K.PHOTO["l2"]
    for key in ("termination_weight", "photometric_weight", "photometric_loss", "depth_weight", "depth_loss", "freespace_weight",
                "geometry_mode", "num_samples_coarse", "num_samples_depth_guided"):
        bad = {k: v for k, v in good.items() if k != key}
        with pytest.raises(KeyError):
            Rr.make_render_cfg(cam, bad)
    no_tsdf = {k: v for k, v in good.items() if k != "tsdf_weight"}
    assert Rr.make_render_cfg(cam, no_tsdf).w_tsdf == 0.0                 # config.get("tsdf_weight", 0.0), rm.py:135
    # every mode of losses.py is built (round 4: the variance-weighted ones too); an unknown mode still raises
    assert Rr.make_render_cfg(cam, {**good, "photometric_loss": "gaussian_nll"}).photometric_mode == K.PHOTO["gaussian_nll"]
    for mode in ("gaussian_nll", "laplacian_nll"):
        assert Rr.make_render_cfg(cam, {**good, "depth_loss": mode}).depth_mode == K.DEPTH[mode]
    with pytest.raises(NotImplementedError):
        Rr.make_render_cfg(cam, {**good, "photometric_loss": "huber"})
    with pytest.raises(NotImplementedError):
        Rr.make_render_cfg(cam, {**good, "depth_loss": "l1"})
    with pytest.raises(ValueError):
        Rr.make_render_cfg(cam, {**good, "geometry_mode": "sdf"})
    with pytest.raises(KeyError):                                          # the renderer's constructor goes through the same check
        _cpu_renderer_cfg({"num_samples_coarse": 8})


def _cpu_renderer_cfg(cfg):
    from neural_graph_mapping_amd import models as M
    model = M.NeuralFieldSet(dim_points=3, field_type="neural_graph_mapping.models.NeuralField", field_kwargs=dict(
        encoding_type="neural_graph_mapping.positional_encodings.PositionalEncodingFourier",
        encoding_kwargs=dict(dim_in=3, dim_out=64, mu=0.0, sigma=4.0, raw_coords=True), num_layers=2, dim_out=4,
        neus_initial_sd=1.0), num_knn=2, distance_factor=10.0, outside_value=1.0, field_radius=1.0, scale_mode="unit_cube")
    return Rr.NeuralGraphRenderer(model, Rr.Camera(32, 24, 27.7, 27.7, 15.5, 11.5), cfg, device="cpu")


def test_balanced_by_owner_field_draw():
    """Opt-in sampler policy of DESIGN.md §7: every owner rank gets num_train_fields / world fields per iteration, half of them
    among its observed fields; same generator state -> same set on every rank; world 1 = the reference's draw."""
    from neural_graph_mapping_amd import distributed as D
    NF, FA, W = 200, 32, 8
    cur = torch.arange(150, 200)                       # the 50 most recent fields are observed
    for seed in range(5):
        g = torch.Generator().manual_seed(seed)
        ids = D.draw_fields_balanced(cur, NF, FA, W, generator=g)
        assert len(ids) == FA and len(torch.unique(ids)) == FA and bool((ids[1:] > ids[:-1]).all())
        per_owner = torch.bincount(ids % W, minlength=W)
        assert per_owner.tolist() == [FA // W] * W
        for o in range(W):                             # half of each quota among the owner's observed fields (it has 6-7)
            assert int(((ids % W == o) & (ids >= 150)).sum()) >= (FA // W) // 2
        ids2 = D.draw_fields_balanced(cur, NF, FA, W, generator=torch.Generator().manual_seed(seed))
        assert torch.equal(ids, ids2)
    # quota remainder goes to the first owners; an owner with fewer fields than its quota trains all it has
    ids = D.draw_fields_balanced(torch.arange(0, 4), 10, 7, 4, generator=torch.Generator().manual_seed(0))
    assert torch.bincount(ids % 4, minlength=4).tolist() == [2, 2, 2, 1]
    ids = D.draw_fields_balanced(torch.empty(0, dtype=torch.int64), 5, 8, 4, generator=torch.Generator().manual_seed(0))
    assert ids.tolist() == [0, 1, 2, 3, 4]
    # the reference's global draw for comparison: the worst rank of an iteration gets well over the mean
    worst = []
    g = torch.Generator().manual_seed(1)
    for _ in range(200):
        r = D.draw_fields_reference(cur, NF, FA, generator=g)[0]
        assert len(r) == FA
        worst.append(int(torch.bincount(r % W, minlength=W).max()))
    assert sum(worst) / len(worst) > 6.0
    # world 1: the reference's draw itself
    a = D.draw_fields_balanced(cur, NF, FA, 1, generator=torch.Generator().manual_seed(3))
    b = D.draw_fields_reference(cur, NF, FA, generator=torch.Generator().manual_seed(3))[0]
    assert torch.equal(a, b)


def test_host_philox_restatement_known_answers():
    """tests/gpu_common.host_philox_uniform (the checker of test_in_kernel_philox_equals_host_philox) IS Philox4x32-10: the
    known-answer vectors of the Random123 distribution (kat_vectors: philox4x32 10 rounds)."""
    import sys
    import os
    sys.path.insert(0, os.path.dirname(__file__))
    import importlib.util
    spec = importlib.util.spec_from_file_location("_philox_host", os.path.join(os.path.dirname(__file__), "_philox_host.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    z = m.philox4x32_10((0, 0, 0, 0), (0, 0))
    assert z == (0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8)
    f = m.philox4x32_10((0xffffffff,) * 4, (0xffffffff, 0xffffffff))
    assert f == (0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd)
    p = m.philox4x32_10((0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344), (0xa4093822, 0x299f31d0))
    assert p == (0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1)
    # the element -> (block, word) mapping of the samplers (ngm_device.h philox_uniform / jitter_fill): element i of a stream is
    # word i & 3 of block i >> 2, so eight consecutive elements are exactly the eight words of two consecutive blocks
    import numpy as np
    seed, off, stream = 0x299f31d0a4093822, 0x03707344, 7
    u = m.host_philox_uniform(seed, off, np.arange(40, 48, dtype=np.uint64), stream)
    exp = []
    for blk in (10, 11):
        exp += list(m.philox4x32_10((blk, 0, stream, off), (seed & 0xffffffff, seed >> 32)))
    assert [int(x * 16777216.0) for x in u] == [w >> 8 for w in exp]
    # the weighted-bin sampler keeps one block per element: word 0 = bin draw, word 1 = offset draw
    ub = m.host_philox_uniform(seed, off, np.array([42], dtype=np.uint64), stream, word=1)
    assert int(ub[0] * 16777216.0) == m.philox4x32_10((42, 0, stream, off), (seed & 0xffffffff, seed >> 32))[1] >> 8


def test_recorded_bench_line_carries_the_contract():
    """the last bench line recorded on an MI355X (profiles/r05b_bench_line.json = stdout of `python bench.py`): every key the driver
    and the judge read, the roofline and cpu_baseline objects, value = units / time"""
    import json
    import os
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "r05b_bench_line.json")
    d = json.loads(open(path).read().strip().split("\n")[-1])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["higher_is_better"] is True and d["scaling"] == "weak" and d["vs_baseline"] is None
    assert "workload" in d["config"] and "model" not in d["config"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in d["roofline"], k
    assert abs(d["roofline"]["frac"] - d["roofline"]["achieved"] / d["roofline"]["peak"]) < 1e-6
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in d["cpu_baseline"], k
    assert d["cpu_baseline"]["kind"] == "port"
    samples = 8 * 512 * 128
    assert abs(d["value"] - samples / (d["ms_per_step"] * 1e-3)) / d["value"] < 1e-6


def test_render_ijs_signature_is_the_reference_one():
    """rm.py:440-451: argument order and defaults of `_render_ijs` -- `use_vmap` defaults to False (the kNN branch), so
    vis_blender.py:236-238's call without it lands there; use_vmap=True without field_ids raises the reference's ValueError.
    train() / eval() switch sample count and scalar bounds like rm.py:1966-1974 (a new map is in train mode, :114)."""
    import inspect
    sig = inspect.signature(Rr.NeuralGraphRenderer.render_ijs)
    names = list(sig.parameters)[1:10]
    assert names == ["ijs", "c2ws", "camera", "field_ids", "use_vmap", "near_distances", "far_distances", "gt_distances",
                     "overwrite_samples_behind_camera"]
    assert sig.parameters["use_vmap"].default is False and sig.parameters["field_ids"].default is None
    assert sig.parameters["overwrite_samples_behind_camera"].default is True
    model = M.NeuralFieldSet(**SET_KW)
    cam = Rr.Camera(64, 48, 50.0, 50.0, 31.5, 23.5, pixel_center=0.0)
    r = Rr.NeuralGraphRenderer(model, cam, Rr.shipped_config(eval_num_samples=40, eval_far_distance=6.0, far_distance=5.0,
                                                           near_distance=0.1), device="cpu")
    assert r._mode_sampling() == (8, 0.1, 5.0)
    r.eval()
    assert r._mode_sampling() == (40, 0.0, 6.0)
    r.train()
    assert r._mode_sampling() == (8, 0.1, 5.0)
    with pytest.raises(ValueError, match="field_ids=None only supported for use_vmap=False"):
        r.render_ijs(torch.zeros(1, 4, 2, dtype=torch.long), torch.eye(4), cam, None, True)
    # the render configuration follows the camera that is handed in, not the constructor's
    other = Rr.Camera(32, 24, 27.7, 26.0, 15.5, 11.5, pixel_center=0.0)
    rc = r._rc_for(other, False)
    assert (round(rc.fx, 3), round(rc.fy, 3), rc.cx, rc.cy) == (27.7, 26.0, 15.5, 11.5) and r._rc_for(None, False).fx == 50.0


def test_sample_rays_weighted_argument_checks():
    """None / half-given arguments raise the reference's ValueError (camera.py:260-261) before anything is indexed"""
    from neural_graph_mapping_amd import ops
    rc = K.render_cfg(num_samples_coarse=4, num_samples_guided=0)
    ijs = torch.zeros(3, 2, dtype=torch.long)
    with pytest.raises(ValueError, match="Either both or none"):
        ops.sample_rays_weighted(rc, ijs, torch.zeros(3, 6), None)
    with pytest.raises(ValueError, match="required"):
        ops.sample_rays_weighted(rc, ijs, None, None)
    with pytest.raises(ValueError, match="go together"):
        ops.sample_rays_weighted(rc, ijs, torch.zeros(3, 6), torch.zeros(3, 5), u_bin=torch.zeros(3, 4))
