"""Render / train step of Neural Graph Mapping on the gfx950 kernels.

Host-side mirror of the renderer half of ``NeuralGraphMap`` (run_mapping.py): ``Target`` /
``Prediction`` records (:43-69), ``render_ijs`` (:439-666), ``quadrature`` (:709-799),
``compute_losses`` (:1769-1872) and the sparse-Adam plumbing (:347-389, :668-707, :1183-1221), plus
the fused fast path ``optimization_iteration`` that the mapping loop calls once per iteration
(:1123-1181).  The SLAM / pose-graph loop itself stays in the caller.
"""
import ctypes as C
from collections import namedtuple
import warnings
from typing import Dict, Optional

import torch

from . import _capi as K
from . import ops
from .models import NeuralFieldSet

Target = namedtuple("Target", ["ijs", "c2ws", "near_distances", "far_distances", "gt_distances", "field_ids", "rgbds",
                               "rgb_mask", "depth_mask", "term_probs", "term_mask"])
Prediction = namedtuple("Prediction", ["rgbds", "color_vars", "depth_vars", "term_probs", "freespace_geometry",
                                       "tsdf_residuals"])


class Camera:
    """Pinhole intrinsics with the reference's pixel-centre convention (camera.py:15-116)."""

    def __init__(self, width, height, fx, fy, cx, cy, s=0.0, pixel_center=0.0):
        if s != 0:
            raise NotImplementedError("Skew != 0 not supported.")
        self.fx, self.fy = fx, fy
        self.cx = cx - pixel_center + 0.5
        self.cy = cy - pixel_center + 0.5
        self.s, self.width, self.height = s, width, height

    def get_pinhole_camera_parameters(self, pixel_center: float):
        return self.fx, self.fy, self.cx - 0.5 + pixel_center, self.cy - 0.5 + pixel_center, self.s


class LossConfig:
    """Loss weights of rm.py:129-135 (defaults of config/neural_graph_map.yaml:31-37)."""

    def __init__(self, termination_weight=0.0, photometric_weight=1.0, depth_weight=1.0, freespace_weight=40.0,
                 tsdf_weight=50.0):
        self.termination_weight, self.photometric_weight = termination_weight, photometric_weight
        self.depth_weight, self.freespace_weight, self.tsdf_weight = depth_weight, freespace_weight, tsdf_weight


# keys NeuralGraphMap._read_config reads unconditionally (rm.py:116-220) among those this path consumes: a config the
# reference would refuse with a KeyError is refused here too, instead of training with a silent default
REQUIRED_CONFIG_KEYS = ("termination_weight", "photometric_weight", "photometric_loss", "depth_weight", "depth_loss",
                        "freespace_weight", "geometry_mode", "num_samples_coarse", "num_samples_depth_guided")


def shipped_config(**overrides) -> dict:
    """The render / loss / optimiser keys of the shipped config (config/neural_graph_map.yaml:26-70) with their shipped
    values -- a complete config for this path to start from (tests, tools, examples); `overrides` replace entries."""
    cfg = dict(color_factor=1.0, geometry_factor=20.0, learning_rate=1e-3, field_radius=1.0, termination_weight=0.0,
               photometric_weight=1.0, photometric_loss="l1", depth_weight=1.0, depth_loss="huber", freespace_weight=40.0,
               tsdf_weight=50.0, near_distance=0.0, far_distance=8.0, pixel_block_size=8192, block_size=3000000,
               geometry_mode="nrgbd", truncation_distance=0.1, num_train_fields=32, num_rays_per_field=512,
               num_samples_coarse=8, num_samples_depth_guided=16, range_depth_guided=None, adam_eps=1e-15,
               adam_weight_decay=1e-5)
    cfg.update(overrides)
    return cfg


def check_config(config: dict) -> None:
    """Fail loudly on what the reference requires and on what is not built (loss modes of losses.py:10-75)."""
    missing = [k for k in REQUIRED_CONFIG_KEYS if k not in config]
    if missing:
        raise KeyError(f"config lacks {missing}: NeuralGraphMap._read_config reads them unconditionally (rm.py:116-220)")
    if config["photometric_loss"] not in K.PHOTO:
        raise NotImplementedError(f"photometric_loss {config['photometric_loss']!r} is not built (built: {sorted(K.PHOTO)}, "
                                  "losses.py:26-36)")
    if config["depth_loss"] not in K.DEPTH:
        raise NotImplementedError(f"depth_loss {config['depth_loss']!r} is not built (built: {sorted(K.DEPTH)}, losses.py:60-75)")
    if config["geometry_mode"] not in K.GEO:
        raise ValueError(f"Unsupported geometry mode {config['geometry_mode']}")        # rm.py:760 wording


def make_render_cfg(camera: Camera, config: dict, guided: bool = True) -> K.RenderCfg:
    """Reference config keys (rm.py:116-220) -> ngm_render_cfg.  Optional keys keep the reference's `.get` defaults
    (tsdf_weight 0.0, color / geometry factor 1.0, range_depth_guided -> truncation_distance)."""
    check_config(config)
    fx, fy, cx, cy, _ = camera.get_pinhole_camera_parameters(0.0)
    tau = config.get("truncation_distance", 0.1)
    rho = config.get("range_depth_guided") or tau
    return K.render_cfg(
        geometry_mode=config["geometry_mode"], num_samples_coarse=config["num_samples_coarse"],
        num_samples_guided=config["num_samples_depth_guided"] if guided else 0,
        geometry_factor=config.get("geometry_factor", 1.0), color_factor=config.get("color_factor", 1.0),
        truncation_distance=tau, range_depth_guided=rho, fx=fx, fy=fy, cx=cx, cy=cy,
        w_termination=config["termination_weight"], w_photometric=config["photometric_weight"],
        w_depth=config["depth_weight"], w_freespace=config["freespace_weight"],
        w_tsdf=config.get("tsdf_weight", 0.0), photometric_loss=config["photometric_loss"], depth_loss=config["depth_loss"])


class NeuralGraphRenderer:
    """Renderer + per-field optimiser state for a ``NeuralFieldSet`` (hot-path half of NeuralGraphMap)."""

    def __init__(self, model: NeuralFieldSet, camera: Camera, config: dict, device="cuda"):
        self._model, self._camera, self._config, self._device = model, camera, dict(config), device
        # two radii, as in the reference: NeuralGraphMap._field_radius (run_mapping.py:137: sampler margins, near / far, grid
        # spacing, the mesh colour pass' `+ 0.1`) and the MODEL's own (models.py:278-285: the scaling of local coordinates and
        # the default inside test of the kNN branch).  The shipped configs set both from one YAML anchor; the kernels' field
        # configuration always carries the model's.
        self._field_radius = config.get("field_radius", model._field_radius)
        self._fc = model.field_cfg()
        # hidden layers of the forward kernels: "f32" = exact-fp32 MFMA; "bf16x3" = exact three-way bf16 split with fp32
        # accumulation (include/ngm_hip.h ngm_matmul_mode; fp32-level accuracy, same tolerances, bitwise deterministic,
        # 16x the matrix rate; fails loudly where not compiled); "auto" (default) = the split wherever it is compiled
        mm = config.get("mlp_matmul", "auto")
        self._fc.matmul_mode = K.MATMUL[mm]
        # activation stash of two-hidden-layer networks on the split path (include/ngm_hip.h, ngm_activation_stash): "full"
        # (default: both hidden layers' outputs, 512 B per sample, fastest) or "half" (layer 0's only, 256 B per sample: half
        # the stash memory and HBM traffic -- a 4096 x 128 x 32-field batch needs 4.3 instead of 8.6 GB --, the backward
        # recomputes the other layer on the matrix pipe: step + 1.5-4 %).  Part of THIS renderer's field configuration
        # (ABI 10): it sizes this renderer's workspaces; other renderers of the process are unaffected.
        self._fc.activation_stash = K.STASH[config.get("activation_stash") or "full"]
        # accumulation of the hash-table gradient (ngm_hash_grad_atomics): "exact" (default: Q23.40 integer LDS atomics, bitwise
        # reproducible) or "float" (opt-in: fp32 LDS atomics like the reference's CUDA package -- not reproducible run to run)
        self._fc.hash_grad_atomics = K.HASH_ATOMICS[config.get("hash_grad_atomics") or "exact"]
        fc = self._fc
        compiled = (fc.encoding in (K.ENC["fourier"], K.ENC["none"]) and fc.skip_mode == K.SKIP["no"] and 1 <= fc.num_layers <= 2
                    and 32 < fc.dim_enc <= 64 and 32 < fc.dim_hidden <= 64)
        # what the library resolves "auto" to for batch shapes whose LDS plan has room for the weight planes
        self.mlp_matmul = ("bf16x3" if compiled else "f32") if mm == "auto" else mm
        self._rc_train = make_render_cfg(camera, config, guided=True)
        self._rc_cache = {}
        self._mode = "train"               # rm.py:114: a new map is in train mode
        self._global_map_dict = None       # supplied by the mapping loop: positions / orientations
        self._learning_rate = config.get("learning_rate", 1e-3)
        self._adam_eps = config.get("adam_eps", 1e-15)
        self._adam_weight_decay = config.get("adam_weight_decay", 0.0)
        self._optim_state: Optional[Dict[str, Dict[str, torch.Tensor]]] = None
        self._step = 0
        self._step_dev = None
        self._ws_cache = {}
        self.process_group = None          # torch.distributed group for the loss all-reduce (None: single GPU)
        self.eval_fused = True             # render_pixels: one ngm_render_eval_knn call (False: the staged per-block entry points)
        self.eval_fallbacks = []           # why a fused evaluation call was retried with a smaller block / left to the staged loop
        self.last_eval_path = None
        self.peer_exchange = None          # distributed.PeerExchange: the same sum as one kernel inside the captured iteration
        self.peer_check_interval = 256     # iterations between PeerExchange.check() calls (a device synchronisation each)
        self._peer_calls = 0
        self.field_draw_generator = None   # torch.Generator of sample_target_mv(field_draw="balanced_by_owner"), see there

    def last_matmul(self, kernel: str = "forward") -> Optional[str]:
        """The arithmetic the library resolved `mlp_matmul` to in the LAST launch of the fused forward ("forward"), the
        point evaluation ("points") or the kNN evaluation ("knn"): "f32" | "bf16x3" (None before the first launch).
        `auto` is resolved per kernel and batch shape (LDS plan), so this -- not `self.mlp_matmul`, the host-side
        expectation -- is what a measurement should be labelled with."""
        v = K.lib().ngm_debug_last_matmul({"forward": 0, "points": 1, "knn": 2}[kernel])
        return {K.MATMUL["f32"]: "f32", K.MATMUL["bf16x3"]: "bf16x3"}.get(v)

    # -- map bookkeeping supplied by the caller ------------------------------------------------
    def set_field_poses(self, positions: torch.Tensor, orientations: torch.Tensor):
        self._global_map_dict = {"positions": positions, "orientations": orientations, "num": positions.shape[0]}

    def add_fields(self, num_new: int):
        """NeuralGraphMap._add_fields (rm.py:364-389): grow params + zero moments, keep the shared step."""
        self._model.add_fields(num_new)
        allp = self._model.all_fields_params
        new_state = {}
        for k, v in allp.items():
            m, s = torch.zeros_like(v), torch.zeros_like(v)
            if self._optim_state is not None:
                m[:-num_new] = self._optim_state[k]["exp_avg"]
                s[:-num_new] = self._optim_state[k]["exp_avg_sq"]
            new_state[k] = {"exp_avg": m, "exp_avg_sq": s}
        self._optim_state = new_state

    # -- checkpoint interchange (rm.py:2147-2173) ------------------------------------------------
    def save_model(self, path: str) -> None:
        """torch.save of {map_dict, all_fields_params, state_dict}: the reference's checkpoint layout
        (optimizer moments are not saved there either).  With a peer exchange the health of every exchange so far is
        checked first: parameters trained behind a timed-out exchange are not written."""
        self.check_exchange()
        torch.save({"map_dict": self._global_map_dict, "all_fields_params": self._model.all_fields_params,
                    "state_dict": self._model.state_dict()}, path)

    def load_model(self, path: str) -> None:
        ck = torch.load(path, map_location=self._device)
        self._model.load_state_dict(ck["state_dict"])
        self._model.all_fields_params = {k: v.to(self._device) for k, v in ck["all_fields_params"].items()}
        self._global_map_dict = ck["map_dict"]
        self._model.refresh_lp()
        self._optim_state = {k: {"exp_avg": torch.zeros_like(v), "exp_avg_sq": torch.zeros_like(v)}
                             for k, v in self._model.all_fields_params.items()}

    @torch.no_grad()
    def evaluate_points(self, points: torch.Tensor, block_size: Optional[int] = None) -> torch.Tensor:
        """kNN-blended field values at arbitrary world points (the dense-grid evaluation that
        _extract_mesh feeds to marching cubes, rm.py:2255-2261, 2320-2332), in blocks."""
        num = self._global_map_dict["num"]
        pos = self._global_map_dict["positions"][:num]
        quat = self._global_map_dict["orientations"][:num]
        params = self._model.kernel_params()
        m = self._model
        lead = points.shape[:-1]
        pts = points.reshape(-1, 3)
        block = int(block_size or self._config.get("block_size", 3000000))
        outs = [ops.field_eval_knn(self._fc, params, pts[s:s + block], pos, quat, m._num_knn, m._distance_factor,
                                   m._outside_value) for s in range(0, pts.shape[0], block)]
        return torch.cat(outs).reshape(*lead, 4)

    def extract_mesh(self, mesh_file_path=None, resolution: Optional[float] = None, threshold: Optional[float] = None,
                     transform: Optional[torch.Tensor] = None, field_ids: Optional[torch.Tensor] = None, block: int = 200,
                     debug: Optional[dict] = None):
        """_extract_mesh (rm.py:2186-2384): dense grid -> HIP marching cubes -> vertex colours -> PLY (see mesh.py)."""
        from . import mesh
        return mesh.extract_mesh(self, mesh_file_path, resolution, threshold, transform, field_ids, block, debug)

    # -- reference-compatible API --------------------------------------------------------------
    def quadrature(self, sample_colors, sample_geometries, sample_distances, sample_depths, neus_isds=None):
        return ops.quadrature(self._rc_train, sample_colors, sample_geometries, sample_distances, sample_depths,
                              neus_isds)

    # -- train / eval sampling parameters (rm.py:1966-1974) ------------------------------------------
    def eval(self) -> None:
        """NeuralGraphMap.eval (rm.py:1966-1969): render_ijs / render_image sample `eval_num_samples` (configured, or derived
        from the training sample spacing, rm.py:199-207) in [eval_near_distance, eval_far_distance]."""
        self._mode = "eval"

    def train(self) -> None:
        """NeuralGraphMap.train (rm.py:1971-1974; the state a new map is in, :114): `num_samples_coarse` samples in
        [near_distance, far_distance]."""
        self._mode = "train"

    def _mode_sampling(self):
        """(_num_samples, _near_distance, _far_distance) of the current mode"""
        cfg = self._config
        if self._mode == "eval":
            return self.eval_num_samples(), cfg.get("eval_near_distance", 0.0), cfg.get("eval_far_distance", 8.0)
        return int(cfg["num_samples_coarse"]), cfg.get("near_distance", 0.0), cfg.get("far_distance", 8.0)

    def _rc_for(self, camera: Optional[Camera], guided: bool, overwrite: bool = True) -> K.RenderCfg:
        """ngm_render_cfg for `camera` (None: the constructor's) in the current mode; cached per (intrinsics, samples, flags)"""
        cam = camera or self._camera
        S, _, _ = self._mode_sampling()
        key = (cam.fx, cam.fy, cam.cx, cam.cy, S, bool(guided), bool(overwrite))
        rc = self._rc_cache.get(key)
        if rc is None:
            rc = make_render_cfg(cam, {**self._config, "num_samples_coarse": S}, guided=guided)
            rc.overwrite_behind_camera = int(bool(overwrite))
            self._rc_cache[key] = rc
        return rc

    def render_ijs(self, ijs, c2ws, camera=None, field_ids=None, use_vmap=False, near_distances=None,
                   far_distances=None, gt_distances=None, overwrite_samples_behind_camera=True, u_coarse=None,
                   u_guided=None, seed=0) -> Prediction:
        """NeuralGraphMap._render_ijs (rm.py:439-666), same arguments, same default branch.

        use_vmap=False (the default, rm.py:445): rays `ijs` (N,2) or (F,R,2) through the kNN-blended map -- all fields, or
        the subset `field_ids` (rm.py:502-508) --, `c2ws` (4,4) or per ray, optional per-ray near / far; with `gt_distances`
        and depth-guided samples configured, the second stratum + the free-space / TSDF vectors (rm.py:521-545, 624-639).
        Runs without autograd (stacked parameters are plain tensors in the reference too).
        use_vmap=True: the training branch, each ray row against ITS field; differentiable w.r.t. vmap_fields_params.
        `camera` is honoured on both (None = the constructor's camera); sample count and scalar near / far follow
        train() / eval().  Beyond the reference: `u_coarse` / `u_guided` replay the torch.rand draws of camera.py:274
        (shape (..., S)); otherwise the kernels draw from Philox with `seed`."""
        if use_vmap and field_ids is None:
            raise ValueError("field_ids=None only supported for use_vmap=False")                  # rm.py:497-498
        if not use_vmap:
            return self._render_ijs_knn(ijs, c2ws, camera, field_ids, near_distances, far_distances, gt_distances,
                                        overwrite_samples_behind_camera, u_coarse, u_guided, seed)
        self._model.set_vmap_fields(field_ids)
        for n, v in self._model.vmap_fields_params.items():
            if n not in K.NO_GRAD_PARAMS:                 # constants (hash shifts): no gradient, no Adam, no weight decay
                v.requires_grad_()
        _, near_c, far_c = self._mode_sampling()
        guided = gt_distances is not None and self._rc_train.num_samples_guided > 0
        # default True (rm.py:450); checked per sample
        rc = self._rc_for(camera, guided, overwrite_samples_behind_camera)
        pos = self._global_map_dict["positions"][field_ids]
        quat = self._global_map_dict["orientations"][field_ids]
        params = self._model.vmap_fields_params
        if rc.geometry_mode == K.GEO["neus"]:
            return self._render_ijs_staged(rc, ijs, c2ws, field_ids, near_distances, far_distances,
                                           gt_distances if guided else None, gt_distances, u_coarse,
                                           u_guided if guided else None, seed, overwrite_samples_behind_camera)
        rgbds, cvars, dvars, term, geoms, dists = ops.render_ijs_fused(
            self._fc, rc, params, ijs, c2ws, near_distances, far_distances, gt_distances if guided else None, pos, quat,
            u_coarse, u_guided if guided else None, seed, near_c, far_c)
        fs = ts = None
        tau = rc.truncation_distance
        if gt_distances is not None and geoms is not None:
            gt = gt_distances[..., None]
            if rc.w_freespace != 0.0:
                fs = geoms[dists < (gt - tau) * (gt != 0.0)] * tau                      # rm.py:624-630
            if rc.w_tsdf != 0.0:
                deltas = gt - dists
                m = (deltas.abs() < tau) & (gt != 0.0)
                ts = geoms[m] * tau - deltas[m]                                         # rm.py:632-639
        return Prediction(rgbds, cvars, dvars, term, fs, ts)

    @torch.no_grad()
    def _render_ijs_knn(self, ijs, c2ws, camera, field_ids, near, far, gt, overwrite, u_coarse, u_guided, seed,
                        params: Optional[dict] = None) -> Prediction:
        """_render_ijs(use_vmap=False) (rm.py:439-666 with :586-595): the rays are flattened to one row; without gt the
        whole call is ngm_render_eval_knn (samples, neighbour blend and quadrature in one pass over blocks of rays); with gt
        the per-sample geometry is needed for the free-space / TSDF vectors, so the staged entry points run
        (ngm_sample_rays_world -> ngm_field_eval_knn -> ngm_composite_fwd_packed), in blocks of `block_size` samples like
        the reference's batched_evaluation (rm.py:587-594)."""
        cfg, m, dev = self._config, self._model, self._device
        if self._rc_train.geometry_mode == K.GEO["neus"]:
            # rm.py:641-647 hands neus_isds=None to _quadrature on this branch, whose `None * float` raises (rm.py:754)
            raise TypeError("render_ijs(use_vmap=False) in geometry mode 'neus': the reference has no kNN branch for it "
                            "(neus_isds is None there, rm.py:641-647, 753-754); render with use_vmap=True")
        if ijs.dtype != torch.int64:
            raise TypeError("ijs must be int64 (row, col)")
        S, near_c, far_c = self._mode_sampling()
        n_g = int(cfg["num_samples_depth_guided"])
        guided = gt is not None and n_g > 0
        if guided and (near is None or far is None):
            # rm.py:522-526 compares the per-ray tensors with gt: None raises there as well
            raise TypeError("render_ijs: gt_distances with depth-guided samples needs per-ray near_distances and far_distances")
        lead = tuple(ijs.shape[:-1])
        N = 1
        for d in lead:
            N *= d
        flat = lambda t, *tail: None if t is None else t.reshape(N, *tail)
        ij, nr, fr, g = ijs.reshape(N, 2), flat(near), flat(far), flat(gt)
        if c2ws.dim() == 2:
            cw = c2ws
        else:
            cw = c2ws.expand(*lead, 4, 4).reshape(N, 4, 4)
        num = self._global_map_dict["num"]
        if field_ids is not None:                                                            # rm.py:502-508
            fids = field_ids.to(dev)
            pos, quat = self._global_map_dict["positions"][fids], self._global_map_dict["orientations"][fids]
        else:
            fids = None
            pos, quat = self._global_map_dict["positions"][:num], self._global_map_dict["orientations"][:num]
        if params is None:
            params = m.kernel_params()
        params = {k: v for k, v in params.items() if k != "_neus_sd"}
        rc = self._rc_for(camera, guided, overwrite)
        empty = lambda *shape: torch.empty(*shape, device=dev)
        if N == 0:
            fs = ts = None
            if gt is not None:
                fs = empty(0) if rc.w_freespace != 0.0 else None
                ts = empty(0) if rc.w_tsdf != 0.0 else None
            return Prediction(empty(*lead, 4), empty(*lead, 3), empty(*lead), empty(*lead), fs, ts)
        if gt is None and self.eval_fused:
            block = int(cfg.get("pixel_block_size", 8192))
            rb = int(cfg.get("eval_ray_block") or max(block, 32768))
            pred = self._fused_eval(lambda ray_block: ops.render_eval_knn(
                self._fc, rc, params, ij, cw, pos, quat, m._num_knn, m._distance_factor, m._outside_value, near=nr, far=fr,
                u=flat(u_coarse, S), seed=seed, near_const=near_c, far_const=far_c, field_index=fids, ray_block=ray_block),
                rb, block)
            if pred is not None:
                rgbd, cv, dv, term = pred
                return Prediction(rgbd.view(*lead, 4), cv.view(*lead, 3), dv.view(*lead), term.view(*lead), None, None)
        else:
            self.last_eval_path = "staged"
        St = S + (n_g if guided else 0)
        rays_per_block = max(1, int(cfg.get("block_size", 3000000)) // St)
        tau = rc.truncation_distance
        want_fs = gt is not None and rc.w_freespace != 0.0
        want_ts = gt is not None and rc.w_tsdf != 0.0
        behind = -100.0 if rc.geometry_mode in (K.GEO["occupancy"], K.GEO["density"]) else 1.0      # rm.py:614-622
        check_behind = overwrite and nr is not None and not bool((nr >= 0).all())                   # rm.py:494-495
        outs, fss, tss = [], [], []
        for s0 in range(0, N, rays_per_block):
            sl = slice(s0, s0 + rays_per_block)
            sub = lambda t: None if t is None else t[sl][None]
            pc, pw, dist = ops.sample_rays_world(rc, ij[sl][None], cw if cw.dim() == 2 else cw[sl][None], sub(nr), sub(fr),
                                                 sub(g) if guided else None, sub(flat(u_coarse, S)),
                                                 sub(flat(u_guided, n_g)) if guided else None, seed + s0,
                                                 near_const=near_c, far_const=far_c)
            out4 = ops.field_eval_knn(self._fc, params, pw.view(-1, 3), pos, quat, m._num_knn, m._distance_factor,
                                      m._outside_value, field_index=fids)
            dist, pc = dist.view(-1, St), pc.view(-1, St, 3)
            outs.append(ops.composite_packed(rc, out4, dist, pc))            # applies the behind-camera constant itself
            if want_fs or want_ts:
                geoms = out4.view(-1, St, 4)[..., 3]
                if check_behind:
                    geoms = torch.where(pc[..., 2] > 0, torch.full_like(geoms, behind), geoms)
                gg = g[sl][:, None]
                if want_fs:
                    fss.append(geoms[dist < (gg - tau) * (gg != 0.0)] * tau)             # rm.py:624-630
                if want_ts:
                    deltas = gg - dist
                    msk = (deltas.abs() < tau) & (gg != 0.0)
                    tss.append(geoms[msk] * tau - deltas[msk])                           # rm.py:632-639
        rgbd, cv, dv, term = (torch.cat([o[i] for o in outs]) for i in range(4))
        return Prediction(rgbd.view(*lead, 4), cv.view(*lead, 3), dv.view(*lead), term.view(*lead),
                          torch.cat(fss) if want_fs else None, torch.cat(tss) if want_ts else None)

    def _fused_eval(self, call, ray_block: int, min_block: int):
        """Run the fused evaluation call `call(ray_block)`; a block that does not fit (torch OOM, NGM_E_WORKSPACE) is halved
        down to `min_block`.  Returns None when the staged entry points must serve the call: the block cannot shrink
        further, or the library reports a shape the fused call does not support (NGM_E_UNSUPPORTED).  Every other error is
        raised.  Each distinct reason is recorded once in `eval_fallbacks`; `last_eval_path` names what ran."""
        rb = ray_block
        while True:
            try:
                out = call(rb)
                self.last_eval_path = f"fused, ray_block {rb}"
                return out
            except (K.NgmError, torch.cuda.OutOfMemoryError) as e:
                oom = isinstance(e, torch.cuda.OutOfMemoryError)
                code = getattr(e, "code", None)
                if not oom and code not in (K.NGM_E_WORKSPACE, K.NGM_E_UNSUPPORTED):
                    raise
                note = f"ray_block {rb}: {type(e).__name__}: {str(e)[:200]}"
                if note not in self.eval_fallbacks and len(self.eval_fallbacks) < 64:
                    self.eval_fallbacks.append(note)
                if oom:
                    torch.cuda.empty_cache()
                if rb <= min_block or code == K.NGM_E_UNSUPPORTED:
                    self.last_eval_path = "staged (fused call failed, see eval_fallbacks)"
                    return None
                rb = max(min_block, rb // 2)

    def _render_ijs_staged(self, rc, ijs, c2ws, field_ids, near, far, gt_sampler, gt, u_coarse, u_guided, seed,
                           overwrite_behind) -> Prediction:
        """_render_ijs as a chain of the standalone stages (sampler -> per-field evaluation -> quadrature), every stage
        an autograd function over its HIP kernels.  Serves the geometry mode the fused kernels do not: neus, whose
        occupancy couples neighbouring samples through the per-field learnable `_neus_sd` (rm.py:641-644, 753-758)."""
        params = self._model.vmap_fields_params
        pos = self._global_map_dict["positions"][field_ids]
        quat = self._global_map_dict["orientations"][field_ids]
        pc, pw, dists = ops.sample_rays_world(rc, ijs, c2ws, near, far, gt_sampler, u_coarse, u_guided, seed,
                                              near_const=self._config.get("near_distance", 0.0),
                                              far_const=self._config.get("far_distance", 8.0))
        F, R, S = dists.shape
        out = ops.field_eval(self._fc, {k: v for k, v in params.items() if k != "_neus_sd"}, pw.reshape(F, R * S, 3),
                             pos, quat).view(F, R, S, 4)
        colors = rc.color_factor * out[..., :3]                                              # rm.py:610
        geoms = out[..., 3]
        depths = -pc[..., 2]                                                                  # rm.py:612
        if overwrite_behind and near is not None and not bool((near >= 0).all()):             # rm.py:494-495
            const = -100.0 if rc.geometry_mode in (K.GEO["occupancy"], K.GEO["density"]) else 1.0   # rm.py:614-622
            geoms = torch.where(pc[..., 2] > 0, torch.full_like(geoms, const), geoms)
        isds = None
        if rc.geometry_mode == K.GEO["neus"]:
            isds = 1.0 / torch.abs(params["_neus_sd"].view(-1, 1, 1))                         # rm.py:641-644
        C, D, Cv, Dv, term, _ = ops.quadrature(rc, colors, geoms, dists, depths, isds)
        fs = ts = None
        tau = rc.truncation_distance
        if gt is not None:
            g = gt[..., None]
            if rc.w_freespace != 0.0:
                fs = geoms[dists < (g - tau) * (g != 0.0)] * tau                              # rm.py:624-630
            if rc.w_tsdf != 0.0:
                deltas = g - dists
                m = (deltas.abs() < tau) & (g != 0.0)
                ts = geoms[m] * tau - deltas[m]                                               # rm.py:632-639
        return Prediction(torch.cat([C, D[..., None]], -1), Cv, Dv, term, fs, ts)

    def optimization_iteration_staged(self, target: Target, u_coarse=None, u_guided=None, seed=0, update=True) -> dict:
        """The training iteration for configurations outside the fused kernels (geometry mode neus): staged render,
        torch loss, autograd through the stage kernels, the same sparse Adam (incl. `_neus_sd`)."""
        fids = target.field_ids
        pred = self.render_ijs(target.ijs, target.c2ws, field_ids=fids, use_vmap=True, near_distances=target.near_distances,
                               far_distances=target.far_distances, gt_distances=target.gt_distances,
                               u_coarse=u_coarse, u_guided=u_guided, seed=seed)
        loss = self.compute_losses(target, pred)
        params = self._model.vmap_fields_params
        names = [n for n, v in params.items() if v.requires_grad and n not in K.NO_GRAD_PARAMS]
        grads = torch.autograd.grad(loss["combined"], [params[n] for n in names], allow_unused=True)
        gd = {n: (g if g is not None else torch.zeros_like(params[n])) for n, g in zip(names, grads)}
        out = {k: v.detach() for k, v in loss.items()}
        out["prediction"] = pred
        if not update:
            out["grads"] = gd
            return out
        self._step += 1
        with torch.no_grad():
            for n in names:
                allp = self._model.all_fields_params[n]
                st = self._optim_state[n]
                ops.adam_sparse_(allp, st["exp_avg"], st["exp_avg_sq"], gd[n].contiguous(), fids, self._step,
                                 lr=self._learning_rate, eps=self._adam_eps, weight_decay=self._adam_weight_decay)
        if self._step_dev is not None:
            self._step_dev.fill_(self._step)
        self._model.refresh_lp()           # this path updates the fp32 masters tensor by tensor
        return out

    def compute_losses(self, target: Target, prediction: Prediction) -> dict:
        """_compute_losses (rm.py:1769-1872) on torch tensors; loss dict keys carry the mode like rm.py:1827, 1837."""
        rc = self._rc_train
        pk = "photometric_" + self._config["photometric_loss"]
        m = target.depth_mask & (prediction.term_probs > rc.term_threshold)
        loss = {}
        tm = target.term_mask
        loss["termination"] = ((prediction.term_probs[tm] - target.term_probs[tm]) ** 2).mean()
        diff = target.rgbds[m][:, :3] - prediction.rgbds[m][:, :3]
        dk = "depth_" + self._config["depth_loss"]
        p_d, t_d = prediction.rgbds[m][:, 3], target.rgbds[m][:, 3]
        if rc.photometric_mode == K.PHOTO["gaussian_nll"]:                   # losses.py:30-36 (the autograd path propagates
            cv = prediction.color_vars[m]                                     # through the variances: render_ijs_bwd with seeds on them)
            nlls = 0.5 * diff ** 2 / cv + torch.log(torch.sqrt(cv))
            loss[pk] = diff.abs().mean() if nlls.mean() > 2 else nlls.mean()
        else:
            loss[pk] = diff.abs().mean() if rc.photometric_mode == K.PHOTO["l1"] else (diff ** 2).mean()   # losses.py:26-29
        if rc.depth_mode == K.DEPTH["gaussian_nll"]:                         # losses.py:64-69
            dv = prediction.depth_vars[m] + 1e-15
            loss[dk] = (0.5 * (p_d - t_d) ** 2 / dv + torch.log(torch.sqrt(dv))).mean()
        elif rc.depth_mode == K.DEPTH["laplacian_nll"]:                      # losses.py:70-75
            dv = prediction.depth_vars[m]
            loss[dk] = ((t_d - p_d).abs() / torch.sqrt(0.5 * dv + 1e-6) + 0.5 * torch.log(2 * dv + 1e-6)).mean()
        else:
            loss[dk] = torch.nn.functional.huber_loss(p_d, t_d, delta=rc.huber_delta)
        total = (rc.w_termination * loss["termination"] + rc.w_photometric * loss[pk]
                 + rc.w_depth * loss[dk])
        if prediction.freespace_geometry is not None:
            loss["freespace"] = ((prediction.freespace_geometry - rc.truncation_distance) ** 2).mean()
            total = total + rc.w_freespace * loss["freespace"]
        if prediction.tsdf_residuals is not None:
            loss["tsdf"] = (prediction.tsdf_residuals ** 2).mean()
            total = total + rc.w_tsdf * loss["tsdf"]
        loss["combined"] = total
        return loss

    # -- training-target sampler (rm.py:1259-1459) ------------------------------------------------
    @torch.no_grad()
    def sample_target_mv(self, current_field_ids, c_c2w, nc_rgbd, frame_cid_to_ncid, num_train_fields, num_rays_per_field,
                         num_fields=None, camera: Optional[Camera] = None, draws: Optional[dict] = None,
                         field_draw: str = "reference", world_size: int = 1) -> Target:
        """NeuralGraphMap._sample_target_mv: choose the fields to train, find the keyframes that see them, sample
        keyframes and pixels, collect the RGB-D supervision.  State that the reference keeps on `self` is passed in:
        c_c2w (Nc,4,4) = _c_c2w_tensor, nc_rgbd (N,H,W,4) = _nc_rgbd_tensor, frame_cid_to_ncid (Nc,).
        The random draws are made with torch on this device in the reference's order (multinomial, multinomial,
        randn, multinomial, rand); `draws` (dict: subset_observed, subset_random, offsets, frame_cids, u_xy)
        replays recorded ones.  The geometry between the draws runs in two HIP kernels.
        field_draw="balanced_by_owner" (opt-in, with world_size) replaces the choice of fields by
        distributed.draw_fields_balanced: the same draw per owner rank with a quota of num_train_fields / world -- every
        rank trains equally many fields per iteration; not the reference's distribution, see DESIGN.md §7."""
        if field_draw not in ("reference", "balanced_by_owner"):
            raise NotImplementedError(f"field_draw={field_draw!r}: 'reference' or 'balanced_by_owner'")
        cam = camera or self._camera
        dev = self._device
        radius = self._field_radius + 0.0
        num_fields = self._global_map_dict["num"] if num_fields is None else num_fields
        d = draws or {}
        cur = current_field_ids.to(dev)
        if field_draw == "balanced_by_owner" and world_size > 1 and not draws:
            from . import distributed as D
            # every rank must draw the SAME set: a dedicated generator, seeded identically everywhere and consumed by nothing
            # else (the process-global CUDA stream also feeds the per-rank ray draws below, whose consumption may differ
            # between ranks -- sharded / early-exit sampling -- and would silently desynchronise the active sets)
            if self.field_draw_generator is None:
                self.field_draw_generator = torch.Generator(device="cpu").manual_seed(0x5EED0F1E1D5)
            field_ids = D.draw_fields_balanced(cur.cpu(), num_fields, num_train_fields, world_size,
                                               generator=self.field_draw_generator).to(dev)
            if getattr(self, "debug_check_field_draw", False) and self.process_group is not None:
                ref = field_ids.clone()
                torch.distributed.broadcast(ref, 0, group=self.process_group)
                assert torch.equal(ref, field_ids), "balanced_by_owner: ranks drew different active sets"
        else:                                                      # rm.py:1280-1319
            n_obs = min(num_train_fields // 2, len(cur))
            sub_obs = d["subset_observed"].to(dev) if draws else torch.multinomial(torch.ones(len(cur), device=dev), n_obs)
            obs_ids = cur[sub_obs]
            n_rand = min(num_train_fields - len(obs_ids), num_fields - len(obs_ids))
            if n_rand > 0:
                dist = torch.ones(num_fields, device=dev)
                dist[obs_ids] = 0.0
                sub_rand = d["subset_random"].to(dev) if draws else torch.multinomial(dist, n_rand)
                field_ids = torch.unique(torch.cat((torch.arange(num_fields, device=dev)[sub_rand], obs_ids)))
            else:
                field_ids = obs_ids
        pos_w = self._global_map_dict["positions"][field_ids].contiguous()
        if draws:
            offsets = d["offsets"].to(dev)
        else:
            offsets = torch.randn((20, 3), device=dev)
            offsets /= torch.linalg.norm(offsets, dim=-1, keepdim=True)
        fx, fy, cx, cy, _ = cam.get_pinhole_camera_parameters(0.0)
        kf = ops.keyframes_struct(c_c2w.contiguous(), nc_rgbd, frame_cid_to_ncid, fx, fy, cx, cy)
        kf_mask, bbox = ops.target_visibility(kf, pos_w, offsets, radius)
        fmask = kf_mask.any(-1)                                    # fields no keyframe sees are dropped (rm.py:1365-1379)
        kf_mask, field_ids, pos_w, bbox = kf_mask[fmask], field_ids[fmask], pos_w[fmask].contiguous(), bbox[fmask].contiguous()
        F, R = len(field_ids), num_rays_per_field
        frame_cids = d["frame_cids"].to(dev) if draws else torch.multinomial(kf_mask.float(), R, replacement=True)
        u_xy = d["u_xy"].to(dev) if draws else torch.rand(F, R, 2, device=dev)
        o = ops.target_rays(kf, pos_w, radius, bbox, frame_cids, u_xy)
        return Target(ijs=o["ijs"], c2ws=o["c2ws"], near_distances=o["near"], far_distances=o["far"],
                      gt_distances=o["gt"], field_ids=field_ids, rgbds=o["rgbds"], rgb_mask=o["rgb_mask"],
                      depth_mask=o["depth_mask"], term_probs=o["term_probs"], term_mask=o["term_mask"])

    def sample_target_sv(self, rgbd_image, c2w, active_field_ids, num_train_fields, num_rays_per_field,
                         camera: Optional[Camera] = None, draws: Optional[dict] = None, num_points: int = 50000) -> Target:
        """NeuralGraphMap._sample_target_sv (`update_mode: single_view`, rm.py:1461-1583): fields and rays from ONE RGB-D
        frame.  The depth image is back-projected, `num_points` (50 000 in the reference) of its points are drawn, fields
        whose sphere is crossed by at least num_rays_per_field of the segments camera -> point are candidates, rays are
        drawn among each field's crossing segments.  The three torch.multinomial draws are made on this device in the
        reference's order; `draws` (dict: subset_points, subset_fields, segments) replays recorded ones.  The
        segment-sphere test (fields x points) and the per-ray targets run in two HIP kernels."""
        cam = camera or self._camera
        dev = self._device
        radius = self._field_radius + 0.0
        d = draws or {}
        rgbd_image = rgbd_image.to(dev).contiguous()
        c2w = c2w.to(dev)
        active = active_field_ids.to(dev)
        pos_w = self._global_map_dict["positions"][active]
        pos_c = (pos_w - c2w[:3, 3]) @ c2w[:3, :3]                          # utils.transform_points(..., inv=True): R^T (p - t)
        fx, fy, cx, cy, _ = cam.get_pinhole_camera_parameters(0.0)
        depth = rgbd_image[..., 3]
        ijs = torch.nonzero(depth)                                         # camera.py:374: only pixels with depth
        dv = depth[ijs[:, 0], ijs[:, 1]]
        points = torch.stack(((ijs[:, 1].float() - cx) * dv / fx, -(ijs[:, 0].float() - cy) * dv / fy, -dv), -1)
        sub = d["subset_points"].to(dev) if draws else torch.multinomial(torch.ones(len(points), device=dev), num_points)
        points, ijs = points[sub].contiguous(), ijs[sub].contiguous()
        mins, maxs = points.min(0)[0], points.max(0)[0]
        aabb_mask = ((pos_c - radius) <= maxs).all(-1) & ((pos_c + radius) >= mins).all(-1)
        pos_in = pos_c[aabb_mask].contiguous()
        hit = ops.target_sv_intersect(pos_in, points, radius)
        seg_mask = hit.sum(-1) >= num_rays_per_field
        hit = hit[seg_mask]
        ids, pc = active[aabb_mask][seg_mask], pos_in[seg_mask]
        if len(hit) > num_train_fields:
            sf = d["subset_fields"].to(dev) if draws else torch.multinomial(torch.ones(len(hit), device=dev), num_train_fields)
            ids, pc, hit = ids[sf], pc[sf], hit[sf]
        segments = d["segments"].to(dev) if draws else torch.multinomial(hit.float(), num_rays_per_field)
        o = ops.target_sv_rays(pc.contiguous(), radius, ijs, segments, rgbd_image, fx, fy, cx, cy)
        return Target(ijs=o["ijs"], c2ws=c2w, near_distances=o["near"], far_distances=o["far"], gt_distances=o["gt"],
                      field_ids=ids, rgbds=o["rgbds"], rgb_mask=o["rgb_mask"], depth_mask=o["depth_mask"],
                      term_probs=o["term_probs"], term_mask=o["term_mask"])

    # -- eval path: render_image / PSNR (rm.py:402-437, 1966-2000; evaluation.py:46-56) ----------
    def eval_num_samples(self) -> int:
        """rm.py:199-207: derived from the training sample spacing unless configured."""
        cfg = self._config
        if cfg.get("eval_num_samples") is not None:
            return int(cfg["eval_num_samples"])
        n_g = cfg.get("num_samples_depth_guided", 0)
        tau = cfg.get("truncation_distance", 0.1)
        rho = cfg.get("range_depth_guided") or tau
        spacing = 2 * rho / n_g if n_g > 0 else 2 * self._field_radius / cfg["num_samples_coarse"]
        return int((cfg.get("eval_far_distance", 8.0) - cfg.get("eval_near_distance", 0.0)) / spacing)

    @torch.no_grad()
    def render_pixels(self, c2w: torch.Tensor, begin: int, end: int, params: Optional[dict] = None,
                      camera: Optional[Camera] = None, u: Optional[torch.Tensor] = None, seed: int = 0):
        """Pixels [begin, end) of the flattened (row-major) image, single stratum, kNN-blended fields (the loop body of
        render_image, rm.py:402-437 = _render_ijs(x, c2w, camera) per block); sample count and near / far of the current
        mode (call eval() first, as rm.py:1879, 1978 do).  `params` overrides the model's stacked parameters (the
        all-gathered set on a sharded map).  Returns (rgbds (n,4), depth_vars (n,))."""
        cam = camera or self._camera
        cfg = self._config
        S, near_c, far_c = self._mode_sampling()
        rc = self._rc_for(cam, False)
        w = cam.width
        dev = self._device
        idx = torch.arange(begin, end, device=dev)
        ijs = torch.stack((idx // w, idx % w), -1)
        num = self._global_map_dict["num"]
        pos = self._global_map_dict["positions"][:num]
        quat = self._global_map_dict["orientations"][:num]
        if params is None:
            params = self._model.kernel_params()
        params = {k: v for k, v in params.items() if k != "_neus_sd"}
        m = self._model
        block = int(cfg.get("pixel_block_size", 8192))
        if self.eval_fused:
            # the whole loop below as one call (ngm_render_eval_knn): same blocks, same draws, same arithmetic; the samples,
            # the blended field outputs and the camera-frame points stay out of memory.  `eval_ray_block` = rays per internal
            # block (default: max(pixel_block_size, 32768) -- the reference chunks at 8192 pixels to bound ITS memory; the
            # pair records of 32768 x 640 samples are 1.3 GB here, and larger blocks mean fewer launches and shorter tails:
            # 14.3 -> 13.1 ms per 640-sample image, 5.8 -> 4.2 ms at 128 samples).  The in-kernel jitter stream is seeded
            # per block, so the block size decides WHICH draws a pixel gets (not their distribution; explicit `u` is
            # unaffected); with eval_ray_block = pixel_block_size the image equals the staged loop below bit for bit.
            # Memory: the transient workspace grows with the block (about 1.2 GB at S = 640, K = 2; 3.7 GB at S = 1024, K = 4).  A
            # block that does not fit (torch OOM, NGM_E_WORKSPACE) is halved down to the reference's pixel_block_size; if even
            # that fails -- or the library reports a shape the fused call does not support (NGM_E_UNSUPPORTED) -- the staged
            # loop below runs, as it would have with eval_fused = False; any other error is raised.  A fall-back is recorded
            # once per reason (`eval_fallbacks`), never silent about WHICH path ran: `last_eval_path`.
            rb = int(cfg.get("eval_ray_block") or max(block, 32768))
            pred = self._fused_eval(lambda ray_block: ops.render_eval_knn(
                self._fc, rc, params, ijs, c2w, pos, quat, m._num_knn, m._distance_factor, m._outside_value,
                u=None if u is None else u[begin:end], seed=seed + begin, near_const=near_c, far_const=far_c,
                ray_block=ray_block), rb, block)
            if pred is not None:
                return pred[0], pred[2]
        else:
            self.last_eval_path = "staged"
        rgbds, dvars = [], []
        for s0 in range(0, ijs.shape[0], block):
            ij = ijs[s0:s0 + block]
            ub = None if u is None else u[begin + s0:begin + s0 + block][None]
            pc, pw, dist = ops.sample_rays_world(rc, ij, c2w, None, None, None, ub, None, seed + begin + s0,
                                                 near_const=near_c, far_const=far_c)
            out4 = ops.field_eval_knn(self._fc, params, pw.view(-1, 3), pos, quat, m._num_knn, m._distance_factor,
                                      m._outside_value)
            rgbd, _, dv, _ = ops.composite_packed(rc, out4, dist.view(-1, S), pc.view(-1, S, 3))
            rgbds.append(rgbd)
            dvars.append(dv)
        if not rgbds:
            return torch.empty(0, 4, device=dev), torch.empty(0, device=dev)
        return torch.cat(rgbds), torch.cat(dvars)

    @torch.no_grad()
    def render_image(self, c2w: torch.Tensor, camera: Optional[Camera] = None, u: Optional[torch.Tensor] = None,
                     seed: int = 0):
        """render_image (rm.py:402-437): every pixel of `camera`, single stratum, kNN-blended fields, in the current mode
        like the reference: its callers switch to eval() first (rm.py:1879-1906, 1978-2019); in train mode this renders
        with `num_samples_coarse` samples in [near_distance, far_distance], as the reference would.

        Returns (rgbds (H,W,4), depth_vars (H,W)).  `u` (H*W, S) optionally supplies the torch.rand draws."""
        cam = camera or self._camera
        rgbd, dv = self.render_pixels(c2w, 0, cam.height * cam.width, camera=cam, u=u, seed=seed)
        return rgbd.reshape(cam.height, cam.width, 4), dv.reshape(cam.height, cam.width)

    @staticmethod
    def psnr(prediction: torch.Tensor, target: torch.Tensor, crop: int = 0) -> float:
        """evaluation.psnr (evaluation.py:46-56): clamp to [0,1], optional border crop, data_range 1."""
        if crop > 0:
            prediction, target = prediction[crop:-crop, crop:-crop], target[crop:-crop, crop:-crop]
        mse = ((prediction.clamp(0.0, 1.0) - target.clamp(0.0, 1.0)) ** 2).mean()
        return float(10.0 * torch.log10(1.0 / mse))

    # -- fused fast path -----------------------------------------------------------------------
    def _workspace(self, F, R):
        key = (F, R)
        if key not in self._ws_cache:
            wsb = K.lib().ngm_render_workspace(C.byref(self._fc), C.byref(self._rc_train), F, R, 1)
            dev = self._device
            self._ws_cache[key] = dict(
                ws=torch.empty(wsb, device=dev, dtype=torch.uint8), wsb=wsb,
                rgbds=torch.empty(F, R, 4, device=dev), color_vars=torch.empty(F, R, 3, device=dev),
                depth_vars=torch.empty(F, R, device=dev), term_probs=torch.empty(F, R, device=dev),
                sums=torch.zeros(K.NGM_NUM_LOSS_SUMS, device=dev), loss=torch.zeros(8, device=dev),
                philox=torch.zeros(1, device=dev, dtype=torch.int64))
        return self._ws_cache[key]

    def optimization_iteration(self, target: Target, u_coarse=None, u_guided=None, seed=0, update=True) -> dict:
        """One training iteration (rm.py:1123-1221): fused forward + losses, (all-reduce), fused backward,
        sparse Adam on the touched fields.  Returns the loss dict (device scalars) and, with
        update=False, also the gradients.  Every launch is asynchronous on the current stream and the
        sequence is hipGraph-capturable (device-side step / jitter counters, no allocation after the
        first call with a given batch shape)."""
        if self._rc_train.geometry_mode == K.GEO["neus"] and not self._neus_fused():
            return self.optimization_iteration_staged(target, u_coarse, u_guided, seed, update)
        if target.ijs.shape[0] == 0:
            return self._idle_iteration(update)
        ctx = self._iteration_forward(target, u_coarse, u_guided, seed, advance=update)
        if self.process_group is not None:
            # the only cross-GPU exchange of the path: global loss sums / counts (64 bytes)
            self._exchange(ctx["w"]["sums"])
        return self._iteration_backward(ctx, update)

    def check_exchange(self):
        """Raise if the in-graph loss exchange ever timed out or lost step (its sums were partial then, i.e. the update of
        that iteration used wrong loss normalisers).  Synchronises; called automatically every `peer_check_interval`
        iterations, call it yourself before trusting losses / checkpoints of a multi-GPU run.  No-op without PeerExchange."""
        if self.peer_exchange is not None:
            self.peer_exchange.check()

    def _count_exchange(self):
        """host-side bookkeeping of one (launched or replayed) exchange; the periodic health check"""
        self._peer_calls += 1
        if self.peer_exchange is not None and self.peer_check_interval and self._peer_calls % self.peer_check_interval == 0:
            if not torch.cuda.is_current_stream_capturing():
                self.peer_exchange.check()

    def _exchange(self, sums: torch.Tensor):
        if self.peer_exchange is not None:
            self.peer_exchange.allreduce(sums)       # one kernel on the current stream: xGMI peer writes, capturable
            self._count_exchange()
        else:
            torch.distributed.all_reduce(sums, group=self.process_group)

    def _neus_fused(self) -> bool:
        """the neus geometry mode runs in the fused kernels for the Fourier / no encoding without skip connections
        (two-pass compositing in the forward, neighbour terms + per-field d/d _neus_sd in the backward); the staged path
        (standalone sampler / field / quadrature kernels under autograd) serves the other encodings"""
        fc = self._fc
        return fc.encoding in (K.ENC["fourier"], K.ENC["none"]) and fc.skip_mode == K.SKIP["no"]

    def _idle_iteration(self, update=True) -> dict:
        """A rank none of whose fields is active in this iteration: it contributes zeros to the loss
        all-reduce (which it must still enter), launches nothing else, and keeps the shared counters in step."""
        from . import distributed as D
        sums = torch.zeros(16, device=self._device)
        if self.process_group is not None:
            self._exchange(sums)
        rc = self._rc_train
        loss = D.loss_values_from_sums(sums, rc.w_termination, rc.w_photometric, rc.w_depth, rc.w_freespace, rc.w_tsdf,
                                       self._config["photometric_loss"], self._config["depth_loss"])
        if update:
            self._step += 1
            if self._step_dev is None:
                self._step_dev = torch.full((1,), self._step, device=self._device, dtype=torch.int64)
            else:
                self._step_dev += 1
        else:
            loss["grads"] = {}
        loss["prediction"] = None
        return loss

    def _iteration_forward(self, target: Target, u_coarse=None, u_guided=None, seed=0, advance=True) -> dict:
        """First half of the iteration: fused forward + local loss sums (everything before the all-reduce).
        One device counter counts the iterations: the forward adds it to the Philox offset, the loss-reduction kernel
        advances it (advance=True) and Adam then reads it as its step -- no launch of its own for the bookkeeping."""
        L = K.lib()
        fc, rc = self._fc, self._rc_train
        fids = target.field_ids
        F, R = target.ijs.shape[0], target.ijs.shape[1]
        names = K.param_names(fc)
        allp = {n: self._model.all_fields_params[n] for n in names}       # fp32 masters (Adam)
        kp = self._model.kernel_params()                                   # what the kernels read (reduced precision or masters)
        lp = None if self._model.lp_fields_params is None else {n: kp[n] for n in names}
        kernel_p = {n: kp[n] for n in names}
        neus = rc.geometry_mode == K.GEO["neus"]
        if neus:
            kernel_p["_neus_sd"] = self._model.all_fields_params["_neus_sd"]
        ps = ops.params_struct(fc, kernel_p, fids)                         # rows field_ids[f] in place: no gather
        w = self._workspace(F, R)
        keep = []
        if self._step_dev is None:
            self._step_dev = torch.full((1,), self._step, device=self._device, dtype=torch.int64)
        rays = ops.make_rays(rc, target.ijs, target.c2ws, target.near_distances, target.far_distances,
                             target.gt_distances, self._global_map_dict["positions"],
                             self._global_map_dict["orientations"], u_coarse, u_guided, seed, keep=keep,
                             pose_index=fids, philox_offset_dev=self._step_dev, philox_autoinc=advance)
        dm = target.depth_mask.view(torch.uint8) if target.depth_mask.dtype == torch.bool else target.depth_mask
        tm = target.term_mask
        if tm is not None and tm.dtype == torch.bool:
            tm = tm.view(torch.uint8)
        rgbds_t = ops._f32c(target.rgbds)
        tg = K.Targets(ops._ptr(rgbds_t), dm.data_ptr(), ops._ptr(tm), ops._ptr(target.term_probs))
        pred = K.Prediction(w["rgbds"].data_ptr(), w["color_vars"].data_ptr(), w["depth_vars"].data_ptr(),
                            w["term_probs"].data_ptr())
        st = ops._stream()
        # single GPU: nothing happens between forward and backward, so the loss partials are reduced by the backward
        # itself (deferred reduction, one launch less); with a process group the sums are needed here for the all-reduce
        defer = self.process_group is None
        K.check(L.ngm_render_fwd(C.byref(fc), C.byref(rc), C.byref(ps), C.byref(rays), C.byref(tg), C.byref(pred),
                                 None if defer else w["sums"].data_ptr(), w["ws"].data_ptr(), w["wsb"], st), "ngm_render_fwd")
        return dict(fc=fc, rc=rc, ps=ps, rays=rays, tg=tg, pred=pred, w=w, F=F, fids=fids, allp=allp, lp=lp, defer=defer,
                    neus=neus, keep=(keep, dm, tm, rgbds_t, target))

    def _iteration_backward(self, ctx: dict, update=True) -> dict:
        """Second half: compositing + MLP backward with the (global) loss sums, sparse Adam, counters."""
        L = K.lib()
        fc, rc, ps, rays, tg, pred, w, F, fids, allp = (ctx[k] for k in ("fc", "rc", "ps", "rays", "tg", "pred", "w", "F",
                                                                         "fids", "allp"))
        st = ops._stream()
        if "grads" not in w:
            w["grads"], w["gs"], w["gflat"] = ops.alloc_grads(fc, F, self._device)
        grads, gs = w["grads"], w["gs"]
        if ctx.get("neus") and "_neus_sd" not in grads:
            grads["_neus_sd"] = torch.zeros(F, device=self._device)
            gs.neus_sd = grads["_neus_sd"].data_ptr()
        sums_ptr = None if ctx.get("defer") else w["sums"].data_ptr()
        if update:
            # backward + sparse Adam in one call: the gradient-reduction kernel applies the update of the MLP tensors
            # itself (rm.py:1183-1221); the device counter already holds the new step (the loss reduction advanced it)
            self._step += 1                                  # one counter for all fields (rm.py:380-385)
            arr, n_mlp, lat = ops.adam_tensor_arrays(fc, allp, self._optim_state, grads, ctx.get("lp"))
            K.check(L.ngm_render_bwd_adam(C.byref(fc), C.byref(rc), C.byref(ps), C.byref(rays), C.byref(tg), C.byref(pred),
                                          sums_ptr, C.byref(gs), arr, n_mlp, lat, ops._ptr(fids), int(self._step),
                                          ops._ptr(self._step_dev), self._learning_rate, 0.9, 0.999, self._adam_eps,
                                          self._adam_weight_decay, w["loss"].data_ptr(), w["ws"].data_ptr(), w["wsb"], st),
                    "ngm_render_bwd_adam")
            if fc.encoding == K.ENC["triplane"]:      # the feature planes: gradient from the fixed-point scatter, same sparse Adam
                n = "_encoding.plane_coef"
                pl, stt, gp = allp[n], self._optim_state[n], grads[n]
                one = (K.AdamTensor * 1)(K.AdamTensor(pl.data_ptr(), stt["exp_avg"].data_ptr(), stt["exp_avg_sq"].data_ptr(),
                                                      gp.data_ptr(), pl.stride(0), gp.stride(0), gp[0].numel()))
                K.check(L.ngm_adam_sparse_multi(one, 1, ops._ptr(fids), F, int(self._step), ops._ptr(self._step_dev),
                                                self._learning_rate, 0.9, 0.999, self._adam_eps, self._adam_weight_decay, 0,
                                                None, st), "ngm_adam_sparse_multi")
            if ctx.get("neus"):      # the per-field standard deviation is a parameter of its own (rm.py:641-644): same Adam
                n = "_neus_sd"
                sd, stt = self._model.all_fields_params[n], self._optim_state[n]
                one = (K.AdamTensor * 1)(K.AdamTensor(sd.data_ptr(), stt["exp_avg"].data_ptr(), stt["exp_avg_sq"].data_ptr(),
                                                      grads[n].data_ptr(), 1, 1, 1))
                K.check(L.ngm_adam_sparse_multi(one, 1, ops._ptr(fids), F, int(self._step), ops._ptr(self._step_dev),
                                                self._learning_rate, 0.9, 0.999, self._adam_eps, self._adam_weight_decay, 0,
                                                None, st), "ngm_adam_sparse_multi")
        else:
            K.check(L.ngm_render_bwd(C.byref(fc), C.byref(rc), C.byref(ps), C.byref(rays), C.byref(tg), C.byref(pred),
                                     sums_ptr, C.byref(gs), w["loss"].data_ptr(), w["ws"].data_ptr(), w["wsb"],
                                     st), "ngm_render_bwd")
        lv = w["loss"]
        loss = {"combined": lv[0], "termination": lv[1], "photometric_" + self._config["photometric_loss"]: lv[2],
                "depth_" + self._config["depth_loss"]: lv[3],
                "freespace": lv[4], "tsdf": lv[5]}
        if not update:
            loss["grads"] = grads
        loss["prediction"] = Prediction(w["rgbds"], w["color_vars"], w["depth_vars"], w["term_probs"], None, None)
        return loss

    def capture_iteration(self, target: Target, seed=0, u_coarse=None, u_guided=None):
        """Capture optimization_iteration(target) into hipGraphs (torch.cuda.CUDAGraph); the returned callable
        replays it.  Tensors of `target` are read in place at every replay; the Adam step counter and the Philox
        jitter offset live on the device and advance inside the graph.  With a process group the iteration becomes
        two graphs (before / after the loss all-reduce) and the 64-byte collective is issued between the replays --
        unless `peer_exchange` is set: then the exchange is a kernel of the ONE captured graph."""
        if self._rc_train.geometry_mode == K.GEO["neus"] and not self._neus_fused():
            raise NotImplementedError("capture_iteration: the staged (neus) iteration allocates under autograd; call "
                                      "optimization_iteration directly")
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            for _ in range(2):                                # warm-up on a side stream (allocations, lazy init)
                out = self.optimization_iteration(target, u_coarse, u_guided, seed=seed)
        torch.cuda.current_stream().wait_stream(s)
        if self.process_group is None or self.peer_exchange is not None:
            # one graph: single GPU, or the loss exchange is a kernel of the iteration (distributed.PeerExchange)
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                out = self.optimization_iteration(target, u_coarse, u_guided, seed=seed)
            self._step -= 1                                   # the capture pass records, it does not execute

            def replay():
                graph.replay()
                self._step += 1
                if self.peer_exchange is not None:
                    self._count_exchange()                    # the exchange is a kernel of this graph: periodic health check
                return out
            replay.graph = graph
            return replay
        # thread-local capture mode: the process group's watchdog thread may touch the HIP runtime meanwhile
        step0 = self._step
        try:
            g1, g2 = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
            with torch.cuda.graph(g1, capture_error_mode="thread_local"):
                ctx = self._iteration_forward(target, u_coarse, u_guided, seed, advance=True)
            with torch.cuda.graph(g2, pool=g1.pool(), capture_error_mode="thread_local"):
                out = self._iteration_backward(ctx, True)
        except RuntimeError as err:
            # capture refused (runtime / collective library combination): plain launches still work, but the caller
            # is told -- an iteration of ~10 launches is launch-bound without the graph (`replay.graph is None`)
            self._step = step0
            torch.cuda.synchronize()
            warnings.warn(f"capture_iteration: graph capture refused ({err}); falling back to plain launches",
                          RuntimeWarning, stacklevel=2)

            def eager():
                return self.optimization_iteration(target, u_coarse, u_guided, seed=seed)
            eager.graph = None
            eager.capture_error = str(err)
            return eager
        self._step -= 1
        sums, group = ctx["w"]["sums"], self.process_group

        def replay2():
            g1.replay()
            torch.distributed.all_reduce(sums, group=group)
            g2.replay()
            self._step += 1
            return out
        replay2.graph = (g1, g2)
        return replay2
