"""Import the *real* reference (read-only, /root/reference) on CPU with stubs.

TEST INFRASTRUCTURE ONLY.  Runs only in the build container (the reference
does not exist on the GPU box); used by ``make_golden.py`` to generate the
committed fixtures and by ``tests/test_oracle_vs_reference.py`` (skipped when
/root/reference is absent).  Nothing from the reference is copied: missing
third-party modules are replaced by mocks, and the two pytorch3d functions the
hot path really executes (``quaternion_invert``/``quaternion_apply``,
``knn_points``) are restated from their public textbook definitions.
Recipe: SURVEY.md section 8c.
"""
import os
import sys
import types
from unittest.mock import MagicMock

import torch

REF_SRC = "/root/reference/src"


def reference_available() -> bool:
    return os.path.isdir(os.path.join(REF_SRC, "neural_graph_mapping"))


def _quaternion_raw_multiply(a, b):
    aw, ax, ay, az = torch.unbind(a, -1)
    bw, bx, by, bz = torch.unbind(b, -1)
    ow = aw * bw - ax * bx - ay * by - az * bz
    ox = aw * bx + ax * bw + ay * bz - az * by
    oy = aw * by - ax * bz + ay * bw + az * bx
    oz = aw * bz + ax * by - ay * bx + az * bw
    return torch.stack((ow, ox, oy, oz), -1)


def _quaternion_invert(q):
    return q * q.new_tensor([1, -1, -1, -1])


def _quaternion_apply(q, p):
    real = p.new_zeros(p.shape[:-1] + (1,))
    p4 = torch.cat((real, p), -1)
    out = _quaternion_raw_multiply(_quaternion_raw_multiply(q, p4), _quaternion_invert(q))
    return out[..., 1:]


def _matrix_to_quaternion(m):
    """real-first unit quaternion of a rotation matrix (textbook: largest of w, x, y, z as the pivot); the sign is
    irrelevant for rotating points, which is all utils.transform_quaternions' result is used for"""
    m = m.double()
    lead = m.shape[:-2]
    m = m.reshape(-1, 3, 3)
    out = []
    for r in m:
        t = r.trace()
        if t > 0:
            s = torch.sqrt(t + 1.0) * 2
            q = torch.stack([0.25 * s, (r[2, 1] - r[1, 2]) / s, (r[0, 2] - r[2, 0]) / s, (r[1, 0] - r[0, 1]) / s])
        else:
            i = int(torch.argmax(torch.diagonal(r)))
            j, k = (i + 1) % 3, (i + 2) % 3
            s = torch.sqrt(1.0 + r[i, i] - r[j, j] - r[k, k]) * 2
            q = torch.zeros(4, dtype=torch.float64)
            q[0] = (r[k, j] - r[j, k]) / s
            q[1 + i] = 0.25 * s
            q[1 + j] = (r[j, i] + r[i, j]) / s
            q[1 + k] = (r[k, i] + r[i, k]) / s
        out.append(q / q.norm())
    return torch.stack(out).reshape(*lead, 4).float()


def _knn_points(p1, p2, K=1, return_sorted=True, **kw):
    d2 = torch.cdist(p1.double(), p2.double()).float() ** 2
    # exact squared distances (cdist may use the mm trick): recompute directly
    d2 = ((p1[:, :, None, :] - p2[:, None, :, :]) ** 2).sum(-1)
    vals, idx = torch.topk(d2, K, dim=-1, largest=False, sorted=True)
    return vals, idx, None


def import_reference():
    """Returns (run_mapping, models, camera, positional_encodings, losses, utils)."""
    if not reference_available():
        raise RuntimeError("reference not present")
    if REF_SRC not in sys.path:
        sys.path.insert(0, REF_SRC)
    mocked = [
        "pytorch3d", "pytorch3d.io", "pytorch3d.io.ply_io", "pytorch3d.ops",
        "pytorch3d.ops.ball_query", "pytorch3d.ops.marching_cubes",
        "open3d", "rerun", "wandb", "yoco", "evo", "evo.core", "evo.core.trajectory",
        "torchmetrics", "torchmetrics.functional", "torchmetrics.image",
        "torchmetrics.image.lpip", "trimesh", "pyrender",
    ]
    for m in mocked:
        if m not in sys.modules:
            sys.modules[m] = MagicMock()
    tr = types.ModuleType("pytorch3d.transforms")
    tr.quaternion_invert = _quaternion_invert
    tr.quaternion_apply = _quaternion_apply
    tr.quaternion_raw_multiply = _quaternion_raw_multiply
    tr.quaternion_multiply = _quaternion_raw_multiply
    tr.matrix_to_quaternion = _matrix_to_quaternion
    sys.modules["pytorch3d.transforms"] = tr
    sys.modules["pytorch3d"].transforms = tr
    knn = types.ModuleType("pytorch3d.ops.knn")
    knn.knn_points = _knn_points
    sys.modules["pytorch3d.ops.knn"] = knn
    pe = types.ModuleType("permutohedral_encoding")

    class PermutoEncoding(torch.nn.Module):  # placeholder, never instantiated here
        def __init__(self, *a, **k):
            super().__init__()

    pe.PermutoEncoding = PermutoEncoding
    sys.modules["permutohedral_encoding"] = pe

    from neural_graph_mapping import camera, losses, models, positional_encodings, utils
    from neural_graph_mapping import run_mapping

    return run_mapping, models, camera, positional_encodings, losses, utils


NRGBD_CAMERA = dict(width=640, height=480, fx=554.2562584220408, fy=554.2562584220408,
                    cx=319.5, cy=239.5, pixel_center=0.0)


def make_config(*, skip_mode="no", encoding="fourier", dim_enc=64, num_layers=2, dim_mlp_out=None,
                num_octaves=8, fourier_mu=0.0, fourier_sigma=4.0,
                geometry_mode="nrgbd", num_samples_coarse=16, num_samples_depth_guided=16,
                geometry_factor=20.0, color_factor=1.0, truncation_distance=0.1,
                field_radius=1.0, scale_mode="unit_cube", termination_weight=0.0,
                freespace_weight=40.0, tsdf_weight=50.0, neus_initial_sd=1.0,
                far_distance=8.0, eval_far_distance=8.0, eval_num_samples=None, photometric_loss="l1", depth_loss="huber"):
    if encoding == "fourier":
        enc_type = "neural_graph_mapping.positional_encodings.PositionalEncodingFourier"
        enc_kwargs = dict(dim_in=3, dim_out=dim_enc, mu=fourier_mu, sigma=fourier_sigma,
                          raw_coords=True)
    elif encoding == "nerf":
        enc_type = "neural_graph_mapping.positional_encodings.PositionalEncodingNeRF"
        enc_kwargs = dict(dim_in=3, num_octaves=num_octaves, start_octave=0)
    else:
        raise ValueError(encoding)
    cfg = dict(
        dataset_type="neural_graph_mapping.slam_dataset.SLAMDataset",
        dataset_config={},
        model_type="neural_graph_mapping.models.NeuralFieldSet",
        model_kwargs=dict(
            dim_points=3,
            field_type="neural_graph_mapping.models.NeuralField",
            field_kwargs=dict(
                encoding_type=enc_type, encoding_kwargs=enc_kwargs, num_layers=num_layers,
                dim_out=4, dim_mlp_out=dim_mlp_out, skip_mode=skip_mode,
                initial_geometry_bias=0.0, neus_initial_sd=neus_initial_sd,
            ),
            num_knn=2, distance_factor=10.0, field_radius=field_radius,
            scale_mode=scale_mode, outside_value=1.0,
        ),
        device="cpu", learning_rate=1e-3, adam_eps=1e-15, adam_weight_decay=1e-5,
        freeze_model=False, termination_weight=termination_weight, photometric_weight=1.0,
        photometric_loss=photometric_loss, depth_weight=1.0, depth_loss=depth_loss,
        freespace_weight=freespace_weight, tsdf_weight=tsdf_weight, field_radius=field_radius,
        block_size=3000000, pixel_block_size=8192, num_train_fields=32,
        num_rays_per_field=512, num_samples_depth_guided=num_samples_depth_guided,
        range_depth_guided=None, preview_res_factor=0.3, render_frames=[],
        render_frame_freq=200, extract_mesh_frame_freq=100, extract_mesh_frames=[],
        extract_mesh_fields=[], log_iteration_freq=100, num_iterations_per_frame=5,
        rerun_vis=False, rerun_save=None, rerun_connect_addr=None,
        disable_relative_fields=False, geometry_mode=geometry_mode,
        truncation_distance=truncation_distance, color_factor=color_factor,
        geometry_factor=geometry_factor, single_field_id=None, update_mode="multi_view",
        near_distance=0.0, far_distance=far_distance, num_samples_coarse=num_samples_coarse,
        eval_far_distance=eval_far_distance, eval_near_distance=0.0,
        benchmark=False, loglevel=30,
    )
    if eval_num_samples is not None:
        cfg["eval_num_samples"] = eval_num_samples
    return cfg


def build_map(rm, cfg, num_fields, positions, orientations, seed=0):
    """Construct the reference NeuralGraphMap on CPU with `num_fields` fields."""
    torch.manual_seed(seed)
    # NeuralField creates its rezero scalars with device="cuda" (models.py:113): there is no GPU in the build
    # container, so torch.empty is asked for the CPU instead while the model is constructed
    real_empty = torch.empty

    def cpu_empty(*a, **k):
        if k.get("device") == "cuda":
            k["device"] = "cpu"
        return real_empty(*a, **k)
    torch.empty = cpu_empty
    try:
        ngm = rm.NeuralGraphMap(cfg)
    finally:
        torch.empty = real_empty
    ngm._optimizer = None
    n = num_fields
    if ngm._global_map_dict["positions"].shape[0] < n:
        ngm._extend_map_dict(n)
    ngm._global_map_dict["num"] = n
    ngm._global_map_dict["positions"][:n] = positions
    ngm._global_map_dict["orientations"][:n] = orientations
    ngm._model.add_fields(n)
    return ngm
