"""Render-only throughput of the evaluation path (render_image: sampler -> kNN-blended field evaluation -> quadrature)
on a synthetic map: 640x480 pixels, eval-style S samples per ray, N fields on a grid.  Prints one JSON object."""
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from neural_graph_mapping_amd import models as M  # noqa: E402
from neural_graph_mapping_amd import renderer as Rr  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    cam = Rr.Camera(640, 480, 554.2562584220408, 554.2562584220408, 319.5, 239.5, pixel_center=0.0)
    g = torch.arange(-2.0, 2.01, 0.5)
    pos = torch.stack(torch.meshgrid(g, g[:7], torch.tensor([-3.0, -2.5]), indexing="ij"), -1).reshape(-1, 3)
    NF = pos.shape[0]
    quat = torch.zeros(NF, 4)
    quat[:, 0] = 1
    out = {}
    for S in tuple(int(v) for v in os.environ.get("NGM_EVAL_S", "128,640").split(",")):
        model = M.NeuralFieldSet(dim_points=3, field_type="neural_graph_mapping.models.NeuralField", field_kwargs=dict(
            encoding_type="neural_graph_mapping.positional_encodings.PositionalEncodingFourier",
            encoding_kwargs=dict(dim_in=3, dim_out=64, mu=0.0, sigma=4.0, raw_coords=True), num_layers=2, dim_out=4),
            num_knn=2, distance_factor=10.0, outside_value=1.0, field_radius=0.5, scale_mode="unit_cube").to(dev)
        cfg = Rr.shipped_config(field_radius=0.5, eval_near_distance=0.0, eval_far_distance=8.0, eval_num_samples=S)
        if os.environ.get("NGM_EVAL_RAY_BLOCK"):
            cfg["eval_ray_block"] = int(os.environ["NGM_EVAL_RAY_BLOCK"])
        r = Rr.NeuralGraphRenderer(model, cam, cfg, device=dev)
        r.add_fields(NF)
        r.set_field_poses(pos.to(dev), quat.to(dev))
        r.eval()
        c2w = torch.eye(4, device=dev)
        r.render_image(c2w)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n = 3
        for _ in range(n):
            r.render_image(c2w)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / n
        out[f"S{S}"] = dict(ms_per_image=round(dt * 1e3, 2), ray_samples_per_s=640 * 480 * S / dt, fields=NF,
                            knn_matmul=r.last_matmul("knn"))
        if S == 640:
            # roofline of the dominant kernel (k_knn_eval: per-field MLP tiles over the (point, neighbour) pairs): HIP events
            # around its launches (C-ABI hooks) + the number of pairs it evaluates, counted here with torch on the same
            # sample points (a point is evaluated for its K nearest fields when the nearest is closer than the radius)
            from neural_graph_mapping_amd import _capi as K
            from neural_graph_mapping_amd import ops
            import ctypes as C
            L = K.lib()
            L.ngm_profile_reset(); L.ngm_profile_enable(1)
            r.render_image(c2w)
            torch.cuda.synchronize()
            L.ngm_profile_enable(0)
            kern = {}
            for name in ("knn_assign", "knn_eval"):
                ms, n = C.c_double(0), C.c_int64(0)
                L.ngm_profile_read(K.KERNEL_IDS[name], C.byref(ms), C.byref(n))
                kern[name] = dict(total_ms=ms.value, launches=n.value)
            rc = Rr.make_render_cfg(cam, {**cfg, "num_samples_coarse": S, "num_samples_depth_guided": 0}, guided=False)
            pairs = 0
            idx = torch.arange(0, 640 * 480, device=dev)
            ijs = torch.stack((idx // 640, idx % 640), -1)
            posd = pos.to(dev)
            for s0 in range(0, ijs.shape[0], 8192):
                _, pw, _ = ops.sample_rays_world(rc, ijs[s0:s0 + 8192], c2w, None, None, None, None, None, s0, near_const=0.0, far_const=8.0)
                pts = pw.reshape(-1, 3)
                for c0 in range(0, pts.shape[0], 1 << 20):
                    d2 = ((pts[c0:c0 + (1 << 20), None, :] - posd[None]) ** 2).sum(-1).min(-1)[0]
                    pairs += 2 * int((d2 < 0.25).sum())
            flop = 16896.0 * pairs
            t = kern["knn_eval"]["total_ms"] * 1e-3
            out["roofline_eval"] = dict(bound="mfma", kernel="k_knn_eval<2,2,2> (" + (r.last_matmul("knn") or "?") + ")", pairs=pairs,
                                        algorithmic_flop=flop, kernel_ms=round(t * 1e3, 3), achieved=flop / t / 1e12, peak=157.3,
                                        unit="TFLOP/s", frac=flop / t / 1e12 / 157.3, knn_assign_ms=round(kern["knn_assign"]["total_ms"], 3),
                                        note="jitter differs between the counted pass and the timed pass (Philox offsets): the pair "
                                             "count is exact to ~1e-3")
    if os.environ.get("NGM_EVAL_HASH", "1") != "0":
        # the reference's default network (permutohedral hash 16 x 2 + 1 x 32, neural_graph_map.yaml:6-20) on the same image
        model = M.NeuralFieldSet(dim_points=3, field_type="neural_graph_mapping.models.NeuralField", field_kwargs=dict(
            encoding_type="neural_graph_mapping.positional_encodings.PermutohedralEncoding",
            encoding_kwargs=dict(pos_dim=3, log2_hashmap_size=12, nr_levels=16, nr_feat_per_level=2, coarsest_scale=1.0,
                                 finest_scale=1e-4, init_scale=1e-5), num_layers=1, dim_out=4),
            num_knn=2, distance_factor=10.0, outside_value=1.0, field_radius=0.5, scale_mode="unit_cube").to(dev)
        cfg = Rr.shipped_config(field_radius=0.5, eval_near_distance=0.0, eval_far_distance=8.0, eval_num_samples=640)
        r = Rr.NeuralGraphRenderer(model, cam, cfg, device=dev)
        r.add_fields(NF)
        r.set_field_poses(pos.to(dev), quat.to(dev))
        r.eval()
        c2w = torch.eye(4, device=dev)
        r.render_image(c2w)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            r.render_image(c2w)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 3
        out["S640_hash"] = dict(ms_per_image=round(dt * 1e3, 2), ray_samples_per_s=640 * 480 * 640 / dt, fields=NF,
                                network="hash 16x2 (T=4096) + 1x32", knn_matmul=r.last_matmul("knn"))
    print(json.dumps(dict(workload="render_image 640x480, Fourier(64)+2x64 fields (and the default hash network), kNN blend K=2", **out)))


if __name__ == "__main__":
    main()
