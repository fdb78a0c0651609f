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
    for S in (128, 640):
        model = M.NeuralFieldSet(dim_points=3, field_type="neural_graph_mapping.models.NeuralField", field_kwargs=dict(
            encoding_type="neural_graph_mapping.positional_encodings.PositionalEncodingFourier",
            encoding_kwargs=dict(dim_in=3, dim_out=64, mu=0.0, sigma=4.0, raw_coords=True), num_layers=2, dim_out=4),
            num_knn=2, distance_factor=10.0, outside_value=1.0, field_radius=0.5, scale_mode="unit_cube").to(dev)
        cfg = Rr.shipped_config(field_radius=0.5, eval_near_distance=0.0, eval_far_distance=8.0, eval_num_samples=S)
        r = Rr.NeuralGraphRenderer(model, cam, cfg, device=dev)
        r.add_fields(NF)
        r.set_field_poses(pos.to(dev), quat.to(dev))
        c2w = torch.eye(4, device=dev)
        r.render_image(c2w)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n = 3
        for _ in range(n):
            r.render_image(c2w)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / n
        out[f"S{S}"] = dict(ms_per_image=round(dt * 1e3, 2), ray_samples_per_s=640 * 480 * S / dt, fields=NF)
    print(json.dumps(dict(workload="render_image 640x480, Fourier(64)+2x64 fields, kNN blend K=2", **out)))


if __name__ == "__main__":
    main()
