"""Diagnostic for the opt-in bf16 split path (ngm_matmul_mode): run the same launch N times, report every launch whose
result differs bitwise from the first, how many rays / which fields differ and by how much -- for the training forward
(activation stash written) and for the plain render (no stash).  python tools/b3_determinism.py [N]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from gpu_common import DEV, make_renderer, make_target, synth_target  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 500
FOURIER = dict(encoding="fourier", dim_enc=64, num_layers=2)
for mm in ("bf16x3", "f32"):
    for (F, R, n_c, n_g) in ((3, 37, 20, 4), (8, 512, 64, 64)):
        r = make_renderer(FOURIER, dict(num_samples_coarse=n_c, num_samples_depth_guided=n_g, mlp_matmul=mm), F)
        g = torch.Generator(device=DEV).manual_seed(1)
        with torch.no_grad():
            for k, v in r._model.all_fields_params.items():
                if v.dim() > 1:
                    v.add_(0.05 * torch.randn(v.shape, device=DEV, generator=g))
        pos, quat, t = synth_target(F, R, seed=5)
        r.set_field_poses(pos.to(DEV), quat.to(DEV))
        tgt = make_target(t, torch.arange(F))
        ids = torch.arange(F, device=DEV)
        for mode in ("train", "render"):
            def run():
                if mode == "train":
                    return r.optimization_iteration(tgt, seed=9, update=False)["prediction"].rgbds.clone()
                with torch.no_grad():
                    return r.render_ijs(tgt.ijs, tgt.c2ws, None, field_ids=ids, near_distances=tgt.near_distances,
                                        far_distances=tgt.far_distances, gt_distances=None, seed=9).rgbds.clone()
            ref = run()
            bad = []
            n = N if F == 3 else max(50, N // 5)
            for i in range(n):
                o = run()
                if not torch.equal(o, ref):
                    d = (o - ref).abs()
                    rays = (d.amax(-1) > 0)
                    bad.append((i, int(rays.sum()), float(d.max()), rays.nonzero()[:4].tolist()))
            print(f"{mm:7s} F={F} R={R} S={n_c + n_g} {mode:6s}: {len(bad)}/{n} launches differ", bad[:6], flush=True)
