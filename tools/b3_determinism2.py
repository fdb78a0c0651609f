"""Second diagnostic: at the full batch, training forward -- which launch is the odd one, and which agrees with fp32?"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from gpu_common import DEV, make_renderer, make_target, synth_target  # noqa: E402

FOURIER = dict(encoding="fourier", dim_enc=64, num_layers=2)
F, R, n_c, n_g = 8, 512, 64, 64
pos, quat, t = synth_target(F, R, seed=5)
tgt = make_target(t, torch.arange(F))
res = {}
for mm in ("f32", "bf16x3"):
    for trial in range(2):
        r = make_renderer(FOURIER, dict(num_samples_coarse=n_c, num_samples_depth_guided=n_g, mlp_matmul=mm), F)
        g = torch.Generator(device=DEV).manual_seed(1)
        with torch.no_grad():
            for k, v in r._model.all_fields_params.items():
                if v.dim() > 1:
                    v.add_(0.05 * torch.randn(v.shape, device=DEV, generator=g))
        r.set_field_poses(pos.to(DEV), quat.to(DEV))
        outs = []
        for i in range(12):
            o = r.optimization_iteration(tgt, seed=9, update=False)
            torch.cuda.synchronize()
            outs.append((o["prediction"].rgbds.clone(), o["prediction"].term_probs.clone(), float(o["combined"])))
        res[(mm, trial)] = outs
        eq = ["=" if torch.equal(outs[i][0], outs[i - 1][0]) else "x" for i in range(1, 12)]
        print(mm, "trial", trial, "consecutive launches equal:", "".join(eq), "losses", [round(x[2], 6) for x in outs[:4]], flush=True)
f = res[("f32", 0)][0][0]
for i in (0, 1, 5, 11):
    b = res[("bf16x3", 0)][i][0]
    d = (b - f).abs()
    print(f"bf16x3 launch {i} vs f32: max {float(d.max()):.3e}, rays with |d| > 1e-4: {int((d.amax(-1) > 1e-4).sum())}, per field "
          f"{(d.amax(-1) > 1e-4).sum(-1).tolist()}")
b0, b1 = res[("bf16x3", 0)][0][0], res[("bf16x3", 1)][0][0]
print("first launches of two fresh renderers equal:", torch.equal(b0, b1), " last launches equal:",
      torch.equal(res[("bf16x3", 0)][11][0], res[("bf16x3", 1)][11][0]))
