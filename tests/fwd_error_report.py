"""Forward parity error of the fused training forward against the oracle on a mid-size batch (prints max abs / rel errors).
    python tests/fwd_error_report.py        (GPU box; uses the oracle as the checker, like the tests do)"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))  # the checker (oracle) is test infrastructure: this report lives under tests/
import test_gpu_parity as T  # noqa: E402

O = T.O


def main():
    F, R, n_c, n_g = 4, 256, 64, 64
    fs = O.FieldSpec(encoding="fourier", dim_enc=64, num_layers=2)
    rs = O.RenderSpec(num_samples_coarse=n_c, num_samples_depth_guided=n_g)
    pos, quat, t = T.synth_target(F, R, seed=21)
    params = O.init_params(fs, F, seed=4, sigma=float(os.environ.get("SIGMA", "4.0")))
    u_c, u_g = torch.rand(F, R, n_c), torch.rand(F, R, n_g)
    with torch.no_grad():
        pred = O.render_ijs(t["ijs"], t["c2ws"], T.NRGBD, pos, quat, params, fs, rs, t["near"], t["far"], t["gt"], u_c, u_g)
    r = T.make_renderer(dict(encoding="fourier", dim_enc=64, num_layers=2),
                        dict(num_samples_coarse=n_c, num_samples_depth_guided=n_g), F, {k: v for k, v in params.items()})
    r.set_field_poses(pos.to(T.DEV), quat.to(T.DEV))
    res = r.optimization_iteration(T.make_target(t, torch.arange(F)), u_c.to(T.DEV), u_g.to(T.DEV), update=False)
    a, b = res["prediction"].rgbds.cpu(), pred["rgbds"]
    print("rgbd max abs err %.3e  max rel err (|ref|>1e-2) %.3e" % (float((a - b).abs().max()),
          float(((a - b).abs() / b.abs().clamp_min(1e-2)).max())))
    a, b = res["prediction"].term_probs.cpu(), pred["term_probs"]
    print("term max abs err %.3e" % float((a - b).abs().max()))


if __name__ == "__main__":
    main()
