"""-m gpu: long training runs against the REAL reference (fixture G13, tests/golden/make_golden.py:g13_training_run).

Why two different assertions.  The train step is a chaotic map: the loss is discontinuous in the prediction (ReLU, the
mask term > 0.8 of rm.py:1787) and Adam with eps = 1e-15 turns every near-zero gradient into a +-lr step, so two runs
of the reference ITSELF that start one ulp apart (parameters scaled by 1 + 1e-7; recorded in the fixture) drift apart
exponentially: ~1e-6 relative after 10 iterations, ~1e-3 after 100, and their held-out PSNRs differ by up to +-1 dB at
any single checkpoint of a long run.  Hence
  A  trajectory parity is asserted over 100 iterations at cfg0 size with the reference's own one-ulp sensitivity as
     the yardstick (the kernels may be no further from the reference than a small multiple of what the reference is
     from itself), plus the plain 2e-3 bar while the trajectories are still resolvable (10 iterations, as G7);
  B  "PSNR within 0.1 dB of the reference" (BASELINE.json north_star) is asserted where it is a well-defined number: the
     ensemble mean over 96 independently initialised fields x 41 checkpoints of the second half of a 1200-iteration
     run (3936 PSNR values; the single values of two runs differ by ~2.5 dB rms, the ensemble means by ~0.04 dB:
     measured kernels 29.160 dB vs reference 29.158 dB).
Every batch is regenerated from its seed by tests/golden/scene.py; the jitter draws are the torch.rand calls the
reference makes after torch.manual_seed (camera.py:274): coarse, then depth-guided."""
import os
import sys

import pytest
import torch

from conftest import load_golden, split_prefix

pytestmark = pytest.mark.gpu

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
import scene  # noqa: E402
from gpu_common import DEV, close, make_renderer, make_target  # noqa: E402

FOURIER = dict(encoding="fourier", dim_enc=64, num_layers=2)
CKW = dict(num_samples_coarse=16, num_samples_depth_guided=16)
R = 256


def _draws(seed, F, Rn):
    torch.manual_seed(seed)
    return torch.rand(F, Rn, 16).to(DEV), torch.rand(F, Rn, 16).to(DEV)


def _rel_dist(a: dict, b: dict, keys):
    num = sum(float(((a[k].cpu().double() - b[k].cpu().double()) ** 2).sum()) for k in keys)
    den = sum(float((b[k].cpu().double() ** 2).sum()) for k in keys)
    return (num / den) ** 0.5


def test_cfg0_100_iterations_follow_the_reference_trajectory():
    g = load_golden("g13_train_cfg0")
    pos, quat = g["pos"], g["quat"]
    p0 = split_prefix(g, "A::p0::")
    keys = [k for k in p0 if k != "_neus_sd"]
    r = make_renderer(FOURIER, CKW, 1, p0)
    r.set_field_poses(pos.to(DEV), quat.to(DEV))
    ids = torch.arange(1)
    losses, snaps = [], {}
    for it in range(scene.A_ITERS):
        t = scene.sphere_scene_batch(1, R, pos, scene.A_BATCH_SEED + it)
        u_c, u_g = _draws(scene.A_U_SEED + it, 1, R)
        losses.append(r.optimization_iteration(make_target(t, ids), u_c, u_g, update=True)["combined"].clone())
        if it + 1 in scene.A_CHECKPOINTS:
            snaps[it + 1] = {k: v.clone() for k, v in r._model.all_fields_params.items()}
    losses = torch.stack(losses).cpu()
    ref_l, ref_lb = g["A::a::losses"], g["A::b::losses"]
    close(losses[:10], ref_l[:10], rtol=2e-3, atol=1e-5)                        # resolvable part: the G7 bar
    report = []
    for c in scene.A_CHECKPOINTS:
        a, b = split_prefix(g, f"A::a{c}::"), split_prefix(g, f"A::b{c}::")
        d_ref, d_gpu = _rel_dist(b, a, keys), _rel_dist(snaps[c], a, keys)
        report.append((c, d_gpu, d_ref))
        # the kernels re-order every fp32 sum in every iteration (forward error vs the reference ~1e-6 relative, the
        # one-ulp run perturbs once by 1e-7): a factor of 30 on the reference's own divergence is that, not a bias
        assert d_gpu <= 30.0 * d_ref + 1e-6, report
    print("relative distance to the reference run (kernels, reference started one ulp apart):",
          ["it %d: %.2e / %.2e" % x for x in report])
    for k in keys:                                                               # after 10 iterations: plain tolerance
        close(snaps[10][k], g[f"A::a10::{k}"], rtol=2e-3, atol=5e-5)
    m = {k: r._optim_state[k]["exp_avg"] for k in keys}
    v = {k: r._optim_state[k]["exp_avg_sq"] for k in keys}
    d_ref100 = _rel_dist(split_prefix(g, "A::b100::"), split_prefix(g, "A::a100::"), keys)
    # the moments follow the gradients of the last ~10 (exp_avg) / ~1000 (exp_avg_sq) iterations
    assert _rel_dist(v, split_prefix(g, "A::v::"), keys) < 0.05
    assert _rel_dist(m, split_prefix(g, "A::m::"), keys) < max(1e3 * d_ref100, 0.5)
    tail = slice(scene.A_ITERS - 20, scene.A_ITERS)
    assert abs(float(losses[tail].mean() - ref_l[tail].mean())) < 0.1 * float(ref_l[tail].mean())
    assert abs(float(ref_lb[tail].mean() - ref_l[tail].mean())) < 0.1 * float(ref_l[tail].mean())   # the yardstick itself


def test_ensemble_psnr_within_a_tenth_of_a_db_of_the_reference():
    g = load_golden("g13_train_ensemble")
    F = scene.B_FIELDS
    pos, quat, phase = g["pos"], g["quat"], g["phase"]
    assert pos.shape[0] == F
    proto = split_prefix(g, "proto::")
    p0 = {k: scene.perturbed_init(v, F, k, 141) for k, v in proto.items()}
    chk = scene.checksum(p0)
    assert abs(chk - float(g["init_checksum"])) <= 1e-9 * chk, "the seeded initial parameters were not reproduced"
    r = make_renderer(FOURIER, CKW, F, p0)
    r.set_field_poses(pos.to(DEV), quat.to(DEV))
    ids = torch.arange(F)
    th = scene.sphere_scene_batch(F, scene.B_HELD_OUT_RAYS, pos, scene.B_HELD_OUT_SEED, phase=phase)
    hu_c, hu_g = _draws(scene.B_HELD_OUT_U_SEED, F, scene.B_HELD_OUT_RAYS)
    th_dev = {k: v.to(DEV) for k, v in th.items()}
    ids_dev = ids.to(DEV)
    losses, psnr, derr = [], [], []
    for it in range(scene.B_ITERS):
        t = scene.sphere_scene_batch(F, R, pos, scene.B_BATCH_SEED + it, phase=phase)
        u_c, u_g = _draws(scene.B_U_SEED + it, F, R)
        losses.append(r.optimization_iteration(make_target(t, ids), u_c, u_g, update=True)["combined"].clone())
        if it + 1 >= scene.B_EVAL_FROM and (it + 1) % scene.B_EVAL_EVERY == 0:
            with torch.no_grad():
                p = r.render_ijs(th_dev["ijs"], th_dev["c2ws"], None, field_ids=ids_dev, use_vmap=True, near_distances=th_dev["near"],
                                 far_distances=th_dev["far"], gt_distances=th_dev["gt"], u_coarse=hu_c, u_guided=hu_g)
            ps, de = scene.held_out_scores(p.rgbds, th)
            psnr.append(ps)
            derr.append(de)
    psnr, derr, losses = torch.stack(psnr), torch.stack(derr), torch.stack(losses).cpu()
    ref_psnr, ref_derr, ref_l = g["psnr"], g["depth_err"], g["losses"]
    assert psnr.shape == ref_psnr.shape
    close(losses[:5], ref_l[:5], rtol=2e-3, atol=1e-5)
    d_mean = float(psnr.mean() - ref_psnr.mean())
    se = float((psnr - ref_psnr).mean(0).std() / F ** 0.5)            # fields are independent: standard error of d_mean
    print("ensemble PSNR: kernels %.3f dB, reference %.3f dB, difference %+.3f dB (standard error %.3f); depth error "
          "%.4f / %.4f m; single values differ by %.2f dB rms"
          % (float(psnr.mean()), float(ref_psnr.mean()), d_mean, se, float(derr.mean()), float(ref_derr.mean()),
             float((psnr - ref_psnr).pow(2).mean().sqrt())))
    assert float(ref_psnr.mean()) > 20.0                               # the reference run did converge
    assert abs(d_mean) < 0.1, (d_mean, se)
    # every checkpoint's 96-field mean on its own: single values differ by ~2.5 dB rms between the two runs, so a
    # checkpoint mean by ~0.37 dB rms and the largest of 41 by < 1.25 dB (measured: 0.62)
    assert float((psnr.mean(1) - ref_psnr.mean(1)).abs().max()) < 1.25
    assert abs(float(derr.mean() - ref_derr.mean())) < 0.05 * float(ref_derr.mean())
    tail = slice(scene.B_ITERS - 100, scene.B_ITERS)
    assert abs(float(losses[tail].mean() - ref_l[tail].mean())) < 0.03 * float(ref_l[tail].mean())
