"""Empty loss selections (rm.py:1787-1872): the reference takes `.mean()` of an empty tensor -- that loss TERM and `combined`
are NaN, while the term's gradient is empty and the gradients of the other terms stay finite.  One case per loss term with
exactly that term's selection empty: the fused step reports the reference's NaNs and its gradients equal the oracle's backward
of the NaN-valued loss (and those of the same batch with the empty term's weight set to zero)."""
import os
import sys

import pytest
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from gpu_common import (DEV, NRGBD, close, compare_losses, grad_close, kink_free_draws, make_renderer, make_target,  # noqa: E402
                        synth_target)
from oracle import ngm_oracle as O  # noqa: E402

pytestmark = pytest.mark.gpu

FOURIER = dict(encoding="fourier", dim_enc=64, num_layers=2)
WEIGHT_OF = dict(photometric=("photometric_weight", "depth_weight"), termination=("termination_weight",),
                 freespace=("freespace_weight",), tsdf=("tsdf_weight",))


def _case(which, seed=3, F=3, R=40, n_c=8, n_g=8):
    torch.manual_seed(seed)
    pos, quat, t = synth_target(F, R, seed=seed)
    if which == "freespace":                     # the surface right behind `near`: no sample lies more than tau in front of it
        t["gt"] = t["near"] + 0.05
    elif which == "tsdf":                        # depth beyond `far`: every sample is free space, none within tau of the surface
        t["gt"] = t["far"] + 1.0
    elif which == "freespace+tsdf":              # no depth at all (gt = 0 is "unavailable", rm.py:470-472)
        t["gt"] = torch.zeros_like(t["gt"])
    fs = O.FieldSpec(**FOURIER)
    rs = O.RenderSpec(num_samples_coarse=n_c, num_samples_depth_guided=n_g, termination_weight=0.3)
    params = O.init_params(fs, F, seed=seed, sigma=3.0)
    params["_linears.2.weight"] *= 2.0
    u_c, u_g = torch.rand(F, R, n_c), torch.rand(F, R, n_g)
    u_c, u_g, t = kink_free_draws(t, pos, quat, params, fs, rs, u_c, u_g, max_neutralised=0.3)
    if which == "photometric":                   # selection m = depth_mask & (term > 0.8), read by the photometric AND the depth term
        t["depth_mask"] = torch.zeros_like(t["depth_mask"])
    elif which == "termination":
        t["term_mask"] = torch.zeros_like(t["term_mask"])
    return pos, quat, t, fs, rs, params, u_c, u_g


@pytest.mark.parametrize("which", ["photometric", "termination", "freespace", "tsdf", "freespace+tsdf"])
def test_one_empty_loss_selection(which):
    pos, quat, t, fs, rs, params, u_c, u_g = _case(which)
    F = pos.shape[0]
    po = {k: v.clone().requires_grad_() for k, v in params.items()}
    pred = O.render_ijs(t["ijs"], t["c2ws"], NRGBD, pos, quat, po, fs, rs, t["near"], t["far"], t["gt"], u_c, u_g)
    ckw = dict(num_samples_coarse=rs.num_samples_coarse, num_samples_depth_guided=rs.num_samples_depth_guided, termination_weight=0.3)
    r = make_renderer(FOURIER, ckw, F, params)
    r.set_field_poses(pos.to(DEV), quat.to(DEV))
    tgt = make_target(t, torch.arange(F))
    res = r.optimization_iteration(tgt, u_c.to(DEV), u_g.to(DEV), update=False)
    loss, empty = compare_losses(res, pred, t, rs)           # NaN where the reference has NaN, the other terms' values equal
    assert empty == sorted(which.split("+")), (which, empty)
    assert bool(torch.isnan(res["combined"]))
    loss["combined"].backward()
    grads = {k: v.clone() for k, v in res["grads"].items()}
    for k in po:
        assert torch.isfinite(po[k].grad).all() and float(po[k].grad.abs().max()) > 0      # the reference keeps training
        assert torch.isfinite(grads[k]).all()
        grad_close(grads[k], po[k].grad, 2e-3, k)
    # the empty term contributes nothing: same gradients as with its weight set to zero (where `combined` is finite again)
    zero_w = {w: 0.0 for e in empty for w in WEIGHT_OF[e]}
    r0 = make_renderer(FOURIER, {**ckw, **zero_w}, F, params)
    r0.set_field_poses(pos.to(DEV), quat.to(DEV))
    res0 = r0.optimization_iteration(tgt, u_c.to(DEV), u_g.to(DEV), update=False)
    for k in grads:
        close(res0["grads"][k], grads[k], rtol=1e-6, atol=1e-9)
    if which in ("freespace", "tsdf", "freespace+tsdf"):
        assert torch.isfinite(res0["combined"])              # a zero-weight free-space / TSDF term does not exist (rm.py:624, 632)
    # an Adam step on such a batch leaves finite parameters
    r.optimization_iteration(tgt, u_c.to(DEV), u_g.to(DEV), update=True)
    for k, v in r._model.all_fields_params.items():
        assert torch.isfinite(v).all(), k
