import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from neural_graph_mapping_amd import _capi as K, ops
from oracle import ngm_oracle as O
kw = dict(encoding="fourier", dim_enc=32, num_layers=2); P = 257
torch.manual_seed(3)
F = 3
fs = O.FieldSpec(**kw); fc = K.field_cfg(**kw)
params = O.init_params(fs, F, seed=5, sigma=3.0)
pos, quat = torch.randn(F, 3), torch.nn.functional.normalize(torch.randn(F, 4), dim=-1)
q = pos[:, None] + 0.5 * torch.randn(F, P, 3)
d_out = torch.randn(F, P, 4)
for dt in (torch.float32, torch.float64):
    po = {k: v.to(dt).clone().requires_grad_() for k, v in params.items()}
    out_o = O.field_set_forward_vmap(q.to(dt), pos.to(dt), quat.to(dt), po, fs)
    (out_o * d_out.to(dt)).sum().backward()
    if dt == torch.float32: g32 = {k: v.grad.clone() for k, v in po.items()}
    else: g64 = {k: v.grad.float() for k, v in po.items()}
pg = {k: v.cuda().requires_grad_() for k, v in params.items()}
out = ops.field_eval(fc, pg, q.cuda(), pos.cuda(), quat.cuda())
(out * d_out.cuda()).sum().backward()
for k in pg:
    a = pg[k].grad.cpu()
    for name, ref in (("cpu32", g32[k]), ("cpu64", g64[k])):
        e = (a - ref).abs()
        print(k, name, "max err/max", float(e.max() / ref.abs().max()), "relL2", float((a - ref).norm() / ref.norm()))
    e = (a - g64[k]).abs() / g64[k].abs().max()
    bad = (e > 5e-4).nonzero()
    print("   bad entries (vs fp64):", bad[:12].tolist(), "count", len(bad))
    e2 = (g32[k] - g64[k]).abs() / g64[k].abs().max()
    print("   cpu32 vs cpu64 max:", float(e2.max()), "bad", int((e2 > 5e-4).sum()))
