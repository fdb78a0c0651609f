"""Developer tool: the Level-1 training pair (ops.field_eval under autograd: ngm_field_eval_fwd_train + ngm_field_eval_bwd_stash) on 64 fields x
65 536 points, for a rocprofv3 --kernel-trace --stats run (which kernels a backward really launches; round 6 found two 2 GB fills in it)."""
import sys, os, torch, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from neural_graph_mapping_amd import _capi as K, ops
dev = torch.device("cuda:0")
F, P = 64, 65536
fca = K.field_cfg(encoding="fourier", dim_enc=64, num_layers=2, matmul_mode="auto")
params = {n: (0.3 * torch.randn(F, *shp, device=dev)).requires_grad_() for n, shp in K.param_shapes(fca).items()}
pts = torch.rand(F, P, 3, device=dev); pos = torch.zeros(F, 3, device=dev); quat = torch.zeros(F, 4, device=dev); quat[:, 0] = 1
out = ops.field_eval(fca, params, pts, pos, quat)
go = torch.randn_like(out)
for _ in range(3):
    torch.autograd.grad(out, list(params.values()), go, retain_graph=True)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10):
    torch.autograd.grad(out, list(params.values()), go, retain_graph=True)
torch.cuda.synchronize()
print("bwd per call ms", (time.perf_counter() - t0) * 100)
