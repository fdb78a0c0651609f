"""Step time of the fused training iteration at the BASELINE config shapes (hipGraph replay): cfg4 = 16 fields x 512 rays x
(128 + 128) samples, M1 = 8 x 512 x (64 + 64); exact-fp32 MFMA against the auto mode.  python tools/config_shapes_bench.py"""
import sys, os, time, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from gpu_common import DEV, make_renderer, make_target, synth_target
from neural_graph_mapping_amd import _capi as K
FOURIER = dict(encoding="fourier", dim_enc=64, num_layers=2)
for F, R, sc, sg in ((16, 512, 128, 128), (8, 512, 64, 64)):
    for mm in ("f32", "auto", "f32", "auto"):
        r = make_renderer(FOURIER, dict(num_samples_coarse=sc, num_samples_depth_guided=sg, mlp_matmul=mm), F)
        pos, quat, t = synth_target(F, R, seed=2)
        r.set_field_poses(pos.to(DEV), quat.to(DEV))
        tgt = make_target(t, torch.arange(F))
        rep = r.capture_iteration(tgt, seed=3)
        for _ in range(30): rep()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(100): o = rep()
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 100
        print(F, R, sc + sg, mm, "%.4f ms" % (1e3 * dt), "%.3f G/s" % (F * R * (sc + sg) / dt / 1e9), "bwd variant", K.lib().ngm_debug_last_bwd_variant(), float(o["combined"]))
