"""Step time of the fused training iteration at the BASELINE config shapes (hipGraph replay): cfg4 = 16 fields x 512 rays x
(128 + 128) samples, M1 = 8 x 512 x (64 + 64), the default training batch 32 x 512 x (8 + 16) and a 4-field rank of it;
exact-fp32 MFMA against the auto mode.  python tools/config_shapes_bench.py"""
import sys, os, time, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from gpu_common import DEV, make_renderer, make_target, synth_target
from neural_graph_mapping_amd import _capi as K
FOURIER = dict(encoding="fourier", dim_enc=64, num_layers=2)
if os.environ.get("NGM_BENCH_WDT"):                 # weight storage type of BASELINE configs 1 / 4: bfloat16 | float16
    FOURIER["weight_dtype"] = os.environ["NGM_BENCH_WDT"]
import ctypes as C
for F, R, sc, sg in ((16, 512, 128, 128), (8, 512, 64, 64), (32, 512, 8, 16), (4, 512, 8, 16)):
    for mm in ("f32", "auto"):
        r = make_renderer(FOURIER, dict(num_samples_coarse=sc, num_samples_depth_guided=sg, mlp_matmul=mm), F)
        pos, quat, t = synth_target(F, R, seed=2)
        r.set_field_poses(pos.to(DEV), quat.to(DEV))
        tgt = make_target(t, torch.arange(F))
        rep = r.capture_iteration(tgt, seed=3)
        dt = 1e9
        for trial in range(3):                       # best of three: an idle GPU ramps its clocks for ~0.1 s
            for _ in range(50): rep()
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(100): o = rep()
            torch.cuda.synchronize(); dt = min(dt, (time.perf_counter() - t0) / 100)
        L = K.lib()
        L.ngm_profile_reset(); L.ngm_profile_enable(1)
        for _ in range(50): r.optimization_iteration(tgt, seed=7, update=True)
        torch.cuda.synchronize(); L.ngm_profile_enable(0)
        kern = {}
        for name, kid in K.KERNEL_IDS.items():
            ms, n = C.c_double(0), C.c_int64(0)
            L.ngm_profile_read(kid, C.byref(ms), C.byref(n))
            if n.value: kern[name] = round(1e3 * ms.value / n.value, 1)
        print(kern)
        print(F, R, sc + sg, mm, "%.4f ms" % (1e3 * dt), "%.3f G/s" % (F * R * (sc + sg) / dt / 1e9), "bwd variant", K.lib().ngm_debug_last_bwd_variant(), float(o["combined"]))
