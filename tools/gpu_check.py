"""Torch-free GPU parity + timing harness (developer loop; the judged tests live in tests/).

Drives libngm_hip.so through the C ABI with hipMalloc'd buffers and compares against the golden
fixtures produced by the real reference (tests/golden/*.npz).  Starts in a second (no torch import),
so a gpurun call spends its time on kernels.

    python tools/gpu_check.py [sampler field quad train time] [--out gpurun_out/check.json]
"""
import ctypes as C
import json
import os
import sys
import time
import traceback

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from neural_graph_mapping_amd import _capi as K  # noqa: E402
from neural_graph_mapping_amd import hiprt as H  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")
REPORT = {}


def gold(name):
    z = np.load(os.path.join(GOLD, name + ".npz"))
    return {k: z[k] for k in z.files}


def pref(d, p):
    return {k[len(p):]: v for k, v in d.items() if k.startswith(p)}


def err(a, b):
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    d = np.abs(a - b)
    scale = max(np.abs(b).max(), 1e-30)
    return dict(max_abs=float(d.max()) if d.size else 0.0, max_rel_to_max=float(d.max() / scale) if d.size else 0.0,
                nan=int(np.isnan(a).sum()))


def record(name, **kw):
    REPORT[name] = kw
    print(f"[{name}] " + " ".join(f"{k}={v}" for k, v in kw.items()), flush=True)


def dev_params(fc, params, ids=None):
    """numpy dict (reference key names) -> device arrays + Params struct"""
    devs, ptrs, strides = {}, {}, {}
    for n in K.param_names(fc):
        a = np.ascontiguousarray(params[n], np.float32)
        devs[n] = H.to_dev(a)
        ptrs[n] = devs[n].ptr
        strides[n] = int(np.prod(a.shape[1:]))
    return devs, K.params_struct(fc, ptrs, strides)


def dev_grads(fc, F):
    devs, ptrs, strides = {}, {}, {}
    for n, shp in K.param_shapes(fc).items():
        devs[n] = H.DeviceArray((F,) + tuple(shp), np.float32)
        ptrs[n] = devs[n].ptr
        strides[n] = int(np.prod(shp))
    return devs, K.grads_struct(fc, ptrs, strides)


NRGBD = dict(fx=554.2562584220408, fy=554.2562584220408, cx=319.5, cy=239.5)


def lin_table(n):
    # torch.linspace(0,1,n+1) scalar formula in fp32
    step = np.float32(1.0) / np.float32(n)
    steps = n + 1
    half = steps // 2
    out = np.empty(steps, np.float32)
    for i in range(steps):
        out[i] = step * np.float32(i) if i < half else np.float32(1.0) - step * np.float32(steps - i - 1)
    return out


# --------------------------------------------------------------------------------------------------
def check_sampler():
    L = K.lib()
    g = gold("g3_sample_merged")
    F, R, n_c = g["u_coarse"].shape
    n_g = g["u_guided"].shape[-1]
    rc = K.render_cfg(num_samples_coarse=n_c, num_samples_guided=n_g, truncation_distance=float(g["rho"]), **NRGBD)
    d = {k: H.to_dev(g[k]) for k in ("ijs", "near", "far", "gt", "u_coarse", "u_guided")}
    c2w = H.to_dev(np.eye(4, dtype=np.float32))
    pos = H.to_dev(np.zeros((F, 3), np.float32))
    quat = H.to_dev(np.tile(np.array([1, 0, 0, 0], np.float32), (F, 1)))
    rays = K.Rays(F, R, d["ijs"].ptr, c2w.ptr, 0, 0, d["near"].ptr, d["far"].ptr, d["gt"].ptr, 0.0, 8.0, pos.ptr,
                  quat.ptr, d["u_coarse"].ptr, d["u_guided"].ptr, None, None, 0, 0)
    S = n_c + n_g
    pts = H.DeviceArray((F, R, S, 3))
    dist = H.DeviceArray((F, R, S))
    dirs = H.DeviceArray((F, R, 3))
    K.check(L.ngm_sample_rays(C.byref(rc), C.byref(rays), pts.ptr, dist.ptr, dirs.ptr, None), "sample_rays")
    t = dist.numpy()
    record("sampler_g3", bit_exact_t=bool(np.array_equal(t, g["distances"])), **err(t, g["distances"]),
           pts=err(pts.numpy(), g["points"])["max_abs"])
    g1 = gold("g1_directions")
    n = g1["ijs"].shape[0]
    rays1 = K.Rays(1, n, H.to_dev(g1["ijs"]).ptr, c2w.ptr, 0, 0, None, None, None, 0.0, 1.0, pos.ptr, quat.ptr,
                   H.to_dev(np.zeros((1, n, 1), np.float32)).ptr, None, None, None, 0, 0)
    rc1 = K.render_cfg(num_samples_coarse=1, num_samples_guided=0, **NRGBD)
    d1 = H.DeviceArray((n, 3))
    K.check(L.ngm_sample_rays(C.byref(rc1), C.byref(rays1), None, None, d1.ptr, None), "sample_rays dirs")
    record("sampler_g1_dirs", **err(d1.numpy(), g1["dirs"]))


def check_field():
    L = K.lib()
    for enc in ("fourier", "nerf"):
        g = gold(f"g4_field_forward_{enc}")
        fc = K.field_cfg(encoding=enc, dim_enc=64, num_layers=2, num_octaves=8)
        params = pref(g, "p::")
        devs, ps = dev_params(fc, params)
        F, P, _ = g["query"].shape
        q, pos, quat = H.to_dev(g["query"]), H.to_dev(g["pos"]), H.to_dev(g["quat"])
        out = H.DeviceArray((F, P, 4))
        K.check(L.ngm_field_eval_fwd(C.byref(fc), C.byref(ps), F, P, q.ptr, pos.ptr, quat.ptr, out.ptr, None),
                "field_eval_fwd")
        record(f"field_fwd_{enc}", **err(out.numpy(), g["out"]))


def check_quad():
    L = K.lib()
    for mode in ("nrgbd", "occupancy", "density", "neus"):
        for S in (2, 24, 128):
            g = gold(f"g5_quadrature_{mode}_S{S}")
            rc = K.render_cfg(geometry_mode=mode, geometry_factor=float(g["geometry_factor"]))
            lead = g["geoms"].shape[:-1]
            N = int(np.prod(lead))
            S_eff = S - 1 if mode in ("density", "neus") else S
            isds = None
            if "isds" in g:
                isds = H.to_dev(np.broadcast_to(g["isds"], lead + (1,)).reshape(N).astype(np.float32))
            col, geo, dis, dep = (H.to_dev(g[k].reshape((N, S) + g[k].shape[len(lead) + 1:]))
                                  for k in ("colors", "geoms", "dists", "depths"))
            Cc, D, Cv, Dv, T, W = (H.DeviceArray(s) for s in ((N, 3), (N,), (N, 3), (N,), (N,), (N, S_eff)))
            K.check(L.ngm_composite_fwd(C.byref(rc), N, S, col.ptr, geo.ptr, dis.ptr, dep.ptr, H.ptr(isds), Cc.ptr,
                                        D.ptr, Cv.ptr, Dv.ptr, T.ptr, W.ptr, None), "composite_fwd")
            e = {k: err(v.numpy().reshape(g[n].shape), g[n])["max_abs"]
                 for k, v, n in (("C", Cc, "C"), ("D", D, "D"), ("Cv", Cv, "Cv"), ("Dv", Dv, "Dv"), ("term", T, "term"),
                                 ("w", W, "w"))}
            record(f"quad_{mode}_S{S}", **e)


TRAIN_CASES = {
    "g6_train_cfg0": (dict(encoding="fourier", dim_enc=64, num_layers=2), dict(num_samples_coarse=16, num_samples_guided=16)),
    "g6_train_3field": (dict(encoding="fourier", dim_enc=64, num_layers=2),
                        dict(num_samples_coarse=8, num_samples_guided=16, w_termination=0.5)),
    "g6_train_nerf_l1": (dict(encoding="nerf", num_octaves=8, num_layers=1), dict(num_samples_coarse=8, num_samples_guided=8)),
}


def run_train_case(name, fckw, rckw, g=None, check=True):
    L = K.lib()
    g = g or gold(name)
    fc = K.field_cfg(**fckw)
    rc = K.render_cfg(**rckw, **NRGBD)
    params = {k: v for k, v in pref(g, "p::").items() if k != "_neus_sd"}
    t = pref(g, "t::")
    F, R = t["near"].shape
    S = rc.num_samples_coarse + rc.num_samples_guided
    pdev, ps = dev_params(fc, params)
    gdev, gs = dev_grads(fc, F)
    d = {k: H.to_dev(t[k]) for k in ("ijs", "c2ws", "near", "far", "gt", "rgbds", "term_probs")}
    dm = H.to_dev(t["depth_mask"].astype(np.uint8))
    tm = H.to_dev(t["term_mask"].astype(np.uint8))
    pos, quat = H.to_dev(g["pos"]), H.to_dev(g["quat"])
    uc, ug = H.to_dev(g["u_coarse"]), H.to_dev(g["u_guided"])
    lc, lg = H.to_dev(lin_table(rc.num_samples_coarse)), H.to_dev(lin_table(rc.num_samples_guided))
    rays = K.Rays(F, R, d["ijs"].ptr, d["c2ws"].ptr, 1, 0, d["near"].ptr, d["far"].ptr, d["gt"].ptr, 0.0, 8.0,
                  pos.ptr, quat.ptr, uc.ptr, ug.ptr, lc.ptr, lg.ptr, 0, 0)
    tg = K.Targets(d["rgbds"].ptr, dm.ptr, tm.ptr, d["term_probs"].ptr)
    pr = {k: H.DeviceArray(s) for k, s in (("rgbds", (F, R, 4)), ("color_vars", (F, R, 3)), ("depth_vars", (F, R)),
                                            ("term_probs", (F, R)))}
    pred = K.Prediction(pr["rgbds"].ptr, pr["color_vars"].ptr, pr["depth_vars"].ptr, pr["term_probs"].ptr)
    wsb = L.ngm_render_workspace(C.byref(fc), C.byref(rc), F, R, 1)
    ws = H.DeviceArray((wsb,), np.uint8)
    sums = H.DeviceArray((16,))
    lout = H.DeviceArray((8,))
    K.check(L.ngm_render_fwd(C.byref(fc), C.byref(rc), C.byref(ps), C.byref(rays), C.byref(tg), C.byref(pred), sums.ptr,
                             ws.ptr, wsb, None), "render_fwd")
    H.synchronize()
    geo, dis = H.DeviceArray((F, R, S)), H.DeviceArray((F, R, S))
    K.check(L.ngm_render_read_samples(C.byref(fc), C.byref(rc), F, R, rc.num_samples_coarse + rc.num_samples_guided, ws.ptr, geo.ptr, dis.ptr, None), "read_samples")
    geo_np, dis_np = geo.numpy(), dis.numpy()
    K.check(L.ngm_render_bwd(C.byref(fc), C.byref(rc), C.byref(ps), C.byref(rays), C.byref(tg), C.byref(pred), sums.ptr,
                             C.byref(gs), lout.ptr, ws.ptr, wsb, None), "render_bwd")
    H.synchronize()
    res = dict(pred={k: v.numpy() for k, v in pr.items()}, sums=sums.numpy(), loss=lout.numpy(),
               grads={k: v.numpy() for k, v in gdev.items()}, geoms=geo_np, dists=dis_np)
    if check:
        rep = {}
        for k, ref in (("rgbds", "pred_rgbds"), ("color_vars", "pred_color_vars"), ("depth_vars", "pred_depth_vars"),
                       ("term_probs", "pred_term_probs")):
            rep[k] = err(res["pred"][k], g[ref])["max_abs"]
        # compacted vectors of the Prediction (rm.py:624-639) rebuilt from the stash
        tau = rc.truncation_distance
        gt = t["gt"][..., None]
        fsm = dis_np < (gt - tau) * (gt != 0)
        tsm = (np.abs(gt - dis_np) < tau) & (gt != 0)
        rep["n_fs"] = (int(fsm.sum()), int(g["pred_freespace"].shape[0]))
        rep["n_ts"] = (int(tsm.sum()), int(g["pred_tsdf"].shape[0]))
        if fsm.sum() == g["pred_freespace"].shape[0]:
            rep["fs_vec"] = err(geo_np[fsm] * tau, g["pred_freespace"])["max_abs"]
        if tsm.sum() == g["pred_tsdf"].shape[0]:
            rep["ts_vec"] = err(geo_np[tsm] * tau - (np.broadcast_to(gt, dis_np.shape) - dis_np)[tsm], g["pred_tsdf"])["max_abs"]
        ref_loss = pref(g, "loss::")
        rep["loss_combined"] = (float(res["loss"][0]), float(ref_loss["combined"]))
        rep["loss_terms"] = [float(x) for x in res["loss"][1:6]]
        rep["ref_terms"] = {k: float(v) for k, v in ref_loss.items() if k != "combined"}
        rep["sums"] = [float(x) for x in res["sums"][:10]]
        ref_g = pref(g, "g::")
        for k, v in ref_g.items():
            rep["grad " + k] = round(err(res["grads"][k], v)["max_rel_to_max"], 7)
        record(name, **rep)
    return res


def check_train():
    for name, (fckw, rckw) in TRAIN_CASES.items():
        try:
            run_train_case(name, fckw, rckw)
        except Exception:
            traceback.print_exc()
            record(name, error=traceback.format_exc()[-600:])


# --------------------------------------------------------------------------------------------------
def synth_batch(F, R, S_c, S_g, seed=0):
    """Synthetic M1-style batch built with numpy only."""
    rng = np.random.default_rng(seed)
    pos = (rng.standard_normal((F, 3)) * 0.5).astype(np.float32)
    quat = rng.standard_normal((F, 4)).astype(np.float32)
    quat /= np.linalg.norm(quat, axis=-1, keepdims=True)
    ijs = np.stack([rng.integers(0, 480, (F, R)), rng.integers(0, 640, (F, R))], -1).astype(np.int64)
    eye_dir = rng.standard_normal((F, R, 3))
    eye_dir /= np.linalg.norm(eye_dir, axis=-1, keepdims=True)
    eye = pos[:, None] + eye_dir * (2.0 + rng.random((F, R, 1)))
    tgt = pos[:, None] + 0.3 * rng.standard_normal((F, R, 3))
    fwd = tgt - eye
    fwd /= np.linalg.norm(fwd, axis=-1, keepdims=True)
    up = np.broadcast_to(np.array([0.0, 1.0, 0.0]), fwd.shape)
    right = np.cross(fwd, up)
    right /= np.linalg.norm(right, axis=-1, keepdims=True)
    up2 = np.cross(right, fwd)
    c2w = np.tile(np.eye(4), (F, R, 1, 1))
    c2w[..., :3, 0], c2w[..., :3, 1], c2w[..., :3, 2], c2w[..., :3, 3] = right, up2, -fwd, eye
    c2w = c2w.astype(np.float32)
    dx = (ijs[..., 1] - 319.5) / 554.2562584220408
    dy = -(ijs[..., 0] - 239.5) / 554.2562584220408
    d = np.stack([dx, dy, -np.ones_like(dx)], -1)
    d /= np.linalg.norm(d, axis=-1, keepdims=True)
    Rm, tr = c2w[..., :3, :3].astype(np.float64), c2w[..., :3, 3].astype(np.float64)
    pos_c = np.einsum("...kd,...k->...d", Rm, pos[:, None] - tr)
    center = (pos_c * d).sum(-1)
    near = np.clip(center - 1.0, 0, None).astype(np.float32)
    far = np.clip(center + 1.0, 0, None).astype(np.float32)
    gt = (near + (far - near) * (0.1 + 0.8 * rng.random((F, R)))).astype(np.float32)
    gt[rng.random((F, R)) < 0.1] = 0.0
    rgbds = np.concatenate([rng.random((F, R, 3)), (gt * np.abs(d[..., 2]))[..., None]], -1).astype(np.float32)
    depth_mask = ((gt > near) & (gt < far) & (gt != 0)).astype(np.uint8)
    return dict(pos=pos, quat=quat, ijs=ijs, c2ws=c2w, near=near, far=far, gt=gt, rgbds=rgbds, depth_mask=depth_mask)


def init_params_np(fc, F, seed=0, sigma=4.0):
    rng = np.random.default_rng(seed)
    out = {}
    for n, shp in K.param_shapes(fc).items():
        if n == "_encoding._linear.weight":
            out[n] = (rng.standard_normal((F,) + shp) * sigma).astype(np.float32)
        elif n == "_encoding.lattice_values":
            out[n] = (rng.standard_normal((F,) + shp) * 0.1).astype(np.float32)
        elif n == "_encoding.random_shift_per_level":
            out[n] = (rng.standard_normal((F,) + shp) * 10).astype(np.float32)
        else:
            fan_in = shp[1] if len(shp) == 2 else K.param_shapes(fc)[n.replace("bias", "weight")][1]
            b = 1.0 / np.sqrt(fan_in)
            out[n] = ((rng.random((F,) + shp) * 2 - 1) * b).astype(np.float32)
    out[f"_linears.{fc.num_layers}.weight"] *= 2.0
    return out


def check_time(F=8, R=512, S_c=64, S_g=64, iters=10, hash_enc=False):
    L = K.lib()
    fc = (K.field_cfg(encoding="permuto", num_layers=1, matmul_mode=os.environ.get("NGM_MATMUL", "f32")) if hash_enc
          else K.field_cfg(encoding="fourier", dim_enc=64, num_layers=2, matmul_mode=os.environ.get("NGM_MATMUL", "f32")))
    rc = K.render_cfg(num_samples_coarse=S_c, num_samples_guided=S_g, **NRGBD)
    b = synth_batch(F, R, S_c, S_g)
    params = init_params_np(fc, F)
    pdev, ps = dev_params(fc, params)
    gdev, gs = dev_grads(fc, F)
    d = {k: H.to_dev(b[k]) for k in ("ijs", "c2ws", "near", "far", "gt", "rgbds", "pos", "quat", "depth_mask")}
    rays = K.Rays(F, R, d["ijs"].ptr, d["c2ws"].ptr, 1, 0, d["near"].ptr, d["far"].ptr, d["gt"].ptr, 0.0, 8.0,
                  d["pos"].ptr, d["quat"].ptr, None, None, None, None, 1234, 0)
    tg = K.Targets(d["rgbds"].ptr, d["depth_mask"].ptr, None, None)
    pr = {k: H.DeviceArray(s) for k, s in (("rgbds", (F, R, 4)), ("color_vars", (F, R, 3)), ("depth_vars", (F, R)),
                                            ("term_probs", (F, R)))}
    pred = K.Prediction(pr["rgbds"].ptr, pr["color_vars"].ptr, pr["depth_vars"].ptr, pr["term_probs"].ptr)
    wsb = L.ngm_render_workspace(C.byref(fc), C.byref(rc), F, R, 1)
    ws = H.DeviceArray((wsb,), np.uint8)
    sums, lout = H.DeviceArray((16,)), H.DeviceArray((8,))

    def fwd():
        K.check(L.ngm_render_fwd(C.byref(fc), C.byref(rc), C.byref(ps), C.byref(rays), C.byref(tg), C.byref(pred),
                                 sums.ptr, ws.ptr, wsb, None), "render_fwd")

    def bwd():
        K.check(L.ngm_render_bwd(C.byref(fc), C.byref(rc), C.byref(ps), C.byref(rays), C.byref(tg), C.byref(pred),
                                 sums.ptr, C.byref(gs), lout.ptr, ws.ptr, wsb, None), "render_bwd")

    for _ in range(2):
        fwd(); bwd()
    H.synchronize()
    e = [H.Event() for _ in range(3)]
    tf = tb = 0.0
    if os.environ.get("NGM_TWICE"):      # experiment: the same kernel twice in a row -- what a warm instruction cache / warm L2 is worth
        ee = [H.Event() for _ in range(5)]
        acc = [0.0] * 4
        for _ in range(iters):
            ee[0].record(); fwd(); ee[1].record(); fwd(); ee[2].record(); bwd(); ee[3].record(); bwd(); ee[4].record()
            ee[4].synchronize()
            for i in range(4):
                acc[i] += ee[i].elapsed_ms(ee[i + 1])
        print(f"[twice F={F} R={R} S={S_c + S_g}] fwd first/second us: {acc[0] / iters * 1e3:.1f} / {acc[1] / iters * 1e3:.1f}   "
              f"bwd (+reduce) first/second us: {acc[2] / iters * 1e3:.1f} / {acc[3] / iters * 1e3:.1f}")
    for _ in range(iters):
        e[0].record(); fwd(); e[1].record(); bwd(); e[2].record()
        e[2].synchronize()
        tf += e[0].elapsed_ms(e[1]); tb += e[1].elapsed_ms(e[2])
    n = F * R * (S_c + S_g)
    tf /= iters; tb /= iters
    if os.environ.get("NGM_PHASE_TIMING"):
        buf = (C.c_ulonglong * (16 + 8 * 64))()
        L.ngm_debug_phase_cycles.argtypes = [C.c_void_p]
        if L.ngm_debug_phase_cycles(buf) == 0:
            names = ["prologue", "inputs", "encode", "fwd", "outlayer", "stage+colsum", "wgrad", "dgrad", "encgrad",
                     "relumask", "-", "epilogue", "TOTAL"]
            if L.ngm_debug_last_bwd_variant() == 3:
                names = ["prologue", "inputs", "dma_wait", "encode", "outlayer", "dma_issue", "wgrad1", "dgrad1", "dgrad0",
                         "mask+store", "wgrad0", "epilogue", "TOTAL"]
            tot = buf[12] or 1
            print("phase cycles (wave 0, block 0):", {n: (int(buf[i]), round(100 * buf[i] / tot, 1)) for i, n in enumerate(names)})
            tl = ["entry", "prologue", "tile", "loopend", "barrier", "end"]
            for w in range(8):
                ev = [(int(buf[16 + 64 * w + i]) >> 48, int(buf[16 + 64 * w + i]) & ((1 << 48) - 1)) for i in range(64)]
                ev = [(k, c) for k, c in ev if c]
                if ev:
                    print(f"  bwd wave {w}: " + " ".join(f"{tl[k][:4]}@{c // 100 / 10:.1f}k" for k, c in ev))
        L.ngm_debug_fwd_phase_cycles.argtypes = [C.c_void_p]
        buf = (C.c_ulonglong * (16 + 8 * 64))()
        if L.ngm_debug_fwd_phase_cycles(buf) == 0:
            names = ["prologue", "raysetup", "sampler", "stephead", "encode", "layers", "actstore", "outlayer", "composite",
                     "variance", "rayout", "blockreduce", "-", "-", "TOTAL", "REALTIME_100MHz"]
            tot = buf[14] or 1
            print("forward phase cycles (wave 0, middle block):",
                  {n: (int(buf[i]), round(100 * buf[i] / tot, 1)) for i, n in enumerate(names)},
                  "counter GHz:", round(buf[14] / max(1, buf[15]) * 0.1, 3))
            for w in range(8):
                ev = [(int(buf[16 + 64 * w + i]) >> 48, int(buf[16 + 64 * w + i]) & ((1 << 48) - 1)) for i in range(64)]
                ev = [(k, c) for k, c in ev if c]
                print(f"  wave {w}: " + " ".join(f"{names[k][:4]}@{c // 100 / 10:.1f}k" for k, c in ev))
    gn = {k: float(np.abs(v.numpy()).max()) for k, v in gdev.items()}
    record(f"time_{'hash' if hash_enc else 'fourier'}_F{F}_R{R}_S{S_c + S_g}", fwd_ms=round(tf, 4), bwd_ms=round(tb, 4),
           Msamples_per_s=round(n / (tf + tb) / 1e3, 1), loss=float(lout.numpy()[0]),
           sums=[float(x) for x in sums.numpy()[:8]], grad_absmax=gn,
           finite=bool(all(np.isfinite(v.numpy()).all() for v in gdev.values())))


def check_sampler_random(n_c=7, n_g=5, F=3, R=37, seed=705):
    """debug: GPU rank-merge vs numpy sort on random rays"""
    f32 = np.float32
    L = K.lib()
    rng = np.random.default_rng(seed)
    near = (rng.random((F, R)) * 2).astype(f32)
    far = (near + 0.5 + rng.random((F, R)) * 3).astype(f32)
    far[0, 0] = near[0, 0]
    gt = (near + (far - near) * rng.random((F, R))).astype(f32)
    gt[0, 1] = 0.0; gt[1, 2] = far[1, 2] + 1.0; gt[2, 3] = near[2, 3] * 0.5
    u_c, u_g = rng.random((F, R, n_c)).astype(f32), rng.random((F, R, n_g)).astype(f32)
    lc, lg = lin_table(n_c), lin_table(n_g)

    def strat(ne, fa, n, u, lin):
        span = (fa - ne).astype(f32)
        delta = (span / f32(n)).astype(f32)
        b = (lin[None, None, :-1] * span[..., None]).astype(f32)
        return (((delta[..., None] * u).astype(f32) + b).astype(f32) + ne[..., None]).astype(f32)
    inv = (gt == 0) | (near > gt) | (far < gt)
    gn = np.where(inv, near, (gt - f32(0.1)).astype(f32)).astype(f32)
    gf = np.where(inv, far, (gt + f32(0.1)).astype(f32)).astype(f32)
    ref = np.sort(np.concatenate([strat(near, far, n_c, u_c, lc), strat(gn, gf, n_g, u_g, lg)], -1), -1)
    rc = K.render_cfg(num_samples_coarse=n_c, num_samples_guided=n_g, **NRGBD)
    ijs = H.to_dev(np.zeros((F, R, 2), np.int64))
    keep = [H.to_dev(x) for x in (near, far, gt, u_c, u_g, lc, lg, np.eye(4, dtype=f32), np.zeros((F, 3), f32),
                                  np.tile(np.array([1, 0, 0, 0], f32), (F, 1)))]
    rays = K.Rays(F, R, ijs.ptr, keep[7].ptr, 0, 0, keep[0].ptr, keep[1].ptr, keep[2].ptr, 0.0, 8.0, keep[8].ptr,
                  keep[9].ptr, keep[3].ptr, keep[4].ptr, keep[5].ptr, keep[6].ptr, 0, 0)
    dist = H.DeviceArray((F, R, n_c + n_g))
    dist.fill_bytes(0xFF)
    K.check(L.ngm_sample_rays(C.byref(rc), C.byref(rays), None, dist.ptr, None, None), "sample_rays")
    t = dist.numpy()
    bad = np.argwhere((t != ref).any(-1))
    record(f"sampler_random_{n_c}_{n_g}", mismatching_rays=len(bad), total=F * R)
    for f, r in bad[:4]:
        print(" ray", f, r, "near/far/gt", near[f, r], far[f, r], gt[f, r], "inv", inv[f, r])
        print("  gpu", t[f, r]); print("  ref", ref[f, r])


if __name__ == "__main__":
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    out = "gpurun_out/check.json"
    for a in sys.argv[1:]:
        if a.startswith("--out="):
            out = a.split("=", 1)[1]
    todo = args or ["sampler", "field", "quad", "train", "time"]
    ncu = C.c_int(0)
    name = C.create_string_buffer(128)
    print("abi", K.lib().ngm_abi_version(), "devices", H.device_count())
    if K.lib().ngm_device_info(C.byref(ncu), name, 128) == 0:
        print("device:", name.value.decode(), "CUs:", ncu.value)
    t0 = time.time()
    for c in todo:
        try:
            {"sampler": check_sampler, "field": check_field, "quad": check_quad, "train": check_train,
             "time": check_time,
             # other shapes of SURVEY 8d: cfg4-like 8192 rays x 256 samples, default-config-like 32 x 512 x 24, one big field
             "time_shapes": lambda: (check_time(F=16, R=512, S_c=128, S_g=128, iters=5), check_time(F=32, R=512, S_c=8, S_g=16),
                                     check_time(F=1, R=4096, S_c=64, S_g=64), check_time(F=64, R=64, S_c=64, S_g=64)), "time_hash": lambda: (check_time(hash_enc=True), check_time(F=32, R=512, S_c=8, S_g=16, hash_enc=True)),
             "time_hash_m1": lambda: check_time(hash_enc=True),
             "time_hash_default": lambda: check_time(F=32, R=512, S_c=8, S_g=16, hash_enc=True),
             # what a rank with few active fields pays: the fixed cost
             "time_small": lambda: (check_time(F=1, R=512, S_c=8, S_g=16, iters=20), check_time(F=4, R=512, S_c=8, S_g=16, iters=20),
                                    check_time(F=4, R=512, S_c=64, S_g=64, iters=20)),
             "sampler_random": lambda: (check_sampler_random(7, 5), check_sampler_random(64, 64),
                                                             check_sampler_random(4, 4))}[c]()
        except Exception:
            traceback.print_exc()
            record(c, error=traceback.format_exc()[-800:])
    print(f"done in {time.time() - t0:.1f}s")
    os.makedirs(os.path.dirname(out) or ".", exist_ok=True)
    with open(out, "w") as fh:
        json.dump(REPORT, fh, indent=1, default=str)
