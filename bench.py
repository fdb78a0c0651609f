"""Headline benchmark: ray-samples/s of the per-field NeRF train step on MI355X.

    python bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json metric, SURVEY 8d "M1"): synthetic 4096-ray x 128-sample batch per GPU --
F = 8 fields x R = 512 rays, S = 64 coarse + 64 depth-guided samples, Fourier(64, raw) + 2x64 MLP,
nrgbd compositing, default loss weights, NRGBD intrinsics, 10 % missing depth.  One "step" = one
pass of the hot path over one batch: fused forward + losses, loss all-reduce (N > 1), fused backward
w.r.t. every field parameter, sparse per-field Adam.  Jitter is drawn in-kernel (Philox), inputs are
resident in HBM before the timed region.  Weak scaling: every rank owns its own 8 fields (field-per-
GPU sharding); the only collective is the 64-byte loss-sum all-reduce (RCCL over xGMI).

Extra objects on the JSON line:
  roofline     -- dominant kernel (MFMA backward) : algorithmic flops / HIP-event time on the launch stream
  cpu_baseline -- the CPU oracle (oracle/ngm_oracle.py, a restatement pinned to the reference) timed on
                  this box's host cores on a bounded sample of the same workload (rank 0, N = 1 only)
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

F_PER_GPU, R, S_C, S_G = 8, 512, 64, 64
D_ENC, N_LAYERS = 64, 2
FLOP_FWD = 2 * (64 * 64 + 64 * 64 + 64 * 4)      # 16 896 flop / sample
FLOP_BWD = 2 * FLOP_FWD                          # wgrad + dgrad: 33 792 flop / sample
PEAK_F32_MFMA_TF = 157.3                         # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32
SPIN_UP = 200                                    # untimed iterations before the warm-up (clock ramp of an idle GPU, ~60 ms)
DTYPE_LABEL = {"f32": "f32",
               "bf16x3": "f32 via 3xbf16 split, fp32 accumulate (hidden-layer matrix products, forward and backward); "
                         "everything else f32"}
PEAK_HBM_GBS = 8000.0                            # MI355X_MICROARCH.md: HBM3E ~8 TB/s


def synth_target(F, Rn, seed, field_offset=0):
    """Synthetic Target in the spirit of _sample_target_mv (rm.py:1383-1459); CPU tensors."""
    from neural_graph_mapping_amd.renderer import Target
    g = torch.Generator().manual_seed(seed)
    pos = 0.5 * torch.randn(F, 3, generator=g)
    quat = torch.nn.functional.normalize(torch.randn(F, 4, generator=g), dim=-1)
    ijs = torch.stack([torch.randint(0, 480, (F, Rn), generator=g), torch.randint(0, 640, (F, Rn), generator=g)], -1)
    eye = pos[:, None] + torch.nn.functional.normalize(torch.randn(F, Rn, 3, generator=g), dim=-1) * (
        2 + torch.rand(F, Rn, 1, generator=g))
    fwd = torch.nn.functional.normalize(pos[:, None] + 0.3 * torch.randn(F, Rn, 3, generator=g) - eye, dim=-1)
    right = torch.nn.functional.normalize(torch.linalg.cross(fwd, torch.tensor([0.0, 1.0, 0.0]).expand_as(fwd)), dim=-1)
    c2w = torch.eye(4).repeat(F, Rn, 1, 1)
    c2w[..., :3, 0], c2w[..., :3, 1], c2w[..., :3, 2], c2w[..., :3, 3] = right, torch.linalg.cross(right, fwd), -fwd, eye
    dx = (ijs[..., 1] - 319.5) / 554.2562584220408
    dy = -(ijs[..., 0] - 239.5) / 554.2562584220408
    d = torch.nn.functional.normalize(torch.stack([dx, dy, -torch.ones_like(dx)], -1), dim=-1)
    pos_c = torch.einsum("...kd,...k->...d", c2w[..., :3, :3], pos[:, None] - c2w[..., :3, 3])
    center = (pos_c * d).sum(-1)
    near, far = (center - 1).clamp_min(0), (center + 1).clamp_min(0)
    gt = near + (far - near) * (0.1 + 0.8 * torch.rand(F, Rn, generator=g))
    gt[torch.rand(F, Rn, generator=g) < 0.1] = 0.0
    rgbds = torch.cat([torch.rand(F, Rn, 3, generator=g), (gt * d[..., 2].abs())[..., None]], -1)
    dm = (gt > near) & (gt < far) & (gt != 0)
    tgt = Target(ijs=ijs, c2ws=c2w, near_distances=near, far_distances=far, gt_distances=gt,
                 field_ids=torch.arange(F) + field_offset, rgbds=rgbds, rgb_mask=dm, depth_mask=dm,
                 term_probs=(gt < far).float(), term_mask=(gt > near) & (gt != 0))
    return pos, quat, tgt


def build_renderer(device, num_fields, variant="fourier", s_c=None, s_g=None, matmul="auto", hash_atomics="exact"):
    from neural_graph_mapping_amd import models as M
    from neural_graph_mapping_amd import renderer as Rr
    torch.manual_seed(0)
    if variant == "hash":      # V-Hash of SURVEY 8d = the reference's default network (config/neural_graph_map.yaml:6-20)
        enc = dict(encoding_type="neural_graph_mapping.positional_encodings.PermutohedralEncoding",
                   encoding_kwargs=dict(pos_dim=3, log2_hashmap_size=12, nr_levels=16, nr_feat_per_level=2, coarsest_scale=1.0,
                                        finest_scale=0.0001, appply_random_shift_per_level=True, concat_points=False,
                                        concat_points_scaling=1.0), num_layers=1)
    else:
        enc = dict(encoding_type="neural_graph_mapping.positional_encodings.PositionalEncodingFourier",
                   encoding_kwargs=dict(dim_in=3, dim_out=D_ENC, mu=0.0, sigma=4.0, raw_coords=True), num_layers=N_LAYERS)
    model = M.NeuralFieldSet(dim_points=3, field_type="neural_graph_mapping.models.NeuralField", field_kwargs=dict(
        **enc, dim_out=4, dim_mlp_out=None, skip_mode="no", initial_geometry_bias=0.0, neus_initial_sd=1.0), num_knn=2,
        distance_factor=10.0, outside_value=1.0, field_radius=1.0, scale_mode="unit_cube").to(device)
    cfg = dict(geometry_mode="nrgbd", geometry_factor=20.0, color_factor=1.0, truncation_distance=0.1, field_radius=1.0,
               termination_weight=0.0, photometric_weight=1.0, photometric_loss="l1", depth_weight=1.0, depth_loss="huber",
               freespace_weight=40.0, tsdf_weight=50.0,
               learning_rate=1e-3, adam_eps=1e-15, adam_weight_decay=1e-5, near_distance=0.0, far_distance=8.0,
               num_samples_coarse=S_C if s_c is None else s_c, num_samples_depth_guided=S_G if s_g is None else s_g,
               mlp_matmul=matmul, hash_grad_atomics=hash_atomics)
    cam = Rr.Camera(640, 480, 554.2562584220408, 554.2562584220408, 319.5, 239.5, pixel_center=0.0)
    r = Rr.NeuralGraphRenderer(model, cam, cfg, device=device)
    r.add_fields(num_fields)
    with torch.no_grad():          # de-correlate the cloned prototype so fields differ (random-init weights)
        for v in model.all_fields_params.values():
            if v.dim() > 1:
                v.add_(0.05 * torch.randn_like(v))
    return r


def cpu_baseline(budget_s=12.0, F=F_PER_GPU):
    """Oracle (CPU restatement of the reference path, pinned to the real reference by the committed fixtures) on the
    SAME batch as the GPU line: F = 8 fields x 512 rays x 128 samples per step, best thread count of the box; the
    1-field sample of earlier rounds rides along as `one_field` (cpu_baseline(F=1))."""
    from oracle import ngm_oracle as O
    fs = O.FieldSpec(encoding="fourier", dim_enc=D_ENC, num_layers=N_LAYERS)
    rs = O.RenderSpec(num_samples_coarse=S_C, num_samples_depth_guided=S_G, geometry_factor=20.0)
    cam = O.CameraSpec(640, 480, 554.2562584220408, 554.2562584220408, 319.5, 239.5)
    pos, quat, t = synth_target(F, R, seed=123)
    params = {k: v.requires_grad_() for k, v in O.init_params(fs, F, seed=1).items()}
    g = torch.Generator().manual_seed(5)

    def step():
        u_c, u_g = torch.rand(F, R, S_C, generator=g), torch.rand(F, R, S_G, generator=g)
        pred = O.render_ijs(t.ijs, t.c2ws, cam, pos, quat, params, fs, rs, t.near_distances, t.far_distances,
                            t.gt_distances, u_c, u_g)
        loss = O.compute_losses(pred, t.rgbds, t.depth_mask, t.term_mask, t.term_probs, rs)
        for p in params.values():
            p.grad = None
        loss["combined"].backward()

    # pick the thread count that serves the CPU best (small batched GEMMs scale poorly to all cores)
    step()
    best = None
    for nt in sorted({torch.get_num_threads(), 64, 32, 16, 8}):
        if nt > (os.cpu_count() or nt):
            continue
        torch.set_num_threads(nt)
        step()
        t0 = time.perf_counter()
        step()
        one = time.perf_counter() - t0
        if best is None or one < best[0]:
            best = (one, nt)
    one, nt = best
    torch.set_num_threads(nt)
    n = max(2, min(200, int(budget_s / max(one, 1e-3))))
    t0 = time.perf_counter()
    for _ in range(n):
        step()
    dt = time.perf_counter() - t0
    res = dict(value=F * R * (S_C + S_G) * n / dt, unit="ray-samples/s", cores=torch.get_num_threads(), kind="port",
               sample=f"{n} train steps (fwd+loss+bwd) of {F} field(s) x {R} rays x {S_C + S_G} samples"
                      + (" = the GPU line's batch" if F == F_PER_GPU else "") + ", oracle/ngm_oracle.py (a restatement pinned "
                      f"to the reference, NOT the reference itself) on torch CPU fp32, {dt:.1f} s")
    if F == F_PER_GPU:
        one_f = cpu_baseline(budget_s=4.0, F=1)
        res["one_field"] = dict(value=one_f["value"], cores=one_f["cores"], sample=one_f["sample"])
    return res


def time_steps(r, tgt, steps, warmup, spin_up, use_graph=True):
    """K timed steps of the captured iteration (single GPU) + the per-kernel eager pass; -> (seconds, kernels_us, loss, bwd variant)"""
    from neural_graph_mapping_amd import _capi as K
    L = K.lib()
    rep = r.capture_iteration(tgt, seed=7) if use_graph else (lambda: r.optimization_iteration(tgt, seed=7, update=True))
    for _ in range(spin_up + warmup):
        out = rep()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        out = rep()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    L.ngm_profile_reset()
    L.ngm_profile_enable(1)
    for _ in range(steps):
        r.optimization_iteration(tgt, seed=7, update=True)
    torch.cuda.synchronize()
    L.ngm_profile_enable(0)
    kern = {}
    for name, kid in K.KERNEL_IDS.items():
        ms, n = C.c_double(0), C.c_int64(0)
        L.ngm_profile_read(kid, C.byref(ms), C.byref(n))
        if n.value:
            kern[name] = dict(avg_us=1e3 * ms.value / n.value, launches=n.value)
    return dt, kern, float(out["combined"]), L.ngm_debug_last_bwd_variant()


PEAK_L2_GBS = 34500.0                            # MI355X_MICROARCH.md: aggregate L2 bandwidth (8 XCDs x 4 MiB)
GATHER_L2_USEFUL_GBS = 2200.0                    # tools/micro/gather_rate.hip: random 8-byte gathers from an L2-resident 512 KB table,
                                                 # useful bytes chip-wide (every gather pulls a 128-byte line; profiles/r03a_gather_rate_microbench.txt)


def hash_rooflines(kern, n_local, scale_note="", pmc_scale=1.0):
    """roofline objects of the hash variant's two dominant kernels (512 B of gathers / of scatter per sample).
    `traffic` = HBM bytes per launch from the PMC passes of profiles/pmc_hash.json (FETCH_SIZE doubled, MI355X_MICROARCH.md),
    `l2_hit_rate` from TCC_HIT_sum / TCC_MISS_sum of the same profile; both belong to the M1 batch (scaled by `pmc_scale`)."""
    out = {}
    pm = pmc_hash_profile()

    def pmc_of(sub):
        for k, v in pm.items():
            if sub in k:
                return v
        return {}
    hg, ff = kern.get("hash_grad"), kern.get("render_fwd")
    algo = 512 * n_local
    src = "profiles/pmc_hash.json: separate rocprofv3 --pmc passes of the hash variant on the M1 batch, NOT this run " + scale_note
    if hg:
        ach = algo / (hg["avg_us"] * 1e-6) / 1e9
        pv = pmc_of("k_hash_grad")
        out["roofline"] = dict(bound="hbm", kernel="k_hash_grad (simplex search + per-level table in LDS, Q23.40 integer atomics)",
                               achieved=ach, peak=PEAK_HBM_GBS, unit="GB/s", frac=ach / PEAK_HBM_GBS,
                               traffic=(pv["hbm_bytes"] * pmc_scale) if pv.get("hbm_bytes") is not None else None,
                               traffic_source=src if pv else None, l2_hit_rate=pv.get("l2_hit_rate"),
                               avg_launch_us=hg["avg_us"], launches_timed=hg["launches"], algorithmic_bytes_per_launch=algo,
                               timing="HIP events on the launch stream, instrumented pass of the same steps",
                               note="priced against the HBM scatter it replaces (512 B/sample, SURVEY 8d); the kernel itself "
                                    "accumulates in LDS (its real HBM traffic is `traffic`: positions + dL/dE in, partial tables "
                                    "out) and is VALU (simplex search) + LDS-atomic bound, see DESIGN.md §4")
    if ff:
        ach = algo / (ff["avg_us"] * 1e-6) / 1e9
        pv = pmc_of("k_render_fwd")
        out["roofline_fwd"] = dict(bound="hbm", kernel="k_render_fwd<1,1,1,hash> (64 table gathers of 8 B per sample through the "
                                                        "vector L1 / the XCD's L2; fp32 MFMA 32x32x2 for the 32-wide layer)",
                                   achieved=ach, peak=PEAK_HBM_GBS, unit="GB/s", frac=ach / PEAK_HBM_GBS,
                                   traffic=(pv["hbm_bytes"] * pmc_scale) if pv.get("hbm_bytes") is not None else None,
                                   traffic_source=src if pv else None, l2_hit_rate=pv.get("l2_hit_rate"),
                                   l2_gather_microbench=dict(useful_GBs=GATHER_L2_USEFUL_GBS, ratio=ach / GATHER_L2_USEFUL_GBS,
                                                             peak_l2_GBs=PEAK_L2_GBS,
                                                             note="NOT a bound (the kernel exceeds it): random 8-byte gathers from a 512 KB, "
                                                                  "L2-resident table read 2.2 TB/s of useful bytes in tools/micro/gather_rate; "
                                                                  "the kernel's gathers also hit the vector L1 (coarse levels: neighbouring "
                                                                  "samples share vertices).  The honest statement is `frac` of the HBM peak "
                                                                  "at `l2_hit_rate`, with no established ceiling for the gather path"),
                                   avg_launch_us=ff["avg_us"], launches_timed=ff["launches"], algorithmic_bytes_per_launch=algo,
                                   note="algorithmic gather bytes (512 B/sample, SURVEY 8d) against the HBM peak as the survey asks")
    return out


def emit_line(d):
    """the ONE JSON line of the contract, as the LAST thing on stdout: RCCL writes a version banner through C stdio, which
    (stdout being a pipe) would otherwise be flushed at exit, i.e. behind the line"""
    import ctypes
    sys.stdout.flush()
    try:
        ctypes.CDLL(None).fflush(None)
    except Exception:
        pass
    print(json.dumps(d), flush=True)


def shader_clock_under_load(step, n=1500, read=True):
    """sclk as rocm-smi reports it WHILE the GPU works through n more (untimed) iterations; None if it cannot be read.
    EVERY rank runs the n iterations (with a process group each of them holds the loss exchange: a rank that skipped them
    would leave the others' collectives unmatched -- rounds 4-5 ran them on rank 0 alone, which no 1-GPU run could notice);
    only the rank with read=True asks rocm-smi."""
    import subprocess
    try:
        for i in range(n):
            step(i)
        if not read:
            torch.cuda.synchronize()
            return None
        r = subprocess.run(["rocm-smi", "--showclocks", "--json"], capture_output=True, text=True, timeout=20)
        torch.cuda.synchronize()
        js = json.loads(r.stdout[r.stdout.index("{"):])
        card = js[sorted(js)[torch.cuda.current_device() if len(js) > torch.cuda.current_device() else 0]]
        for k, v in card.items():
            if "sclk" in k.lower():
                import re
                m = re.search(r"(\d+)\s*Mhz", str(v), re.I)
                if m:
                    return int(m.group(1))
    except Exception:
        torch.cuda.synchronize()
    return None


def pmc_hash_profile():
    """profiles/pmc_hash.json: HBM bytes (FETCH_SIZE x 2 + WRITE_SIZE, separate rocprofv3 --pmc passes) and L2 hit rate
    (TCC_HIT_sum / (TCC_HIT_sum + TCC_MISS_sum)) per launch of the hash variant's kernels on the M1 batch -- NOT this run"""
    pth = os.path.join(ROOT, "profiles", "pmc_hash.json")
    if not os.path.exists(pth):
        return {}
    try:
        return json.load(open(pth)).get("kernels", {})
    except Exception:
        return {}


def aux_default_line(dev, args, use_graph):
    """The iteration the reference actually runs by default (config/neural_graph_map.yaml:6-20, 60-63): 32 active fields x
    512 rays x (8 coarse + 16 depth-guided) samples, permutohedral hash 16 levels x 2 features + ONE hidden layer of 32."""
    Fd, Sc, Sg = 32, 8, 16
    r = build_renderer(dev, Fd, "hash", s_c=Sc, s_g=Sg, matmul=args.matmul, hash_atomics=args.hash_atomics)
    pos, quat, t = synth_target(Fd, R, seed=4242)
    r.set_field_poses(pos.to(dev), quat.to(dev))
    tgt = type(t)(*[v.to(dev) if isinstance(v, torch.Tensor) else v for v in t])
    tgt = tgt._replace(field_ids=torch.arange(Fd, device=dev))
    dth, kh, lh, bvh = time_steps(r, tgt, args.steps, args.warmup, 50, use_graph)
    n = Fd * R * (Sc + Sg)
    roof = hash_rooflines(kh, n, scale_note="(traffic / L2 hit rate: PMC of the M1 batch scaled by samples)", pmc_scale=n / (8 * 512 * 128))
    return dict(workload="32 fields x 512 rays x (8 coarse + 16 depth-guided) samples, permutohedral hash (16 levels x 2 features, "
                         "2^12 entries) + 1x32 MLP: the reference's default iteration (config/neural_graph_map.yaml:6-20, 60-63; "
                         "parity of the hash encoding unpinned: third-party CUDA package absent)",
                value=n * args.steps / dth, unit="ray-samples/s", ms_per_step=1e3 * dth / args.steps, final_loss=lh, bwd_variant=bvh,
                hash_grad_atomics=args.hash_atomics,
                launch="hipGraph replay" if use_graph else "eager", kernels_us={k: round(v["avg_us"], 2) for k, v in kh.items()}, **roof)


def aux_render_image_line(dev):
    """The evaluation path: `render_image` 640 x 480 pixels x 640 eval samples, kNN-blended (K = 2) over a
    map of 126 Fourier(64) + 2x64 fields on a 9 x 7 x 2 grid in front of the camera -- ONE C call (ngm_render_eval_knn).
    Median of five renders after one warm-up; the same workload as tools/eval_bench.py."""
    import time
    from neural_graph_mapping_amd import models as M
    from neural_graph_mapping_amd import renderer as Rr
    torch.manual_seed(0)
    g = torch.arange(-2.0, 2.01, 0.5)
    pos = torch.stack(torch.meshgrid(g, g[:7], torch.tensor([-3.0, -2.5]), indexing="ij"), -1).reshape(-1, 3)
    NF, S = pos.shape[0], 640
    quat = torch.zeros(NF, 4)
    quat[:, 0] = 1
    model = M.NeuralFieldSet(dim_points=3, field_type="neural_graph_mapping.models.NeuralField", field_kwargs=dict(
        encoding_type="neural_graph_mapping.positional_encodings.PositionalEncodingFourier",
        encoding_kwargs=dict(dim_in=3, dim_out=D_ENC, mu=0.0, sigma=4.0, raw_coords=True), num_layers=N_LAYERS, dim_out=4),
        num_knn=2, distance_factor=10.0, outside_value=1.0, field_radius=0.5, scale_mode="unit_cube").to(dev)
    cfg = Rr.shipped_config(field_radius=0.5, eval_near_distance=0.0, eval_far_distance=8.0, eval_num_samples=S)
    cam = Rr.Camera(640, 480, 554.2562584220408, 554.2562584220408, 319.5, 239.5, pixel_center=0.0)
    r = Rr.NeuralGraphRenderer(model, cam, cfg, device=dev)
    r.add_fields(NF)
    r.set_field_poses(pos.to(dev), quat.to(dev))
    r.eval()                                # rm.py:1978: the evaluation's sampling parameters
    c2w = torch.eye(4, device=dev)
    r.render_image(c2w)
    times = []
    for _ in range(5):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        r.render_image(c2w)
        torch.cuda.synchronize()
        times.append(time.perf_counter() - t0)
    dt = sorted(times)[2]
    return dict(workload=f"render_image 640 x 480 x {S} eval samples, kNN blend (K = 2) over {NF} Fourier({D_ENC}) + {N_LAYERS}x64 fields, "
                         "in-kernel Philox jitter, one ngm_render_eval_knn call (internal blocks of 32768 rays)",
                ms_per_image=1e3 * dt, value=640 * 480 * S / dt, unit="ray-samples/s (render only)", knn_matmul=r.last_matmul("knn"))


def aux_m2_line(dev, variant, matmul="auto", launches=200):
    """M2 of SURVEY 8d: RENDER ONLY, no gradient, no stash -- F x R = 8 x 512 = 4096 rays, S = 128 samples uniform in
    [near, far], no depth guidance (eval-style).  Timed AT THE C ABI: `launches` back-to-back ngm_render_fwd calls (inference
    workspace = NULL, loss sums = NULL) on torch's current stream -- HIP events around every launch (the library's profile
    hooks) give the kernel time, a synchronize-bracketed wall clock around the whole loop gives the end-to-end rate incl.
    launch overhead (no Python layer between the launches beyond the ctypes call).  Fourier: MFMA roofline on the 16 896
    algorithmic flop / sample; hash: HBM roofline on the 512 B of table gathers / sample (as SURVEY 8d prices them)."""
    from neural_graph_mapping_amd import _capi as K
    from neural_graph_mapping_amd import ops
    L = K.lib()
    S = S_C + S_G
    r = build_renderer(dev, F_PER_GPU, variant, s_c=S, s_g=0, matmul=matmul)
    pos, quat, t = synth_target(F_PER_GPU, R, seed=1000)
    fc, rc = r._fc, r._rc_for(None, False)
    keep = []
    rays = ops.make_rays(rc, t.ijs.to(dev), t.c2ws.to(dev), t.near_distances.to(dev), t.far_distances.to(dev), None, pos.to(dev),
                         quat.to(dev), seed=11, keep=keep)
    ps = ops.params_struct(fc, r._model.kernel_params())
    rgbds, cvars = torch.empty(F_PER_GPU, R, 4, device=dev), torch.empty(F_PER_GPU, R, 3, device=dev)
    dvars, term = torch.empty(F_PER_GPU, R, device=dev), torch.empty(F_PER_GPU, R, device=dev)
    pred = K.Prediction(rgbds.data_ptr(), cvars.data_ptr(), dvars.data_ptr(), term.data_ptr())
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)

    def launch():
        K.check(L.ngm_render_fwd(C.byref(fc), C.byref(rc), C.byref(ps), C.byref(rays), None, C.byref(pred), None, None, 0, st),
                "ngm_render_fwd (M2)")
    for _ in range(50):
        launch()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(launches):
        launch()
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / launches
    L.ngm_profile_reset()
    L.ngm_profile_enable(1)
    for _ in range(launches):
        launch()
    torch.cuda.synchronize()
    L.ngm_profile_enable(0)
    ms, n = C.c_double(0), C.c_int64(0)
    L.ngm_profile_read(K.KERNEL_IDS["render_fwd"], C.byref(ms), C.byref(n))
    us = 1e3 * ms.value / max(n.value, 1)
    n_samp = F_PER_GPU * R * S
    if variant == "hash":
        ach = 512 * n_samp / (us * 1e-6) / 1e9
        roof = dict(bound="hbm", kernel="k_render_fwd<1,1,1,hash> (inference instance: no stash)", achieved=ach, peak=PEAK_HBM_GBS,
                    unit="GB/s", frac=ach / PEAK_HBM_GBS, traffic=None, algorithmic_bytes_per_launch=512 * n_samp,
                    note="512 B of table gathers per sample (SURVEY 8d) against the HBM peak; the tables are L2-resident "
                         "(aux_hash.roofline_fwd.l2_hit_rate): no bound for the L2 -> L1 gather path is established")
    else:
        ach = FLOP_FWD * n_samp / (us * 1e-6) / 1e12
        roof = dict(bound="mfma", kernel="k_render_fwd<2,2,2> (inference instance: no stash)", achieved=ach, peak=PEAK_F32_MFMA_TF,
                    unit="TFLOP/s", frac=ach / PEAK_F32_MFMA_TF, traffic=None, algorithmic_flop_per_launch=FLOP_FWD * n_samp)
    roof.update(avg_launch_us=us, launches_timed=int(n.value), timing="HIP events on the launch stream around every launch")
    return dict(workload=f"M2: render only (no gradient, no stash), {F_PER_GPU} fields x {R} rays x {S} samples uniform in [near, far], "
                         + ("Fourier(64,raw)+2x64 MLP" if variant != "hash" else "permutohedral hash 16x2 + 1x32 MLP (parity unpinned)")
                         + ", in-kernel Philox jitter, ngm_render_fwd called back to back at the C ABI",
                value=n_samp / wall, unit="ray-samples/s (render only)", us_per_call_wall=1e6 * wall, kernel_us=us,
                matmul=r.last_matmul("forward") or r.mlp_matmul, checksum=float(rgbds.double().sum()), roofline=roof)


def launch_ranks(n):
    """Re-exec this script under torch.distributed.run with n ranks on this node; returns the exit code."""
    import socket
    import subprocess
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def scene_sim(args, rank, world, dev):
    """Strong-scaling picture of a REAL mapping iteration (not the weak-scaling headline): 200 pose-graph fields in the
    map, sharded owner = id % world; every iteration trains 32 of them (the count of config/neural_graph_map.yaml:60,
    drawn like rm.py:1280-1319 draws its random half: uniformly without replacement, same seed on every rank) with 512
    rays each.  A rank renders only the active fields it owns; the one collective is the 64-byte loss all-reduce.
    Reports ms per iteration (max over ranks, barrier-bracketed) for the default sample count 8+16 and for the metric's
    64+64, the spread of active fields per rank, and on one GPU the step time against the number of active fields."""
    from neural_graph_mapping_amd import distributed as D
    NF, FA, Rn, NB = 200, 32, 512, 8
    out = dict(mode="scene-sim", n_gpus=world, fields_total=NF, active_per_iteration=FA, rays_per_field=Rn,
               launch="eager (the active set and with it the per-rank batch shape change every iteration)", configs={})
    cur = torch.arange(NF - 50, NF)                 # the observed fields: the 50 most recent ones
    # spread of the active set over 8 owner ranks under both sampler policies (host arithmetic, 1000 draws each)
    spread = {}
    for pol in ("reference", "balanced_by_owner"):
        g8 = torch.Generator().manual_seed(7)
        mx = []
        for _ in range(1000):
            ids = (D.draw_fields_reference(cur, NF, FA, generator=g8)[0] if pol == "reference"
                   else D.draw_fields_balanced(cur, NF, FA, 8, generator=g8))
            mx.append(int(torch.bincount(ids % 8, minlength=8).max()))
        spread[pol] = dict(mean_per_rank=FA / 8, max_over_ranks_mean=sum(mx) / len(mx), max_over_ranks_worst=max(mx))
    out["active_fields_per_rank_at_world8"] = spread
    policies = ("reference", "balanced_by_owner") if world > 1 else ("reference",)
    peer = None
    out["exchange"] = args.exchange
    for policy in policies:
        gen = torch.Generator().manual_seed(2024)
        active_sets = [(D.draw_fields_reference(cur, NF, FA, generator=gen)[0] if policy == "reference"
                        else D.draw_fields_balanced(cur, NF, FA, world, generator=gen)) for _ in range(NB)]
        for label0, (s_c, s_g) in dict(default_8p16=(8, 16), metric_64p64=(64, 64)).items():
            label = label0 if policy == "reference" else label0 + "_balanced_by_owner"
            owned = D.local_field_slots(NF, rank, world)
            r = build_renderer(dev, len(owned), args.variant, s_c, s_g)
            pos_all, quat_all, _ = synth_target(NF, 1, seed=77)
            r.set_field_poses(pos_all[owned].to(dev), quat_all[owned].to(dev))
            if world > 1:
                r.process_group = torch.distributed.group.WORLD
                if args.exchange == "peer":
                    if peer is None:
                        peer = D.PeerExchange.try_create(torch.distributed.group.WORLD, dev)
                    r.peer_exchange = peer
            batches, counts = [], []
            for b, ids in enumerate(active_sets):
                _, _, t = synth_target(FA, Rn, seed=500 + b)
                shift = (pos_all[ids] - synth_target(FA, 1, seed=500 + b)[0])[:, None]     # rays around the map's field centres
                c2w = t.c2ws.clone()
                c2w[..., :3, 3] += shift
                t = t._replace(c2ws=c2w, field_ids=ids)
                tl = D.shard_target(t, rank, world)
                tl = tl._replace(field_ids=D.global_to_local(tl.field_ids, world))
                batches.append(type(tl)(*[v.to(dev) if isinstance(v, torch.Tensor) else v for v in tl]))
                counts.append(int(tl.ijs.shape[0]))

            def run(n):
                for i in range(n):
                    r.optimization_iteration(batches[i % NB], seed=3, update=True)
            # the clocks of an idle GPU take ~0.1 s to ramp up: a generous, FIXED number of untimed iterations (the same on
            # every rank: each iteration contains a collective)
            run(max(args.warmup, 256))
            torch.cuda.synchronize()
            if world > 1:
                torch.distributed.barrier()
            t0 = time.perf_counter()
            run(args.steps)
            torch.cuda.synchronize()
            if world > 1:
                torch.distributed.barrier()
            dt = time.perf_counter() - t0
            cnt = torch.tensor(counts, device=dev, dtype=torch.float32)
            if world > 1:
                tt = torch.tensor([dt], device=dev, dtype=torch.float64)
                torch.distributed.all_reduce(tt, op=torch.distributed.ReduceOp.MAX)
                dt = float(tt.item())
                allc = [torch.empty_like(cnt) for _ in range(world)]
                torch.distributed.all_gather(allc, cnt)
                cnt = torch.stack(allc)                                           # (world, NB)
            else:
                cnt = cnt[None]
            S = s_c + s_g
            res = dict(ms_per_iteration=1e3 * dt / args.steps, ray_samples_per_s=FA * Rn * S * args.steps / dt,
                       active_fields_per_rank=dict(mean=float(cnt.mean()), min=float(cnt.min()), max=float(cnt.max()),
                                                   max_over_ranks_mean=float(cnt.max(0).values.mean())))
            if world == 1:                         # fixed cost vs batch size: what a rank with few active fields pays
                sweep = {}
                for Fa in (1, 2, 4, 8, 32):
                    _, _, t = synth_target(Fa, Rn, seed=900 + Fa)
                    ids = torch.arange(Fa)
                    t = t._replace(c2ws=t.c2ws + 0, field_ids=ids)
                    c2w = t.c2ws.clone()
                    c2w[..., :3, 3] += (pos_all[ids] - synth_target(Fa, 1, seed=900 + Fa)[0])[:, None]
                    tb = type(t)(*[v.to(dev) if isinstance(v, torch.Tensor) else v for v in t._replace(c2ws=c2w)])
                    for graph in (False, True):
                        step = r.capture_iteration(tb, seed=3) if graph else (lambda: r.optimization_iteration(tb, seed=3, update=True))
                        for _ in range(10):
                            step()
                        torch.cuda.synchronize()
                        t0 = time.perf_counter()
                        for _ in range(50):
                            step()
                        torch.cuda.synchronize()
                        sweep[f"F={Fa},{'graph' if graph else 'eager'}"] = round(1e3 * (time.perf_counter() - t0) / 50, 4)
                    # where the fixed cost sits: HIP-event time of every kernel of the step at this batch size
                    from neural_graph_mapping_amd import _capi as K
                    L = K.lib()
                    L.ngm_profile_reset()
                    L.ngm_profile_enable(1)
                    for _ in range(50):
                        r.optimization_iteration(tb, seed=3, update=True)
                    torch.cuda.synchronize()
                    L.ngm_profile_enable(0)
                    kk = {}
                    for name, kid in K.KERNEL_IDS.items():
                        ms, n = C.c_double(0), C.c_int64(0)
                        L.ngm_profile_read(kid, C.byref(ms), C.byref(n))
                        if n.value:
                            kk[name] = round(1e3 * ms.value / n.value, 2)
                    sweep[f"F={Fa},kernels_us"] = kk
                res["ms_per_step_vs_active_fields"] = sweep
            out["configs"][label] = res
    if rank == 0:
        emit_line(out)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--windows", type=int, default=11,
                    help="number of consecutive timed windows of --steps steps each (after ONE --warmup phase); the line's "
                         "ms_per_step / value are the median window, all windows are listed in ms_per_step_windows")
    ap.add_argument("--min-seconds", type=float, default=6.0,
                    help="keep adding timed windows of --steps steps (same protocol, median headline) until this much timed GPU "
                         "work has run, so that an external sampler (rocm-smi, the driver's gpu_busy) sees the load: 11 windows of "
                         "20 steps are 55 ms.  0: exactly --windows windows")
    ap.add_argument("--no-aux-default", action="store_true",
                    help="skip `aux_default` and `aux_render_image`: the reference's default iteration (32 fields x 512 rays x (8 + 16) samples, "
                         "hash 16 x 2 + 1 x 32 network: config/neural_graph_map.yaml)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--eager", action="store_true", help="launch every kernel from Python instead of replaying a hipGraph")
    ap.add_argument("--fields-total", type=int, default=0,
                    help="strong scaling (SURVEY 8d): this many fields in total, split field-per-GPU over the ranks "
                         "(default 0 = weak scaling, 8 fields per GPU)")
    ap.add_argument("--variant", choices=["fourier", "hash"], default="fourier",
                    help="field network: fourier = the headline M1 workload (default); hash = the reference's default "
                         "permutohedral-hash network on the same batch (auxiliary measurement)")
    ap.add_argument("--matmul", choices=["auto", "f32", "bf16x3"], default="auto",
                    help="hidden layers of the forward kernels: auto (library default) = the exact three-way bf16 split with fp32 "
                         "accumulation where it is compiled (this workload), f32 = exact-fp32 MFMA everywhere; the line's `dtype` "
                         "says which ran, and the other one is measured next to it (`matmul_alternative`)")
    ap.add_argument("--hash-atomics", choices=["exact", "float"], default="exact",
                    help="accumulation of the hash-table gradient in the hash lines (ngm_hash_grad_atomics): exact = Q23.40 integer LDS "
                         "atomics, bitwise reproducible (default, and what the lines report unless named); float = fp32 LDS atomics")
    ap.add_argument("--no-aux-hash", action="store_true",
                    help="skip the auxiliary measurement of the reference's default network (hash encoding + 1x32 MLP) that the "
                         "default line carries as `aux_hash`")
    ap.add_argument("--exchange", choices=["rccl", "peer"], default="rccl",
                    help="multi-GPU: how the 64-byte loss sums are exchanged.  rccl (default): torch.distributed.all_reduce "
                         "between two hipGraphs; peer: ngm_loss_exchange, one kernel of xGMI peer writes inside ONE graph "
                         "(distributed.PeerExchange; falls back to rccl when its set-up self-test fails)")
    ap.add_argument("--scene-sim", action="store_true",
                    help="auxiliary strong-scaling measurement of a realistic mapping iteration (200 fields, 32 active per "
                         "iteration, sharded id %% world) instead of the headline line; see scene_sim()")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # plain `python bench.py --gpus N`: start the N ranks ourselves (one process per GPU, torchrun rendezvous on
        # 127.0.0.1) and pass their output through; the rank-0 child prints the JSON line
        raise SystemExit(launch_ranks(args.gpus))

    from neural_graph_mapping_amd import _capi as K
    from neural_graph_mapping_amd import distributed as D
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a ROCm GPU (no CPU fallback for the product path)")
    share = os.environ.get("NGM_BENCH_SHARE_GPU", "0") == "1"       # test rigs with fewer GPUs than ranks (gloo only)
    backend = os.environ.get("NGM_DIST_BACKEND") or None
    world_env, local_env = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("LOCAL_RANK", "0"))
    if world_env != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world_env}: refusing to report a line for the wrong job size")
    ndev = torch.cuda.device_count()
    if local_env >= ndev and not share:
        raise SystemExit(f"rank with LOCAL_RANK={local_env} has no GPU of its own ({ndev} visible); one process per GPU")
    dev_index = local_env % ndev
    torch.cuda.set_device(dev_index)                                 # rank -> GPU binding before any collective
    os.environ["LOCAL_RANK"] = str(dev_index)
    rank, local, world = D.init_from_env(backend)
    dev = torch.device("cuda", dev_index)
    ranks_seen = 1
    if world > 1:
        ones = torch.ones(1, device=dev)
        torch.distributed.all_reduce(ones)
        ranks_seen = int(ones.item())
        if ranks_seen != args.gpus:
            raise SystemExit(f"--gpus {args.gpus} but the all-reduce saw {ranks_seen} ranks")
    L = K.lib()
    if args.scene_sim:
        scene_sim(args, rank, world, dev)
        if torch.distributed.is_initialized():
            torch.distributed.destroy_process_group()
        return

    # field-per-GPU sharding: rank r owns global fields r, r+world, ... ; local slot = id // world
    strong = args.fields_total > 0
    if strong and args.fields_total % world:
        raise SystemExit("--fields-total must be a multiple of the number of GPUs")
    F_PER_GPU = args.fields_total // world if strong else globals()["F_PER_GPU"]
    r = build_renderer(dev, F_PER_GPU, args.variant, matmul=args.matmul, hash_atomics=args.hash_atomics)
    pos, quat, tgt_cpu = synth_target(F_PER_GPU, R, seed=1000 + rank)
    r.set_field_poses(pos.to(dev), quat.to(dev))
    tgt = type(tgt_cpu)(*[v.to(dev) if isinstance(v, torch.Tensor) else v for v in tgt_cpu])
    tgt = tgt._replace(field_ids=torch.arange(F_PER_GPU, device=dev))      # local slots of this rank's fields
    if world > 1 or torch.distributed.is_initialized():
        r.process_group = torch.distributed.group.WORLD
        if args.exchange == "peer":
            r.peer_exchange = D.PeerExchange.try_create(torch.distributed.group.WORLD, dev)

    # the whole iteration (3 kernels on the split path: fused forward, MLP backward incl. the compositing backward,
    # gradient reduction + Adam) is captured once into a hipGraph and replayed;
    # the Adam step counter and the Philox jitter offset advance on the device inside the graph.
    # multi-GPU: two graphs around the loss all-reduce (renderer.capture_iteration); --eager launches every kernel
    use_graph = not args.eager
    replay = r.capture_iteration(tgt, seed=7) if use_graph else None

    def step(i):
        return replay() if use_graph else r.optimization_iteration(tgt, seed=7, update=True)

    # an idle GPU needs ~0.1 s under load before its clocks are up: SPIN_UP untimed iterations first (declared in the line),
    # then the W warm-up steps and the K timed steps of the contract
    for i in range(SPIN_UP):
        out = step(i)
    for i in range(args.warmup):
        out = step(i)

    def window():
        """EXACTLY K steps between barrier + synchronize on both sides; max over ranks"""
        torch.cuda.synchronize()
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(args.steps):
            step(i)
        torch.cuda.synchronize()
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()
        d = time.perf_counter() - t0
        if world > 1:
            tt = torch.tensor([d], device=dev, dtype=torch.float64)
            torch.distributed.all_reduce(tt, op=torch.distributed.ReduceOp.MAX)
            d = float(tt.item())
        return d

    # measurement protocol: the contract's timed region (W warm-up steps, then exactly K steps) repeated as `--windows`
    # consecutive windows; the headline is the MEDIAN window (one 20-step window is 5 ms: +-8 % between boxes and runs).
    # --min-seconds keeps adding windows until that much timed work has run (lets an external sampler see the load).
    wins = [window() for _ in range(max(1, args.windows))]
    while args.min_seconds > 0 and sum(wins) < args.min_seconds:
        wins.append(window())
    if world > 1:        # the same count on every rank (each window holds collectives): agree on rank 0's
        nw = torch.tensor([len(wins)], device=dev)
        torch.distributed.broadcast(nw, 0)
        while len(wins) < int(nw.item()):
            wins.append(window())
        wins = wins[:int(nw.item())]
    dt = sorted(wins)[len(wins) // 2] if len(wins) % 2 else 0.5 * (sorted(wins)[len(wins) // 2 - 1] + sorted(wins)[len(wins) // 2])
    sclk = shader_clock_under_load(step, read=(rank == 0))
    loss = float(out["combined"])

    # per-kernel device time: HIP events recorded on the launch stream around every kernel launch
    # (C-ABI hooks).  Under hipGraph replay host-side event records are not part of the graph, so the
    # same K steps are re-run eagerly right after the timed region for this breakdown.
    L.ngm_profile_reset()
    L.ngm_profile_enable(1)
    for i in range(args.steps):
        r.optimization_iteration(tgt, seed=7, update=True)
    torch.cuda.synchronize()
    kern = {}
    L.ngm_profile_enable(0)
    bwd_variant = L.ngm_debug_last_bwd_variant()     # of the measured path (read before the side measurement below runs the other one)
    # the arithmetic the library resolved `auto` to for THIS batch shape (not the host-side expectation)
    resolved = r.last_matmul("forward") or r.mlp_matmul
    for name, kid in K.KERNEL_IDS.items():
        ms, n = C.c_double(0), C.c_int64(0)
        L.ngm_profile_read(kid, C.byref(ms), C.byref(n))
        if n.value:
            kern[name] = dict(avg_us=1e3 * ms.value / n.value, launches=n.value)

    # side measurement, after and outside the timed region above: the same K steps with the OTHER arithmetic of the hidden
    # layers' matrix products (exact-fp32 MFMA <-> exact three-way bf16 split, forward and backward), so that one line carries both
    side = None
    if world == 1 and args.variant == "fourier" and args.matmul == "auto" and not strong:
        other = "f32" if resolved == "bf16x3" else "bf16x3"
        r3 = build_renderer(dev, F_PER_GPU, args.variant, matmul=other)
        r3.set_field_poses(pos.to(dev), quat.to(dev))
        rep3 = r3.capture_iteration(tgt, seed=7) if use_graph else (lambda: r3.optimization_iteration(tgt, seed=7, update=True))
        for _ in range(args.warmup):
            rep3()
        torch.cuda.synchronize()
        t3 = time.perf_counter()
        for _ in range(args.steps):
            o3 = rep3()
        torch.cuda.synchronize()
        dt3 = time.perf_counter() - t3
        side = dict(matmul=other, ms_per_step=1e3 * dt3 / args.steps, value=F_PER_GPU * R * (S_C + S_G) * args.steps / dt3,
                    dtype=DTYPE_LABEL[other], final_loss=float(o3["combined"]),
                    note=f"python bench.py --matmul {other} makes this the measured path")
        del r3, rep3

    devs = [dev_index]
    if world > 1:                                    # which GPU every rank really ran on (rank -> GPU binding evidence)
        dv = torch.zeros(world, device=dev, dtype=torch.int64)
        dv[rank] = dev_index
        torch.distributed.all_reduce(dv)
        devs = dv.tolist()
    if args.variant == "hash" and bwd_variant == 5:
        resolved_note = "hidden layer: fp32 MFMA in the fused forward, three-way bf16 split in the backward (k_hash_mlp_bwd)"
    else:
        resolved_note = None
    # auxiliary: the reference's DEFAULT network (hash 16x2 + 1x32 MLP) on the same batch, in the same run
    aux_hash = None
    if world == 1 and args.variant == "fourier" and not strong and not args.no_aux_hash:
        rh = build_renderer(dev, F_PER_GPU, "hash", matmul=args.matmul, hash_atomics=args.hash_atomics)
        rh.set_field_poses(pos.to(dev), quat.to(dev))
        dth, kh, lh, bvh = time_steps(rh, tgt, args.steps, args.warmup, 20, use_graph)
        n_h = F_PER_GPU * R * (S_C + S_G)
        aux_hash = dict(workload="same M1 batch, permutohedral hash (16 levels x 2 features, 2^12 entries) + 1x32 MLP = the network of "
                                 "config/neural_graph_map.yaml (parity unpinned: third-party CUDA package absent)",
                        value=n_h * args.steps / dth, unit="ray-samples/s", ms_per_step=1e3 * dth / args.steps, final_loss=lh,
                        bwd_variant=bvh, kernels_us={k: round(v["avg_us"], 2) for k, v in kh.items()}, **hash_rooflines(kh, n_h))
        del rh
    aux_default = None
    if world == 1 and args.variant == "fourier" and not strong and not args.no_aux_default:
        aux_default = aux_default_line(dev, args, use_graph)
    aux_image = aux_m2 = None
    if world == 1 and args.variant == "fourier" and not strong and not args.no_aux_default:
        aux_image = aux_render_image_line(dev)
        aux_m2 = dict(fourier=aux_m2_line(dev, "fourier", args.matmul), hash=aux_m2_line(dev, "hash", args.matmul))
    if rank == 0:
        n_local = F_PER_GPU * R * (S_C + S_G)
        value = world * n_local * args.steps / dt
        res = dict(metric="ray-samples/sec (train step: fwd+loss+bwd+Adam, 4096 rays x 128 samples per GPU)",
                   value=value, unit="ray-samples/s", n_gpus=ranks_seen, steps=args.steps, warmup=args.warmup,
                   ms_per_step=1e3 * dt / args.steps,
                   ms_per_step_windows=dict(n=len(wins), steps_each=args.steps, median=1e3 * dt / args.steps,
                                            min=1e3 * min(wins) / args.steps, max=1e3 * max(wins) / args.steps,
                                            first=1e3 * wins[0] / args.steps, timed_seconds=round(sum(wins), 4),
                                            all=[round(1e3 * w / args.steps, 5) for w in wins[:64]],
                                            all_note="the first 64 windows" if len(wins) > 64 else "every window",
                                            headline="median window"),
                   sclk_mhz=sclk, higher_is_better=True, scaling="strong" if strong else "weak", vs_baseline=None,
                   dtype=DTYPE_LABEL[resolved], data="synthetic",
                   config=dict(workload=f"M1: {F_PER_GPU} fields x 512 rays x (64 coarse + 64 depth-guided) samples per GPU, "
                                        + ("Fourier(64,raw)+2x64 MLP" if args.variant == "fourier" else
                                           "permutohedral hash (16 levels x 2, 2^12 entries)+1x32 MLP [auxiliary variant]")
                                        + ", nrgbd compositing, NRGBD intrinsics",
                               fields_per_gpu=F_PER_GPU, rays_per_field=R, samples_per_ray=S_C + S_G,
                               sharding=f"field-per-GPU x{world}", ranks_seen=ranks_seen, spin_up_steps=SPIN_UP,
                               devices=sorted(set(devs)), jitter="in-kernel Philox",
                               launch=("eager" if (not use_graph or getattr(replay, "graph", None) is None) else
                                       "hipGraph replay" if r.process_group is None else
                                       "1 hipGraph incl. the peer loss exchange" if r.peer_exchange is not None else
                                       "2 hipGraphs + all-reduce"), final_loss=loss))
        fb, ff = kern.get("field_bwd"), kern.get("render_fwd")
        if fb and args.variant == "fourier":
            achieved = FLOP_BWD * n_local / (fb["avg_us"] * 1e-6) / 1e12
            traffic, traffic_src = None, None
            pmc = os.path.join(ROOT, "profiles", "pmc_field_bwd.json")
            if os.path.exists(pmc):
                traffic = json.load(open(pmc)).get("hbm_bytes_per_launch")
                traffic_src = "profiles/pmc_field_bwd.json: separate rocprofv3 --pmc pass of this workload, NOT this run"
            variant = bwd_variant
            kname = {0: "k_field_bwd<2,2,2> (v_mfma_f32_32x32x2_f32, forward recompute)",
                     1: "k_field_bwd16<4,4,2> (v_mfma_f32_16x16x4_f32, forward recompute)",
                     2: "k_field_bwd16s<2> (v_mfma_f32_16x16x4_f32, activations from the forward's stash)",
                     3: "k_field_bwd_b3<2> (v_mfma_f32_32x32x16_bf16, 6 products, activations from the forward's stash)"}.get(variant, "?")
            fused_comp = bool(L.ngm_debug_last_comp_fused())
            if fused_comp:
                kname += " + the compositing backward of each tile (what k_stash_bwd did in a launch of its own: +4 % kernel time " \
                         "against the plain kernel, -17 us of launch per step; the algorithmic flops counted here are the MLP's only)"
            res["roofline"] = dict(bound="mfma", kernel=kname, fused_compositing_backward=fused_comp, achieved=achieved,
                                   peak=PEAK_F32_MFMA_TF, unit="TFLOP/s", frac=achieved / PEAK_F32_MFMA_TF,
                                   traffic=traffic, traffic_source=traffic_src, avg_launch_us=fb["avg_us"],
                                   launches_timed=fb["launches"],
                                   timing="HIP events on the launch stream, instrumented pass of the same steps",
                                   algorithmic_flop_per_launch=FLOP_BWD * n_local)
            if variant == 3:    # both fractions, as for the forward: algorithmic fp32 flops / fp32 peak, issued bf16 flops / bf16 peak
                issued_b = 6 * 2 * 4 * (64 * 64) * n_local         # weight + data gradient of both 64x64 layers, six bf16 products each
                res["roofline"].update(issued_bf16_tflops=issued_b / (fb["avg_us"] * 1e-6) / 1e12, peak_bf16=2500.0,
                                       frac_bf16=issued_b / (fb["avg_us"] * 1e-6) / 1e12 / 2500.0,
                                       note="frac = algorithmic fp32 flops against the fp32 MFMA peak (the arithmetic is fp32-exact "
                                            "products on the bf16 pipe, so it may exceed what fp32 MFMA could reach); frac_bf16 = "
                                            "issued bf16 flops against the dense bf16 peak")
            if ff:      # the second MFMA kernel and the whole step against the same peak (algorithmic MLP flops only)
                a_f = FLOP_FWD * n_local / (ff["avg_us"] * 1e-6) / 1e12
                res["roofline_fwd"] = dict(bound="mfma", kernel="k_render_fwd<2,2,2> (v_mfma_f32_32x32x2_f32)", achieved=a_f,
                                           peak=PEAK_F32_MFMA_TF, unit="TFLOP/s", frac=a_f / PEAK_F32_MFMA_TF,
                                           avg_launch_us=ff["avg_us"], algorithmic_flop_per_launch=FLOP_FWD * n_local)
                if resolved == "bf16x3":     # both fractions: algorithmic fp32 flops / fp32 peak, issued bf16 flops / bf16 peak
                    issued = 6 * 2 * (64 * 64 + 64 * 64) * n_local         # six bf16 products per fp32 product, hidden layers
                    res["roofline_fwd"].update(kernel="k_render_fwd<2,2,2,bf16x3> (v_mfma_f32_32x32x16_bf16, 6 products)",
                                               issued_bf16_tflops=issued / (ff["avg_us"] * 1e-6) / 1e12, peak_bf16=2500.0,
                                               frac_bf16=issued / (ff["avg_us"] * 1e-6) / 1e12 / 2500.0)
            a_s = (FLOP_FWD + FLOP_BWD) * n_local / (dt / args.steps) / 1e12
            res["roofline_step"] = dict(bound="mfma", achieved=a_s, peak=PEAK_F32_MFMA_TF, unit="TFLOP/s",
                                        frac=a_s / PEAK_F32_MFMA_TF, note="whole timed step (all kernels + launch gaps), "
                                        "algorithmic MLP flops fwd 16 896 + bwd 33 792 per sample")
        if args.variant == "hash":
            res.update(hash_rooflines(kern, n_local))
            if resolved_note:
                res["dtype"] = DTYPE_LABEL["f32"] + "; " + resolved_note
        res["kernels_us"] = {k: round(v["avg_us"], 2) for k, v in kern.items()}
        if side:
            res["matmul_alternative"] = side
        if aux_hash:
            res["aux_hash"] = aux_hash
        if aux_default:
            res["aux_default"] = aux_default
        if aux_image:
            res["aux_render_image"] = aux_image
        if aux_m2:
            res["aux_m2"] = aux_m2
        if world == 1 and not args.no_cpu_baseline:
            res["cpu_baseline"] = cpu_baseline()
        emit_line(res)
    if torch.distributed.is_initialized():
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
