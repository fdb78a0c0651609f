"""Standalone stage throughput on the GPU (which roofline each stage sits on).

    python tools/stage_bench.py [--rays 262144] [--samples 128] [--out gpurun_out/stages.json]

Times the standalone C-ABI entry points with HIP events (torch.cuda.Event on the launch stream = torch's
current stream, which is the stream the ops launch on) on a batch large enough to leave the latency regime,
and reports achieved GB/s against the algorithmic bytes of SURVEY 8(d) for the HBM-bound stages and TFLOP/s
for the MLP stage.  Prints one JSON object.
"""
import argparse
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from neural_graph_mapping_amd import _capi as K  # noqa: E402
from neural_graph_mapping_amd import ops  # noqa: E402

HBM_PEAK_GBS = 8000.0
MFMA_F32_PEAK_TF = 157.3


def timeit(fn, iters=20, warmup=3):
    for _ in range(warmup):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3      # seconds


def kernel_us(fn, kid, iters=10):
    """average duration of the library's launches of kernel class `kid` inside fn(): HIP events around every launch on the
    launch stream (the library's own profile scopes, as bench.py's kernels_us) -- the op-level time next to it also holds
    torch's allocator, the dispatcher and, for backward stages, autograd's bookkeeping"""
    import ctypes as C
    L = K.lib()
    fn()
    torch.cuda.synchronize()
    L.ngm_profile_reset()
    L.ngm_profile_enable(1)
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    L.ngm_profile_enable(0)
    ms, n = C.c_double(0), C.c_int64(0)
    L.ngm_profile_read(K.KERNEL_IDS[kid], C.byref(ms), C.byref(n))
    return (ms.value / max(n.value, 1)) * 1e3, int(n.value) // iters


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rays", type=int, default=262144)
    ap.add_argument("--samples", type=int, default=128)
    ap.add_argument("--out", default=None)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    N, S = a.rays, a.samples
    F, R = 64, N // 64
    res = dict(device=torch.cuda.get_device_name(0), rays=N, samples_per_ray=S, stages={})

    def add(name, secs, bytes_=None, flops=None, note="", kern=None):
        d = dict(us=round(secs * 1e6, 1), note=note)
        if kern is not None:          # (us per launch, launches per call): the kernel alone
            kus, per_call = kern
            d.update(kernel_us=round(kus * per_call, 1), launches_per_call=per_call)
            if bytes_ is not None:
                d.update(kernel_GBps=round(bytes_ / (kus * per_call * 1e-6) / 1e9, 1),
                         kernel_frac_of_hbm_peak=round(bytes_ / (kus * per_call * 1e-6) / 1e9 / HBM_PEAK_GBS, 3))
            if flops is not None:
                d.update(kernel_TFLOPs=round(flops / (kus * per_call * 1e-6) / 1e12, 1),
                         kernel_frac_of_f32_mfma_peak=round(flops / (kus * per_call * 1e-6) / 1e12 / MFMA_F32_PEAK_TF, 3))
        if bytes_ is not None:
            d.update(algorithmic_MB=round(bytes_ / 1e6, 1), GBps=round(bytes_ / secs / 1e9, 1),
                     frac_of_hbm_peak=round(bytes_ / secs / 1e9 / HBM_PEAK_GBS, 3))
        if flops is not None:
            d.update(algorithmic_GFLOP=round(flops / 1e9, 2), TFLOPs=round(flops / secs / 1e12, 1),
                     frac_of_f32_mfma_peak=round(flops / secs / 1e12 / MFMA_F32_PEAK_TF, 3))
        res["stages"][name] = d

    # ---- compositor (k_composite_fwd / k_composite_bwd): 24 B/sample in + 36 B/ray out; bwd 24 B + 36 B/ray in, 16 B out
    rc = K.render_cfg(geometry_mode="nrgbd", geometry_factor=20.0)
    colors = torch.rand(N, S, 3, device=dev)
    geoms = 0.1 * torch.randn(N, S, device=dev)
    dists = torch.sort(torch.rand(N, S, device=dev) * 3 + 0.5, -1)[0]
    depths = dists * 0.9
    add("composite_fwd", timeit(lambda: ops.quadrature(rc, colors, geoms, dists, depths)), bytes_=N * S * 24 + N * (36 + 4 * S),
        note="k_composite_fwd incl. the (N,S) weights output", kern=kernel_us(lambda: ops.quadrature(rc, colors, geoms, dists, depths), "composite_fwd"))
    cg, gg = colors.clone().requires_grad_(), geoms.clone().requires_grad_()
    Cc, D, _, _, T, _ = ops.quadrature(rc, cg, gg, dists, depths)
    seeds = (torch.randn_like(Cc), torch.randn_like(D), torch.randn_like(T))

    def comp_bwd():
        torch.autograd.grad((Cc, D, T), (cg, gg), seeds, retain_graph=True)
    add("composite_bwd", timeit(comp_bwd), bytes_=N * S * 40 + N * 36, note="k_composite_bwd via autograd (includes torch's grad bookkeeping)",
        kern=kernel_us(comp_bwd, "composite_bwd"))
    del colors, geoms, depths, cg, gg, Cc, D, T

    # ---- sampler (k_sample_rays): ~(104 + 4S)/S B read + 16 B written per sample
    rcs = K.render_cfg(num_samples_coarse=S // 2, num_samples_guided=S // 2)
    ijs = torch.stack([torch.randint(0, 480, (F, R), device=dev), torch.randint(0, 640, (F, R), device=dev)], -1)
    near = torch.rand(F, R, device=dev) + 0.5
    far = near + 2.0
    gt = near + 2.0 * torch.rand(F, R, device=dev)
    add("sample_rays", timeit(lambda: ops.sample_rays(rcs, ijs, near, far, gt, seed=1)), bytes_=N * S * 16 + N * 44,
        note="k_sample_rays, in-kernel Philox jitter, rank merge of the two strata; includes the output allocation",
        kern=kernel_us(lambda: ops.sample_rays(rcs, ijs, near, far, gt, seed=1), "sampler"))
    B = 32
    edges = torch.sort(torch.rand(F, R, B + 1, device=dev) * 4 + 0.2, -1)[0]
    wts = torch.softmax(4 * torch.randn(F, R, B, device=dev), -1)
    rcw = K.render_cfg(num_samples_coarse=S, num_samples_guided=0)
    add("sample_rays_weighted", timeit(lambda: ops.sample_rays_weighted(rcw, ijs, edges, wts, seed=1)),
        bytes_=N * S * 16 + N * (12 + 8 * B + 4),
        note="k_sample_rays_weighted (camera.py:277-289): 32 bins per ray, two Philox draws per sample, sequential search of the fp64 running sum",
        kern=kernel_us(lambda: ops.sample_rays_weighted(rcw, ijs, edges, wts, seed=1), "sampler"))

    # ---- field evaluation (encode + MLP), forward and backward on flat points
    fc = K.field_cfg(encoding="fourier", dim_enc=64, num_layers=2)
    P = N * S // F // 8                       # 1/8 of the samples: the output tensors are (F,P,4)
    params = {}
    for n, shp in K.param_shapes(fc).items():
        params[n] = (0.3 * torch.randn(F, *shp, device=dev)).requires_grad_()
    pts = torch.rand(F, P, 3, device=dev)
    pos = torch.zeros(F, 3, device=dev)
    quat = torch.zeros(F, 4, device=dev)
    quat[:, 0] = 1
    nsamp = F * P
    with torch.no_grad():
        add("field_eval_fwd", timeit(lambda: ops.field_eval(fc, params, pts, pos, quat)), flops=nsamp * 16896,
            bytes_=nsamp * 28, note="k_field_points_fwd: Fourier(64) + 2x64 MLP, 16 896 flop/sample, mlp_matmul f32 (exact-fp32 MFMA)",
            kern=kernel_us(lambda: ops.field_eval(fc, params, pts, pos, quat), "points_fwd"))
        fca = K.field_cfg(encoding="fourier", dim_enc=64, num_layers=2, matmul_mode="auto")
        add("field_eval_fwd_auto", timeit(lambda: ops.field_eval(fca, params, pts, pos, quat)), flops=nsamp * 16896,
            bytes_=nsamp * 28, note="the same with mlp_matmul auto (what the renderer uses): exact three-way bf16 split on the bf16 matrix pipe; "
            "algorithmic fp32 flops against the fp32 MFMA peak",
            kern=kernel_us(lambda: ops.field_eval(fca, params, pts, pos, quat), "points_fwd"))
    out = ops.field_eval(fc, params, pts, pos, quat)
    go = torch.randn_like(out)

    def fbwd():
        torch.autograd.grad(out, list(params.values()), go, retain_graph=True)
    add("field_eval_bwd", timeit(fbwd, iters=10), flops=nsamp * 33792, bytes_=nsamp * 28,
        note="point-mode backward (k_field_bwd16, recomputes the forward), 33 792 algorithmic flop/sample")
    # the training pair with mlp_matmul auto (what the renderer's models use; round 6, ABI 11): the forward also writes the
    # activation stash (512 B / sample), the backward is k_field_bwd_b3 in point mode reading it back
    fca = K.field_cfg(encoding="fourier", dim_enc=64, num_layers=2, matmul_mode="auto")
    add("field_eval_fwd_train_auto", timeit(lambda: ops.field_eval(fca, params, pts, pos, quat)), flops=nsamp * 16896, bytes_=nsamp * (28 + 512),
        note="ngm_field_eval_fwd_train: the forward under autograd, mlp_matmul auto, + 512 B / sample of activation stash written",
        kern=kernel_us(lambda: ops.field_eval(fca, params, pts, pos, quat), "points_fwd"))
    out_a = ops.field_eval(fca, params, pts, pos, quat)

    def fbwd_a():
        torch.autograd.grad(out_a, list(params.values()), go, retain_graph=True)
    add("field_eval_bwd_auto", timeit(fbwd_a, iters=10), flops=nsamp * 33792, bytes_=nsamp * (28 + 512),
        note="ngm_field_eval_bwd_stash: point-mode backward with mlp_matmul auto = k_field_bwd_b3 (bf16 split, reads the forward's "
             "activation stash, no recompute) + k_grad_reduce; 33 792 algorithmic flop/sample against the fp32 MFMA peak",
        kern=kernel_us(fbwd_a, "field_bwd"))
    del out, out_a, go, params, pts
    # ---- hash encode stage (SURVEY 8d "Hash encode gathers": 16 levels x 4 vertices x 2 features x 4 B = 512 B of table
    # gathers per sample): the reference's default field (hash 16 x 2 + 1 x 32 MLP, config/neural_graph_map.yaml:6-20) evaluated on
    # flat points; the MLP adds 2 304 flop per sample (1.4 % of the kernel's instructions), so this IS the encode stage's time
    fch = K.field_cfg(encoding="permuto", num_layers=1, nr_levels=16, log2_hashmap_size=12, coarsest_scale=1.0, finest_scale=1e-4)
    Fh, Ph = 8, N * S // 8 // 8
    ph = {}
    for n, shp in K.param_shapes(fch).items():
        ph[n] = 0.3 * torch.randn(Fh, *shp, device=dev)
    ptsh = torch.rand(Fh, Ph, 3, device=dev) * 1.6 - 0.8
    posh = torch.zeros(Fh, 3, device=dev)
    quath = torch.zeros(Fh, 4, device=dev)
    quath[:, 0] = 1
    with torch.no_grad():
        add("hash_encode_fwd", timeit(lambda: ops.field_eval(fch, ph, ptsh, posh, quath)), bytes_=Fh * Ph * 512,
            note="k_field_points_fwd<hash>: permutohedral simplex search + 64 gathers of 8 B per sample (8 fields x 512 KB tables: "
                 "L2-resident) + 1x32 MLP; algorithmic gather bytes against the HBM peak as SURVEY 8d prices them")
    res["stages"]["hash_encode_fwd"]["samples"] = Fh * Ph
    with torch.no_grad():
        add("hash_encode_only", timeit(lambda: ops.encode(fch, ph, ptsh, posh, quath)), bytes_=Fh * Ph * 512,
            note="ngm_encode_fwd (k_encode_points): the encoding alone, 32 features per sample written to HBM (128 B / sample on top "
                 "of the gathers; real HBM traffic and L2 hit rate: a rocprofv3 --pmc pass over this script)")
    res["stages"]["hash_encode_fwd"]["gather_GBps_vs_L2_gather_microbench_2200"] = round(
        res["stages"]["hash_encode_fwd"]["GBps"] / 2200.0, 3)
    deh = torch.randn(Fh, Ph, fch.dim_enc, device=dev)
    with torch.no_grad():
        add("hash_encode_bwd", timeit(lambda: ops.encode_bwd(fch, ph, ptsh, deh, posh, quath), iters=10), bytes_=Fh * Ph * 512,
            note="ngm_encode_bwd: transposition of d_enc to the level-major stream + k_hash_grad (+ k_hash_reduce): 512 B of table "
                 "scatter per sample as SURVEY 8d prices it (the kernel accumulates in LDS: Q23.40 integer atomics, deterministic)")
    del ph, ptsh, deh
    ptsf = torch.rand(8, N * S // 8 // 8, 3, device=dev) * 1.6 - 0.8
    pf = {n: 0.3 * torch.randn(8, *shp, device=dev) for n, shp in K.param_shapes(fc).items()}
    def_ = torch.randn(8, ptsf.shape[1], fc.dim_enc, device=dev)
    with torch.no_grad():
        add("fourier_encode_bwd", timeit(lambda: ops.encode_bwd(fc, pf, ptsf, def_, posh, quath), iters=10),
            bytes_=8 * ptsf.shape[1] * (12 + 4 * (fc.dim_enc - 3)),
            note="ngm_encode_bwd (Fourier): reads the points and d_enc once (12 + 244 B / sample), 61 cosines per sample")
    del ptsf, pf, def_
    # ---- M2: render only, no_grad, 4096 rays x 128 samples, eval-style single stratum (ngm_render_fwd
    # without targets / stash), through the reference-shaped render_ijs
    from neural_graph_mapping_amd import models as M
    from neural_graph_mapping_amd import renderer as Rr
    F2, R2, S2 = 8, 512, 128
    model = M.NeuralFieldSet(dim_points=3, field_type="neural_graph_mapping.models.NeuralField", field_kwargs=dict(
        encoding_type="neural_graph_mapping.positional_encodings.PositionalEncodingFourier",
        encoding_kwargs=dict(dim_in=3, dim_out=64, mu=0.0, sigma=4.0, raw_coords=True), num_layers=2, dim_out=4),
        num_knn=2, distance_factor=10.0, outside_value=1.0, field_radius=1.0, scale_mode="unit_cube").to(dev)
    cam = Rr.Camera(640, 480, 554.2562584220408, 554.2562584220408, 319.5, 239.5, pixel_center=0.0)
    r = Rr.NeuralGraphRenderer(model, cam, Rr.shipped_config(num_samples_coarse=S2, num_samples_depth_guided=0), device=dev)
    r.add_fields(F2)
    r.set_field_poses(torch.zeros(F2, 3, device=dev), torch.tensor([[1.0, 0, 0, 0]], device=dev).repeat(F2, 1))
    ijs2 = torch.stack([torch.randint(0, 480, (F2, R2), device=dev), torch.randint(0, 640, (F2, R2), device=dev)], -1)
    c2w = torch.eye(4, device=dev)
    c2w[2, 3] = 2.5
    near2, far2 = torch.full((F2, R2), 1.5, device=dev), torch.full((F2, R2), 3.5, device=dev)
    ids = torch.arange(F2, device=dev)
    with torch.no_grad():
        secs = timeit(lambda: r.render_ijs(ijs2, c2w, field_ids=ids, use_vmap=True, near_distances=near2, far_distances=far2, seed=3))
    add("M2_render_ijs_nograd", secs, flops=F2 * R2 * S2 * 16896,
        note="8 fields x 512 rays x 128 samples, render only (k_render_fwd without stash), incl. the Python layer of render_ijs")
    res["stages"]["M2_render_ijs_nograd"]["ray_samples_per_s"] = round(F2 * R2 * S2 / secs / 1e6, 1) * 1e6
    print(json.dumps(res))
    if a.out:
        os.makedirs(os.path.dirname(a.out), exist_ok=True)
        json.dump(res, open(a.out, "w"), indent=1)


if __name__ == "__main__":
    main()
