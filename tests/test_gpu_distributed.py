"""Multi-rank runs of the PRODUCT renderer on the GPU (-m gpu): two processes share GPU 0 (the test boxes have one
GPU; gloo moves the device tensors of the 64-byte loss all-reduce and of the evaluation gathers), fields sharded
`owner = id % world` (rm.py:1383-1427).  Checked against the reference fixtures and against the
one-process run of the same renderer:
  * eager `optimization_iteration` with `process_group` set on `shard_target` slices: loss, prediction, union of grads;
  * `capture_iteration` (two hipGraphs around the all-reduce): trained parameters after 5 updates;
  * a rank none of whose fields is active (idle rank) still completes the collective;
  * `render_image_sharded` vs `render_image` / G9.
"""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import load_golden, split_prefix

pytestmark = pytest.mark.gpu

from gpu_common import CASES, DEV, close, grad_close, make_renderer, make_target  # noqa: E402
from neural_graph_mapping_amd import distributed as D  # noqa: E402
from neural_graph_mapping_amd import renderer as Rr  # noqa: E402

NAME = "g6_train_3field"
N_REPLAY = 3


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _init(rank, world, port):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK="0")
    torch.cuda.set_device(0)
    D.init_from_env(backend="gloo")


def _local_renderer(g, rank, world, active=None):
    """Renderer holding only this rank's fields (local slot = id // world) + its slice of the fixture's Target."""
    fkw, ckw = CASES[NAME]
    F = g["pos"].shape[0]
    ids = torch.arange(F)
    own = D.owned_mask(ids, rank, world)
    r = make_renderer(fkw, ckw, int(own.sum()), {k: v[own] for k, v in split_prefix(g, "p::").items()})
    r.set_field_poses(g["pos"][own].to(DEV), g["quat"][own].to(DEV))
    r.process_group = dist.group.WORLD
    t = dict(split_prefix(g, "t::"))
    t["field_ids"] = ids
    t["u_coarse"], t["u_guided"] = g["u_coarse"], g["u_guided"]
    if active is not None:
        t = {k: v[active] for k, v in t.items()}
    tgt_all = make_target(t, t["field_ids"])
    tgt = D.shard_target(tgt_all, rank, world)                     # the rows (fields) this rank owns
    keep = D.owned_mask(t["field_ids"], rank, world)
    tgt = tgt._replace(field_ids=D.global_to_local(tgt.field_ids, world))
    return r, tgt, t["u_coarse"][keep].to(DEV), t["u_guided"][keep].to(DEV), t["field_ids"][keep]


def _train_worker(rank, world, port, out):
    _init(rank, world, port)
    g = load_golden(NAME)
    r, tgt, uc, ug, gids = _local_renderer(g, rank, world)
    res = r.optimization_iteration(tgt, uc, ug, update=False)
    rec = dict(ids=gids, loss={k: v.cpu() for k, v in res.items() if k not in ("grads", "prediction")},
               grads={k: v.cpu().clone() for k, v in res["grads"].items()}, rgbds=res["prediction"].rgbds.cpu().clone())
    # captured iteration: 2 warm-up updates + N_REPLAY replays, explicit draws so that the trajectory is comparable
    replay = r.capture_iteration(tgt, u_coarse=uc, u_guided=ug)
    for _ in range(N_REPLAY):
        last = replay()
    torch.cuda.synchronize()
    rec.update(graphs=replay.graph is not None, step=r._step, step_dev=int(r._step_dev.item()),
               last_loss=last["combined"].cpu(), params={k: v.cpu().clone() for k, v in r._model.all_fields_params.items()})
    torch.save(rec, os.path.join(out, f"train{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


def _single_process(g, n_updates, active=None):
    fkw, ckw = CASES[NAME]
    F = g["pos"].shape[0]
    r = make_renderer(fkw, ckw, F, split_prefix(g, "p::"))
    r.set_field_poses(g["pos"].to(DEV), g["quat"].to(DEV))
    t = dict(split_prefix(g, "t::"))
    ids = torch.arange(F)
    uc, ug = g["u_coarse"], g["u_guided"]
    if active is not None:
        t, ids, uc, ug = {k: v[active] for k, v in t.items()}, ids[active], uc[active], ug[active]
    tgt = make_target(t, ids)
    first = r.optimization_iteration(tgt, uc.to(DEV), ug.to(DEV), update=False)
    first = dict(loss=first["combined"].cpu(), grads={k: v.cpu().clone() for k, v in first["grads"].items()},
                 rgbds=first["prediction"].rgbds.cpu().clone())
    last = None
    for _ in range(n_updates):
        last = r.optimization_iteration(tgt, uc.to(DEV), ug.to(DEV), update=True)
    return first, last, {k: v.cpu().clone() for k, v in r._model.all_fields_params.items()}


def test_two_ranks_product_renderer_eager_and_captured(tmp_path):
    world = 2
    mp.spawn(_train_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    g = load_golden(NAME)
    ref_g, ref_loss = split_prefix(g, "g::"), split_prefix(g, "loss::")
    first, last, params1 = _single_process(g, 2 + N_REPLAY)
    seen = []
    for rank in range(world):
        res = torch.load(os.path.join(tmp_path, f"train{rank}.pt"))
        ids = res["ids"]
        seen += ids.tolist()
        # eager, against the reference fixture (global loss on every rank, this rank's rows of every gradient) ...
        for k, v in ref_loss.items():
            close(res["loss"][k], v, rtol=2e-4, atol=1e-5)
        close(res["rgbds"], g["pred_rgbds"][ids])
        for k, v in ref_g.items():
            scale = v.abs().max().clamp_min(1e-12)
            assert float((res["grads"][k] - v[ids]).abs().max() / scale) < 2e-3, k
            # ... and against the one-process run of the same kernels (same arithmetic, only the loss sums are
            # accumulated in a different order: per rank, then across ranks)
            assert float((res["grads"][k] - first["grads"][k][ids]).abs().max() / scale) < 1e-5, k
        assert res["graphs"], "the sharded iteration was not captured into hipGraphs (fell back to eager launches)"
        assert res["step"] == 2 + N_REPLAY == res["step_dev"]
        close(res["last_loss"], last["combined"], rtol=1e-4, atol=1e-6)
        for k, v in res["params"].items():
            close(v, params1[k][ids], rtol=1e-4, atol=1e-6)
    assert sorted(seen) == list(range(g["pos"].shape[0]))


def _idle_worker(rank, world, port, out):
    _init(rank, world, port)
    g = load_golden(NAME)
    active = torch.tensor([0, 2])                                   # both owned by rank 0: rank 1 has nothing to do
    r, tgt, uc, ug, gids = _local_renderer(g, rank, world, active)
    assert tgt.ijs.shape[0] == (2 if rank == 0 else 0)
    res = r.optimization_iteration(tgt, uc, ug, update=True)
    torch.cuda.synchronize()
    torch.save(dict(loss=res["combined"].cpu(), step=r._step, step_dev=int(r._step_dev.item())),
               os.path.join(out, f"idle{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


def test_idle_rank_completes_the_collective(tmp_path):
    world = 2
    mp.spawn(_idle_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    g = load_golden(NAME)
    first, _, _ = _single_process(g, 0, active=torch.tensor([0, 2]))
    for rank in range(world):
        res = torch.load(os.path.join(tmp_path, f"idle{rank}.pt"))
        close(res["loss"], first["loss"], rtol=1e-5, atol=1e-6)       # the idle rank reports the same global loss
        assert res["step"] == 1 == res["step_dev"]                    # and keeps the shared Adam step counter


# ------------------------------------------------------------------------------------------ evaluation path
def _g9_renderer(g, own=None):
    fkw = dict(encoding="fourier", dim_enc=64, num_layers=2)
    ckw = dict(num_samples_coarse=8, num_samples_depth_guided=16, eval_far_distance=float(g["eval_far"]),
               eval_num_samples=int(g["eval_num_samples"]))
    p = split_prefix(g, "p::")
    if own is not None:
        p = {k: v[own] for k, v in p.items()}
    NF = next(iter(p.values())).shape[0]
    r = make_renderer(fkw, ckw, NF, p)
    r.set_field_poses(g["pos"].to(DEV), g["quat"].to(DEV))          # poses are replicated on every rank
    r.eval()                                                         # rm.py:1978: evaluation switches the sampling parameters
    w, h, fx, fy, cx, cy = [float(x) for x in g["cam"]]
    return r, Rr.Camera(int(w), int(h), fx, fy, cx, cy, pixel_center=0.0)


def _eval_worker(rank, world, port, out):
    _init(rank, world, port)
    g = load_golden("g9_render_image")
    NF = g["pos"].shape[0]
    r, cam = _g9_renderer(g, D.local_field_slots(NF, rank, world))
    rgbd, dvar = D.render_image_sharded(r, g["c2w"].to(DEV), NF, camera=cam, u=g["u"].to(DEV))
    assert (rgbd is None) == (rank != 0)
    if rank == 0:
        torch.save(dict(rgbd=rgbd.cpu(), dvar=dvar.cpu()), os.path.join(out, "image.pt"))
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_render_image_sharded(tmp_path):
    world = 2
    mp.spawn(_eval_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    g = load_golden("g9_render_image")
    res = torch.load(os.path.join(tmp_path, "image.pt"))
    close(res["rgbd"], g["rgbd"], rtol=5e-4, atol=5e-5)               # the reference's render_image
    close(res["dvar"], g["dvar"], rtol=5e-4, atol=5e-5)
    r, cam = _g9_renderer(g)
    rgbd, dvar = r.render_image(g["c2w"].to(DEV), cam, u=g["u"].to(DEV))
    assert torch.equal(res["rgbd"], rgbd.cpu()) and torch.equal(res["dvar"], dvar.cpu())   # pixels are independent


# ------------------------------------------------------------------------------------------ one-shot peer exchange
def _expected_sum(vs):
    """the kernel's order: 0 + v_0 + v_1 + ... in fp32"""
    tot = torch.zeros(16)
    for v in vs:
        tot = tot + v
    return tot


def _peer_worker(rank, world, port, out):
    _init(rank, world, port)
    px = D.PeerExchange(dist.group.WORLD)                 # set-up incl. the self-test exchange
    vals = lambda it, r: torch.randn(16, generator=torch.Generator().manual_seed(1000 * it + r)) * 10.0 ** (it % 5 - 2)  # noqa: E731
    eager = []
    for it in range(40):                                  # plain launches; both parities, many sequence numbers
        x = vals(it, rank).to(DEV)
        px.allreduce(x)
        eager.append(x.cpu())
    # captured: a graph of (copy the staged input, exchange) replayed with fresh inputs
    stage, buf = torch.zeros(16, device=DEV), torch.zeros(16, device=DEV)
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        buf.copy_(stage)
        px.allreduce(buf)                                 # warm-up outside the capture keeps the ranks' counts equal
    torch.cuda.current_stream().wait_stream(s)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        buf.copy_(stage)
        px.allreduce(buf)
    replayed = []
    for it in range(40, 60):
        stage.copy_(vals(it, rank))
        graph.replay()
        replayed.append(buf.cpu())
    status = px.status()
    torch.save(dict(eager=eager, replayed=replayed, status=status), os.path.join(out, f"peer{rank}.pt"))
    dist.barrier()
    px.close()
    dist.destroy_process_group()


def test_peer_exchange_two_ranks_bitwise(tmp_path):
    """ngm_loss_exchange (csrc/ngm_peer.hip) between two processes sharing GPU 0 through hipIpc-mapped mailboxes: every
    exchange returns, on BOTH ranks, exactly 0 + v_0 + v_1 (rank order, fp32) -- eager launches and hipGraph replays
    (the sequence number lives on the device), no time-out."""
    world = 2
    mp.spawn(_peer_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    res = [torch.load(os.path.join(tmp_path, f"peer{r}.pt")) for r in range(world)]
    vals = lambda it, r: torch.randn(16, generator=torch.Generator().manual_seed(1000 * it + r)) * 10.0 ** (it % 5 - 2)  # noqa: E731
    for r in range(world):
        assert res[r]["status"] == 0
        for it in range(40):
            assert torch.equal(res[r]["eager"][it], _expected_sum([vals(it, q) for q in range(world)])), (r, it)
        for j, it in enumerate(range(40, 60)):
            assert torch.equal(res[r]["replayed"][j], _expected_sum([vals(it, q) for q in range(world)])), (r, it)


def _train_peer_worker(rank, world, port, out):
    _init(rank, world, port)
    from neural_graph_mapping_amd import _capi
    # (the compositing backward inside the MLP backward reads the EXCHANGED sums here; single-GPU runs take them from the
    # forward's partials)
    g = load_golden(NAME)
    r, tgt, uc, ug, gids = _local_renderer(g, rank, world)
    r.peer_exchange = D.PeerExchange(dist.group.WORLD)
    replay = r.capture_iteration(tgt, u_coarse=uc, u_guided=ug)
    for _ in range(N_REPLAY):
        last = replay()
    torch.cuda.synchronize()
    one_graph = isinstance(replay.graph, torch.cuda.CUDAGraph)
    rec = dict(ids=gids, one_graph=one_graph, status=r.peer_exchange.status(), step=r._step, step_dev=int(r._step_dev.item()),
               fused=_capi.lib().ngm_debug_last_comp_fused(), variant=_capi.lib().ngm_debug_last_bwd_variant(),
               last_loss=last["combined"].cpu(), params={k: v.cpu().clone() for k, v in r._model.all_fields_params.items()})
    torch.save(rec, os.path.join(out, f"trainpx{rank}.pt"))
    dist.barrier()
    r.peer_exchange.close()
    dist.destroy_process_group()


def test_two_ranks_peer_exchange_inside_one_graph(tmp_path):
    """The sharded iteration with `peer_exchange` set: forward, loss exchange and backward + Adam in ONE hipGraph per rank
    (no host-side collective between replays); trained parameters after 5 updates = the one-process run."""
    world = 2
    mp.spawn(_train_peer_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    g = load_golden(NAME)
    _, last, params1 = _single_process(g, 2 + N_REPLAY)
    for rank in range(world):
        res = torch.load(os.path.join(tmp_path, f"trainpx{rank}.pt"))
        assert res["one_graph"] and res["status"] == 0
        assert res["variant"] == 3 and res["fused"] == 1              # two launches + the exchange kernel per iteration
        assert res["step"] == 2 + N_REPLAY == res["step_dev"]
        close(res["last_loss"], last["combined"], rtol=1e-4, atol=1e-6)
        for k, v in res["params"].items():
            close(v, params1[k][res["ids"]], rtol=1e-4, atol=1e-6)


def _peer_fail_worker(rank, world, port, out, where):
    """one rank's set-up breaks at `where`; every rank must come back with None, nobody may hang in a collective"""
    _init(rank, world, port)
    from neural_graph_mapping_amd import _capi as K
    L = K.lib()
    if rank == 1:                    # sabotage THIS rank only: the failure is asymmetric
        if where == "alloc":
            real = L.ngm_peer_alloc
            L.__dict__["ngm_peer_alloc"] = lambda *a: -4
        elif where == "open":
            real = L.ngm_ipc_open
            L.__dict__["ngm_ipc_open"] = lambda *a: -4
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        px = D.PeerExchange.try_create(dist.group.WORLD, torch.device(DEV))
    # the process group is still in step afterwards: a collective issued by everyone pairs up
    x = torch.ones(1)
    dist.all_reduce(x)
    torch.save(dict(px_is_none=px is None, seen=float(x)), os.path.join(out, f"fail_{where}{rank}.pt"))
    dist.destroy_process_group()


@pytest.mark.parametrize("where", ["alloc", "open"])
def test_peer_exchange_setup_survives_an_asymmetric_failure(tmp_path, where):
    """ADVICE (round 3): a set-up that fails on a subset of ranks used to leave the ranks with unequal collective counts
    (gathers issued from the exception path) and could hang the fall-back to RCCL.  The set-up now runs the same three
    gathers on every rank whatever fails where: `try_create` returns None on BOTH ranks and the next collective pairs."""
    world = 2
    mp.spawn(_peer_fail_worker, args=(world, _free_port(), str(tmp_path), where), nprocs=world, join=True)
    for r in range(world):
        res = torch.load(os.path.join(tmp_path, f"fail_{where}{r}.pt"))
        assert res["px_is_none"] and res["seen"] == world


def _peer_timeout_worker(rank, world, port, out):
    _init(rank, world, port)
    px = D.PeerExchange(dist.group.WORLD, timeout_s=2.0)  # the default is 30 s (an all-reduce would wait for ever)
    assert px.timeout_s == 2.0
    x = torch.full((16,), float(rank + 1), device=DEV)
    px.allreduce(x)                                       # a healthy exchange
    torch.cuda.synchronize()
    healthy = (px.status(), float(x[0]))
    raised = None
    if rank == 0:
        # rank 1 does NOT enter the next exchange: rank 0 waits ~2 s for it, gives up, and must say so
        y = torch.ones(16, device=DEV)
        px.allreduce(y)
        torch.cuda.synchronize()
        try:
            px.check()
        except RuntimeError as e:
            raised = str(e)
        partial = bool(torch.isnan(y).all())
        # the exchange is dead on this rank now: later launches do not wait out another time-out each (a dead peer
        # used to stall the survivors for peer_check_interval x time-out), they return NaN at once
        import time
        t0 = time.perf_counter()
        for _ in range(20):
            z = torch.ones(16, device=DEV)
            px.allreduce(z)
        torch.cuda.synchronize()
        fast_after = (time.perf_counter() - t0, bool(torch.isnan(z).all()))
    else:
        partial = fast_after = None
    dist.barrier()
    torch.save(dict(healthy=healthy, raised=raised, status=px.status(), partial=partial, fast_after=fast_after),
               os.path.join(out, f"timeout{rank}.pt"))
    px.close()
    dist.destroy_process_group()


def test_peer_exchange_timeout_is_reported_not_swallowed(tmp_path):
    """ADVICE (rounds 3, 4): a peer that does not arrive within the (configurable, default 30 s) time-out poisons the sums of
    that exchange with NaN -- loss and update of the iteration are NaN at once, never silently mis-normalised -- and it is fatal: `PeerExchange.check()` (called by the renderer every `peer_check_interval` iterations and by
    `check_exchange()`) raises on the rank that timed out."""
    world = 2
    mp.spawn(_peer_timeout_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    r0, r1 = (torch.load(os.path.join(tmp_path, f"timeout{r}.pt")) for r in range(world))
    assert r0["healthy"] == (0, 3.0) and r1["healthy"] == (0, 3.0)
    assert r0["status"] & 1 and r0["raised"] and "loss exchange failed" in r0["raised"]
    assert r0["partial"] is True                         # the sums of a timed-out exchange are NaN: the iteration is visibly dead
    assert r0["fast_after"][1] and r0["fast_after"][0] < 1.0     # 20 further exchanges: NaN without another 20 x 2 s of waiting
    assert r1["status"] == 0


def test_bench_line_with_two_ranks_runs_to_the_end():
    """bench.py --gpus 2 as the driver launches it (two ranks, here sharing the one GPU over gloo): every rank must issue the
    same sequence of collectives from the first spin-up step to the last profiling step.  Rounds 4-5 read the shader clock by
    running 1 500 more iterations on rank 0 ALONE -- each of them holds the loss all-reduce, so the other rank's next collective
    met the wrong partner (gloo: size mismatch, abort; RCCL: a hang at the end of the run).  No 1-GPU run could notice."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, NGM_BENCH_SHARE_GPU="1", NGM_DIST_BACKEND="gloo")
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--windows", "2",
                        "--min-seconds", "0", "--no-cpu-baseline"], capture_output=True, text=True, env=env, timeout=600, cwd=root)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads(r.stdout.strip().split("\n")[-1])
    assert line["n_gpus"] == 2 and line["config"]["ranks_seen"] == 2 and line["steps"] == 3 and line["value"] > 0
