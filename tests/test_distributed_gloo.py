"""world_size-2 check (gloo, CPU) of the multi-GPU design: fields sharded by id % world, ONE
all-reduce of the loss sum/count vector, local backward with the GLOBAL normalisers -> the union of
the per-rank gradients equals the single-process gradients (reference fixture g6_train_3field)."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import load_golden, split_prefix
from neural_graph_mapping_amd import distributed as D
from oracle import ngm_oracle as O


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _local_sums_and_loss(pred, t, rs):
    """differentiable local sums (slot layout of include/ngm_hip.h) for one shard"""
    m = t["depth_mask"] & (pred["term_probs"] > 0.8)
    e = pred["rgbds"][m][:, 3] - t["rgbds"][m][:, 3]
    hub = torch.where(e.abs() < rs.huber_delta, 0.5 * e * e, rs.huber_delta * (e.abs() - 0.5 * rs.huber_delta))
    tm = t["term_mask"]
    tau = rs.truncation_distance
    s = [(t["rgbds"][m][:, :3] - pred["rgbds"][m][:, :3]).abs().sum(), m.sum().float(), hub.sum(), m.sum().float(),
         ((pred["freespace_geometry"] - tau) ** 2).sum(), torch.tensor(float(pred["freespace_geometry"].numel())),
         (pred["tsdf_residuals"] ** 2).sum(), torch.tensor(float(pred["tsdf_residuals"].numel())),
         ((pred["term_probs"][tm] - t["term_probs"][tm]) ** 2).sum(), tm.sum().float()]
    return torch.stack(s + [torch.zeros(())] * 6)


def _worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    D.init_from_env(backend="gloo")
    torch.set_num_threads(1)
    g = load_golden("g6_train_3field")
    fs = O.FieldSpec(encoding="fourier", dim_enc=64, num_layers=2)
    rs = O.RenderSpec(num_samples_coarse=8, num_samples_depth_guided=16, termination_weight=0.5)
    cam = O.CameraSpec(640, 480, 554.2562584220408, 554.2562584220408, 319.5, 239.5)
    t = split_prefix(g, "t::")
    ids = torch.arange(g["pos"].shape[0])
    keep = D.owned_mask(ids, rank, world)
    params = {k: v[keep].clone().requires_grad_() for k, v in split_prefix(g, "p::").items() if k != "_neus_sd"}
    tl = {k: v[keep] for k, v in t.items()}
    pred = O.render_ijs(tl["ijs"], tl["c2ws"], cam, g["pos"][keep], g["quat"][keep], params, fs, rs, tl["near"],
                        tl["far"], tl["gt"], g["u_coarse"][keep], g["u_guided"][keep])
    sums = _local_sums_and_loss(pred, tl, rs)
    glob = D.allreduce_loss_sums(sums.detach().clone())          # the ONLY collective
    w = [rs.photometric_weight / 3.0, rs.depth_weight, rs.freespace_weight, rs.tsdf_weight, rs.termination_weight]
    local = sum(w[i] * sums[2 * i] / glob[2 * i + 1].clamp_min(1.0) for i in range(5))
    local.backward()
    vals = D.loss_values_from_sums(glob, rs.termination_weight, rs.photometric_weight, rs.depth_weight,
                                   rs.freespace_weight, rs.tsdf_weight)
    torch.save(dict(ids=ids[keep], grads={k: v.grad for k, v in params.items()}, loss=vals["combined"], sums=glob),
               os.path.join(out, f"rank{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_field_sharding_matches_single_process(tmp_path):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    g = load_golden("g6_train_3field")
    ref_g = split_prefix(g, "g::")
    ref_loss = split_prefix(g, "loss::")["combined"]
    seen = []
    for r in range(world):
        res = torch.load(os.path.join(tmp_path, f"rank{r}.pt"))
        torch.testing.assert_close(res["loss"], ref_loss, rtol=2e-4, atol=1e-6)
        for k, gr in res["grads"].items():
            ref = ref_g[k][res["ids"]]
            scale = ref_g[k].abs().max().clamp_min(1e-12)
            assert ((gr - ref).abs().max() / scale) < 2e-3, k
        seen += res["ids"].tolist()
    assert sorted(seen) == list(range(g["pos"].shape[0]))
    assert torch.equal(torch.load(os.path.join(tmp_path, "rank0.pt"))["sums"],
                       torch.load(os.path.join(tmp_path, "rank1.pt"))["sums"])


# ------------------------------------------------------------------ evaluation path collectives
def _eval_worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    D.init_from_env(backend="gloo")
    NF, npix = 7, 101                                        # odd counts: ranks hold 4 and 3 fields, 51 and 50 pixels
    glob = {"w": torch.arange(NF * 6, dtype=torch.float32).view(NF, 2, 3), "b": torch.arange(NF, dtype=torch.float32).view(NF, 1)}
    slots = D.local_field_slots(NF, rank, world)
    local = {k: v[slots].clone() for k, v in glob.items()}
    full = D.gather_field_params(local, NF)
    ok = all(torch.equal(full[k], glob[k]) for k in glob)
    b, e = D.pixel_shard(npix, rank, world)
    rows = torch.stack([torch.arange(b, e, dtype=torch.float32), torch.full((e - b,), float(rank))], -1)
    img = D.gather_image(rows, npix, dst=0)
    torch.save(dict(ok=ok, img=img, shard=(b, e)), os.path.join(out, f"eval{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_eval_gather_interleaves_fields_and_pixels(tmp_path):
    world = 2
    mp.spawn(_eval_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    r0, r1 = (torch.load(os.path.join(tmp_path, f"eval{r}.pt")) for r in range(world))
    assert r0["ok"] and r1["ok"]                              # every rank ends up with all fields in global order
    assert r1["img"] is None and r0["img"].shape == (101, 2)
    assert torch.equal(r0["img"][:, 0], torch.arange(101, dtype=torch.float32))
    assert r0["shard"] == (0, 51) and r1["shard"] == (51, 101)
    assert torch.equal(r0["img"][:, 1], torch.cat([torch.zeros(51), torch.ones(50)]))


def _draw_worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    D.init_from_env(backend="gloo")
    NF, FA = 41, 12
    cur = torch.arange(30, 41)
    sets = []
    gen = torch.Generator().manual_seed(99)                      # the same seed on every rank, as for the rest of the sampler
    for _ in range(20):
        ids = D.draw_fields_balanced(cur, NF, FA, world, generator=gen)
        mine = ids[D.owned_mask(ids, rank, world)]
        sets.append((ids, mine))
    # every rank drew the same sets ...
    flat = torch.cat([s[0] for s in sets])
    gathered = [torch.empty_like(flat) for _ in range(world)]
    dist.all_gather(gathered, flat)
    same = all(torch.equal(gathered[0], x) for x in gathered)
    # ... and owns exactly its quota of each
    counts = torch.tensor([len(s[1]) for s in sets])
    torch.save(dict(same=same, counts=counts), os.path.join(out, f"draw{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


def test_balanced_field_draw_two_ranks(tmp_path):
    """the opt-in `balanced_by_owner` draw under two real ranks: identical sets everywhere (same generator state), each rank
    owning exactly num_train_fields / world of every set"""
    world = 2
    mp.spawn(_draw_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    for rank in range(world):
        res = torch.load(os.path.join(tmp_path, f"draw{rank}.pt"))
        assert res["same"]
        assert res["counts"].tolist() == [6] * 20
