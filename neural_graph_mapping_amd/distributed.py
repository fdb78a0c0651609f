"""Field-per-GPU sharding of the train step.

Each field's rays touch only that field's parameters (the vmap axis of models.py:342-344) and loop
closure moves only poses (rm.py:936-952), so fields are partitioned by ``owner = field_id % world``.
Every rank runs the (cheap, seeded) target sampler identically and keeps the slice of the Target it
owns; the ONLY data-path collective is a sum all-reduce of the 16-float loss sum/count vector
between the fused forward and the fused backward (RCCL over xGMI: backend "nccl" on ROCm; "gloo" in
the CPU tests).  No ray or activation ever crosses GPUs.
"""
import os
from typing import Optional

import torch
import torch.distributed as dist


def field_owner(field_ids: torch.Tensor, world_size: int) -> torch.Tensor:
    """Rank that owns each field (fields are appended monotonically, rm.py:327-336)."""
    return field_ids % world_size


def owned_mask(field_ids: torch.Tensor, rank: int, world_size: int) -> torch.Tensor:
    return field_owner(field_ids, world_size) == rank


def shard_target(target, rank: int, world_size: int):
    """Keep the rows (fields) of a Target namedtuple owned by `rank`.  May return F == 0."""
    keep = owned_mask(target.field_ids, rank, world_size)
    vals = []
    for name, v in zip(target._fields, target):
        if isinstance(v, torch.Tensor) and v.dim() >= 1 and v.shape[0] == target.field_ids.shape[0]:
            vals.append(v[keep])
        else:
            vals.append(v)
    return type(target)(*vals)


def local_field_slots(num_fields: int, rank: int, world_size: int) -> torch.Tensor:
    """Global ids of the fields stored on this rank, in local slot order (slot = id // world)."""
    return torch.arange(rank, num_fields, world_size)


def global_to_local(field_ids: torch.Tensor, world_size: int) -> torch.Tensor:
    return field_ids // world_size


def allreduce_loss_sums(loss_sums: torch.Tensor, group=None) -> torch.Tensor:
    """Sum the per-rank loss sums / counts in place.  Ranks without active fields pass zeros but
    MUST still call this."""
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(loss_sums, op=dist.ReduceOp.SUM, group=group)
    return loss_sums


def loss_values_from_sums(s: torch.Tensor, w_term, w_photo, w_depth, w_fs, w_tsdf, photometric_loss="l1",
                          depth_loss="huber") -> dict:
    """rm.py:1803-1871 from the global sums (slot layout: include/ngm_hip.h ngm_loss_slot); the photometric / depth
    keys carry the mode like rm.py:1827, 1837 (the sums already are of |e| or e^2, whichever the kernels were told)."""
    def mean(num, den, scale=1.0):
        return torch.where(den > 0, num / (scale * den.clamp_min(1.0)), torch.zeros_like(num))
    pk, dk = "photometric_" + photometric_loss, "depth_" + depth_loss
    out = {pk: mean(s[0], s[1], 3.0), dk: mean(s[2], s[3]), "freespace": mean(s[4], s[5]),
           "tsdf": mean(s[6], s[7]), "termination": mean(s[8], s[9])}
    out["combined"] = (w_term * out["termination"] + w_photo * out[pk] + w_depth * out[dk]
                       + w_fs * out["freespace"] + w_tsdf * out["tsdf"])
    return out


def init_from_env(backend: Optional[str] = None):
    """torchrun-style rendezvous (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_*); returns (rank, local, world)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    force = os.environ.get("NGM_FORCE_DIST", "0") == "1"      # exercise the RCCL path with a single rank
    if (world > 1 or force) and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            torch.cuda.set_device(local)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, local, world


# ------------------------------------------------------------------------------------------------
# evaluation path: rays cross several fields (kNN blend), so every rank needs every field
# ------------------------------------------------------------------------------------------------
def gather_field_params(local_params: dict, num_fields: int, group=None) -> dict:
    """All-gather the stacked parameter dictionaries of a field-per-GPU sharded map into the global order.

    Rank r stores field g = slot * world + r at local slot `slot` (local_field_slots); the result has row g for
    every g < num_fields on every rank.  One all_gather per parameter tensor (106 MB for 200 hash fields, 7 MB
    for Fourier fields); ranks with fewer fields pad their block."""
    world = dist.get_world_size(group) if (dist.is_available() and dist.is_initialized()) else 1
    if world == 1:
        return {k: v[:num_fields] for k, v in local_params.items()}
    per_rank = (num_fields + world - 1) // world
    out = {}
    for k, v in local_params.items():
        pad = torch.zeros((per_rank,) + tuple(v.shape[1:]), dtype=v.dtype, device=v.device)
        pad[: min(per_rank, v.shape[0])] = v[:per_rank]
        blocks = [torch.empty_like(pad) for _ in range(world)]
        dist.all_gather(blocks, pad, group=group)
        # blocks[r][slot] is field slot * world + r  ->  interleave
        out[k] = torch.stack(blocks, 1).reshape((per_rank * world,) + tuple(v.shape[1:]))[:num_fields].contiguous()
    return out


def pixel_shard(num_pixels: int, rank: int, world_size: int):
    """Contiguous slice [begin, end) of the flattened image this rank renders."""
    per = (num_pixels + world_size - 1) // world_size
    return min(num_pixels, rank * per), min(num_pixels, (rank + 1) * per)


def gather_image(local_rows: torch.Tensor, num_pixels: int, group=None, dst: int = 0) -> Optional[torch.Tensor]:
    """Collect the per-rank pixel slices (pixel_shard order) on rank `dst`; other ranks get None."""
    world = dist.get_world_size(group) if (dist.is_available() and dist.is_initialized()) else 1
    if world == 1:
        return local_rows
    per = (num_pixels + world - 1) // world
    pad = torch.zeros((per,) + tuple(local_rows.shape[1:]), dtype=local_rows.dtype, device=local_rows.device)
    pad[: local_rows.shape[0]] = local_rows
    # gather to ONE rank: only `dst` allocates the image (an all_gather would hand every rank a copy it drops)
    me = dist.get_rank(group)
    blocks = [torch.empty_like(pad) for _ in range(world)] if me == dst else None
    dist.gather(pad, blocks, dst=dist.get_global_rank(group, dst) if group is not None else dst, group=group)
    if me != dst:
        return None
    return torch.cat(blocks)[:num_pixels]


def render_image_sharded(renderer, c2w, num_fields: int, camera=None, group=None, dst: int = 0, u=None, seed: int = 0):
    """render_image (rm.py:402-437) on a field-per-GPU sharded map: all-gather the field parameters once, every
    rank renders its slice of the pixels with the kNN-blended evaluation kernels, rank `dst` receives the image."""
    cam = camera or renderer._camera
    rank = dist.get_rank(group) if (dist.is_available() and dist.is_initialized()) else 0
    world = dist.get_world_size(group) if (dist.is_available() and dist.is_initialized()) else 1
    full = gather_field_params(renderer._model.all_fields_params, num_fields, group)
    b, e = pixel_shard(cam.height * cam.width, rank, world)
    rgbd, dvar = renderer.render_pixels(c2w, b, e, params=full, camera=cam, u=u, seed=seed)
    img = gather_image(torch.cat([rgbd, dvar[:, None]], -1), cam.height * cam.width, group, dst)
    if img is None:
        return None, None
    return img[:, :4].reshape(cam.height, cam.width, 4), img[:, 4].reshape(cam.height, cam.width)
