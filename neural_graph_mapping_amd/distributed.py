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


def draw_fields_reference(current_field_ids: torch.Tensor, num_fields: int, num_train_fields: int, generator=None):
    """The reference's draw (rm.py:1280-1319): half of `num_train_fields` among the currently observed fields, the rest
    uniformly among all others, union sorted.  -> (field_ids, subset_observed, subset_random)"""
    dev = current_field_ids.device
    n_obs = min(num_train_fields // 2, len(current_field_ids))
    if n_obs > 0:
        sub_obs = torch.multinomial(torch.ones(len(current_field_ids), device=dev), n_obs, generator=generator)
    else:
        sub_obs = torch.empty(0, dtype=torch.int64, device=dev)
    obs_ids = current_field_ids[sub_obs]
    n_rand = min(num_train_fields - len(obs_ids), num_fields - len(obs_ids))
    if n_rand <= 0:
        return obs_ids, sub_obs, None
    w = torch.ones(num_fields, device=dev)
    w[obs_ids] = 0.0
    sub_rand = torch.multinomial(w, n_rand, generator=generator)
    return torch.unique(torch.cat((sub_rand, obs_ids))), sub_obs, sub_rand


def draw_fields_balanced(current_field_ids: torch.Tensor, num_fields: int, num_train_fields: int, world_size: int,
                         generator=None) -> torch.Tensor:
    """Opt-in sampler policy `field_draw: balanced_by_owner` (NOT the reference's distribution): the reference's draw
    made once PER OWNER RANK with a quota of num_train_fields / world (the first `num_train_fields % world` owners get one
    more), over that owner's fields only -- half of the quota among its currently observed fields, the rest uniformly
    among its other fields.  Every rank then trains the same number of fields in every iteration (as far as it owns
    that many), where the reference's global draw puts 7 of 32 on the worst of 8 ranks against a mean of 4.
    Same generator state on every rank -> same set everywhere.  With world_size 1 this is the reference's draw."""
    if world_size <= 1:
        return draw_fields_reference(current_field_ids, num_fields, num_train_fields, generator)[0]
    dev = current_field_ids.device
    chosen = []
    base, extra = divmod(num_train_fields, world_size)
    for owner in range(world_size):
        quota = base + (1 if owner < extra else 0)
        mine = torch.arange(owner, num_fields, world_size, device=dev)
        if quota == 0 or len(mine) == 0:
            continue
        cur_o = current_field_ids[current_field_ids % world_size == owner]
        # local indices (slot = id // world) so that the per-owner draw is the reference's draw on the owner's sub-map
        ids_local, _, _ = draw_fields_reference(cur_o // world_size, len(mine), quota, generator)
        chosen.append(ids_local * world_size + owner)
    return torch.sort(torch.cat(chosen)).values if chosen else torch.empty(0, dtype=torch.int64, device=dev)


def allreduce_loss_sums(loss_sums: torch.Tensor, group=None) -> torch.Tensor:
    """Sum the per-rank loss sums / counts in place.  Ranks without active fields pass zeros but
    MUST still call this."""
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(loss_sums, op=dist.ReduceOp.SUM, group=group)
    return loss_sums


class PeerExchange:
    """The 64-byte loss exchange as ONE small kernel inside the captured iteration instead of an RCCL call between two
    graphs (include/ngm_hip.h, ngm_loss_exchange; csrc/ngm_peer.hip): every rank writes its 16 sums into every rank's
    mailbox (peer memory mapped through hipIpc, xGMI stores), polls its own and sums in rank order -- bit-identical on
    all ranks.  One node, world <= 8.  Set-up is collective over `group` (handles travel by all_gather_object) and ends
    with a self-test exchange; any failure raises, so that a caller can keep the RCCL path (`try_create`).
    Every rank must call `allreduce` the same number of times (idle ranks with zeros)."""

    def __init__(self, group=None, device=None, timeout_s: Optional[float] = None):
        """Collective with a FIXED number of collectives on every rank whatever fails where: (1) handle gather, (2) "opened
        every peer" gather, (3) self-test verdict gather.  A rank that fails locally carries ok = False into the next
        gather instead of leaving the protocol, so no rank is ever left waiting in a gather nobody pairs with.
        `timeout_s`: how long an exchange waits for a peer (process-wide, `ngm_peer_set_timeout`; default 30 s or
        NGM_PEER_TIMEOUT_S) -- a rank that stalls longer (checkpoint I/O, a rank-0-only evaluation) poisons that iteration's
        sums with NaN and raises the sticky status; size it for the longest stall the application has between iterations."""
        import ctypes as C
        from . import _capi as K
        self._K, self._C = K, C
        L = K.lib()
        if timeout_s is not None:
            L.ngm_peer_set_timeout(float(timeout_s))
        self.timeout_s = float(L.ngm_peer_set_timeout(0.0))
        self.world, self.rank = dist.get_world_size(group), dist.get_rank(group)
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        self._opened, self._own, self._state, self._px = [], None, None, None
        err, raw = None, None
        # ---- local set-up; nothing collective in here
        try:
            if self.world > 8:
                raise NotImplementedError("PeerExchange: one node, at most 8 ranks")
            with torch.cuda.device(self.device):
                own = C.c_void_p()
                K.check(L.ngm_peer_alloc(L.ngm_peer_mailbox_bytes(), C.byref(own)), "ngm_peer_alloc")
                self._own = own
                state = C.c_void_p()
                K.check(L.ngm_peer_alloc(64, C.byref(state)), "ngm_peer_alloc")      # seq (8 bytes) + status (4 bytes)
                self._state = state
                handle = C.create_string_buffer(64)
                K.check(L.ngm_ipc_export(own, handle), "ngm_ipc_export")
                raw = handle.raw
        except Exception as e:                                   # noqa: BLE001
            err = e
        # ---- (1) handles; a rank whose allocation / export failed sends None
        handles = [None] * self.world
        dist.all_gather_object(handles, (self.rank, raw), group=group)
        ok = err is None and all(h is not None and h[1] is not None for h in handles)
        if ok:
            try:
                with torch.cuda.device(self.device):
                    px = K.PeerExchange()
                    px.world, px.rank = self.world, self.rank
                    for r, hraw in handles:
                        if r == self.rank:
                            px.mailbox[r] = self._own.value
                        else:
                            ptr = C.c_void_p()
                            K.check(L.ngm_ipc_open(hraw, C.byref(ptr)), f"ngm_ipc_open(rank {r})")
                            self._opened.append(ptr)
                            px.mailbox[r] = ptr.value
                    px.seq = self._state.value
                    px.status = self._state.value + 16
                    self._px = px
            except Exception as e:                               # noqa: BLE001
                err, ok = e, False
        # ---- (2) did every rank map every peer?  only then is the self-test exchange safe to enter (it polls for all ranks)
        opened = [None] * self.world
        dist.all_gather_object(opened, bool(ok), group=group)
        ok = all(opened)
        if ok:
            try:
                with torch.cuda.device(self.device):
                    # self-test: every rank contributes rank + 1 in slot 0 -> world (world + 1) / 2 everywhere
                    probe = torch.zeros(16, device=self.device)
                    probe[0] = self.rank + 1.0
                    self.allreduce(probe)
                    torch.cuda.synchronize(self.device)
                    ok = self.status() == 0 and float(probe[0]) == self.world * (self.world + 1) / 2
            except Exception as e:                               # noqa: BLE001
                err, ok = e, False
        # ---- (3) verdict
        verdict = [None] * self.world
        dist.all_gather_object(verdict, bool(ok), group=group)
        if not all(verdict):
            self.close()
            raise RuntimeError(f"PeerExchange set-up failed (handles {[h is not None and h[1] is not None for h in handles]}, "
                               f"opened {opened}, self-test {verdict}; this rank: {err})")
        self._calls = 0

    @classmethod
    def try_create(cls, group=None, device=None):
        """-> PeerExchange or None (the caller keeps torch.distributed.all_reduce).  Collective: the constructor runs the
        same three gathers on every rank whatever fails, so this either succeeds on every rank or returns None on every
        rank; nothing collective happens in the failure path."""
        try:
            return cls(group, device)
        except Exception as e:                                   # noqa: BLE001 - any failure means "use RCCL"
            import warnings
            warnings.warn(f"PeerExchange unavailable, using the process group's all_reduce: {e}")
            return None

    def allreduce(self, loss_sums: torch.Tensor) -> torch.Tensor:
        """Sum the (16,) float32 device tensor over the ranks, in place, on the current stream (capturable)."""
        assert loss_sums.dtype == torch.float32 and loss_sums.numel() == 16 and loss_sums.is_contiguous()
        K, C = self._K, self._C
        st = torch.cuda.current_stream(self.device).cuda_stream
        K.check(K.lib().ngm_loss_exchange(C.byref(self._px), loss_sums.data_ptr(), st), "ngm_loss_exchange")
        return loss_sums

    def check(self):
        """Raise if any exchange so far timed out (its sums were PARTIAL: the loss normalisers of that iteration were wrong on
        this rank) or found the ranks out of step.  Synchronises with the device; the renderer calls it every
        `peer_check_interval` iterations and whenever it is asked for the exchange's health -- a time-out is fatal for the
        run, exactly as a hung all-reduce would have been, only later."""
        st = self.status()
        if st:
            raise RuntimeError(f"PeerExchange: loss exchange failed (status {st}: bit 0 = a peer did not deliver within "
                               f"{self.timeout_s:g} s and the sums of that iteration were poisoned with NaN, bit 1 = ranks out of "
                               "step); the fields trained in that iteration hold NaN -- restart from a checkpoint, raise the "
                               "time-out (PeerExchange(timeout_s=...)) or use the process group")

    def status(self) -> int:
        """0 = every exchange so far completed; bit 0 = some exchange waited `timeout_s` for a peer and gave up, bit 1 = a slot carried a
        later sequence number (ranks out of step after a time-out); sticky.  Synchronises with the device."""
        from . import hiprt as H
        torch.cuda.synchronize(self.device)
        out = self._C.c_int32(-1)
        with torch.cuda.device(self.device):
            H._chk(H.hip().hipMemcpy(self._C.byref(out), self._px.status, 4, 2), "hipMemcpy D2H")
        return int(out.value)

    def close(self):
        L = self._K.lib()
        for p in self._opened:
            L.ngm_ipc_close(p)
        self._opened = []
        for name in ("_own", "_state"):
            p = getattr(self, name, None)
            if p is not None and p.value:
                L.ngm_peer_free(p)
                setattr(self, name, None)


def loss_values_from_sums(s: torch.Tensor, w_term, w_photo, w_depth, w_fs, w_tsdf, photometric_loss="l1",
                          depth_loss="huber") -> dict:
    """rm.py:1803-1871 from the global sums (slot layout: include/ngm_hip.h ngm_loss_slot); the photometric / depth
    keys carry the mode like rm.py:1827, 1837 (the sums already are of |e| or e^2, whichever the kernels were told)."""
    def mean(num, den, scale=1.0):
        # an empty selection is NaN, as the reference's `.mean()` of an empty tensor (rm.py:1803-1835)
        return torch.where(den > 0, num / (scale * den.clamp_min(1.0)), torch.full_like(num, float("nan")))
    pk, dk = "photometric_" + photometric_loss, "depth_" + depth_loss
    photo = mean(s[0], s[1], 3.0)
    if photometric_loss == "gaussian_nll":                 # losses.py:34-35: the L1 loss whenever the mean NLL exceeds 2
        photo = torch.where(photo > 2.0, mean(s[10], s[1], 3.0), photo)
    zero = torch.zeros_like(s[0])
    out = {pk: photo, dk: mean(s[2], s[3]), "freespace": mean(s[4], s[5]) if w_fs != 0.0 else zero,
           "tsdf": mean(s[6], s[7]) if w_tsdf != 0.0 else zero, "termination": mean(s[8], s[9])}
    out["combined"] = w_term * out["termination"] + w_photo * out[pk] + w_depth * out[dk]
    if w_fs != 0.0:                                        # the terms exist only with a non-zero weight (rm.py:1847-1871)
        out["combined"] = out["combined"] + w_fs * out["freespace"]
    if w_tsdf != 0.0:
        out["combined"] = out["combined"] + w_tsdf * out["tsdf"]
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
