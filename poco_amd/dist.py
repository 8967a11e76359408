"""Multi-GPU data parallelism for the per-crop regressor (new functionality; the reference has no
distributed inference).  One process per GPU; crops are independent, so a batch is cut
into contiguous shards and the only exchange is an all-gather of a packed per-crop record
[rotmat 216 | betas 10 | cam 3 | var 24 | global-uncert 1] = 254 floats (RCCL over xGMI when the
process group's backend is "nccl"; the same code runs on gloo for the CPU tests)."""
from __future__ import annotations

from typing import Dict, List, Tuple

import torch

REC = 254
VIDEO_REC = REC + 98       # video mode: + the 49 raw 2-D joints of the prediction (the finished pose may be a smoothed one)
FIELDS = (("pred_pose", 0, 216), ("pred_shape", 216, 226), ("pred_cam", 226, 229), ("var_pose", 229, 253),
          ("var_global", 253, 254))


def shard_range(n: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous shard [lo, hi) of n crops for `rank`: the first n % world ranks get one extra."""
    base, extra = divmod(n, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def global_confidence(var_pose: torch.Tensor, head: str, kinematic: bool = True, thr: float = 0.40) -> torch.Tensor:
    """The post-processed per-crop scalar of the reference, on the device (same arithmetic as postproc.prepare_uncert
    + postproc.global_uncert = poco_utils.py:21-25,50-60 and the clip of tester.py:245): kinematic accumulation along
    the SMPL tree, rows whose root exceeds the threshold set to 1, CLIFF: root value, PARE: mean over joints."""
    from .synth import SMPL_PARENTS
    var = var_pose.clone()
    if kinematic:
        for i in range(1, 24):
            var[:, i] += var[:, int(SMPL_PARENTS[i])]
    cliff = "cliff" in head
    var[var[:, 0] > (2 * thr if cliff else thr)] = 1.0
    g = var[:, 0] if cliff else var.mean(-1)
    return g.clamp(0.0, 0.99)


def pack_records(out: Dict[str, torch.Tensor], var_global: torch.Tensor | None = None, head: str = "cliff",
                 kinematic: bool = True) -> torch.Tensor:
    """[rotmat 216 | betas 10 | cam 3 | var_pose 24 (raw network output) | var_global 1 (post-processed)]."""
    B = out["pred_shape"].shape[0]
    rec = torch.empty(B, REC, device=out["pred_shape"].device, dtype=torch.float32)
    rec[:, 0:216] = out["pred_pose"].reshape(B, 216)
    rec[:, 216:226] = out["pred_shape"]
    rec[:, 226:229] = out["pred_cam"]
    rec[:, 229:253] = out["var_pose"]
    rec[:, 253] = global_confidence(out["var_pose"], head, kinematic) if var_global is None else var_global
    return rec


def unpack_records(rec: torch.Tensor) -> Dict[str, torch.Tensor]:
    d = {name: rec[:, lo:hi] for name, lo, hi in FIELDS}
    d["pred_pose"] = d["pred_pose"].reshape(-1, 24, 3, 3)
    d["var_global"] = d["var_global"][:, 0]
    return d


def all_gather_records(rec: torch.Tensor, n_total: int, group=None) -> torch.Tensor:
    """Gather the per-rank [n_r, 254] records into [n_total, 254] in crop order (ragged shards are
    padded to the largest shard for the collective and trimmed afterwards)."""
    import torch.distributed as dist
    world = dist.get_world_size(group)
    sizes = [shard_range(n_total, r, world) for r in range(world)]
    nmax = max(hi - lo for lo, hi in sizes)
    send = rec
    if rec.shape[0] < nmax:
        send = torch.zeros(nmax, REC, device=rec.device, dtype=rec.dtype)
        send[: rec.shape[0]] = rec
    buf = torch.empty(world * nmax, REC, device=rec.device, dtype=rec.dtype)
    dist.all_gather_into_tensor(buf, send.contiguous(), group=group)
    parts: List[torch.Tensor] = [buf[r * nmax: r * nmax + (hi - lo)] for r, (lo, hi) in enumerate(sizes)]
    return torch.cat(parts, 0)


def check_collective_buffers(send: torch.Tensor, recv: torch.Tensor, world: int, backend: str, device=None) -> None:
    """What RCCL's all_gather_into_tensor needs, checked before the first collective so that a multi-GPU run can only fail
    on the transport itself: float32, contiguous, recv = world x send rows, and - for the nccl (RCCL) backend - both
    buffers in the HBM of THIS rank's own device (a host tensor or another rank's device would fault or hang inside RCCL)."""
    if send.dtype != torch.float32 or recv.dtype != torch.float32:
        raise TypeError(f"collective buffers must be float32, got {send.dtype} / {recv.dtype}")
    if not (send.is_contiguous() and recv.is_contiguous()):
        raise ValueError("collective buffers must be contiguous")
    if recv.shape[0] != world * send.shape[0] or recv.shape[1:] != send.shape[1:]:
        raise ValueError(f"recv {tuple(recv.shape)} is not world ({world}) x send {tuple(send.shape)}")
    if backend == "nccl":
        dev = torch.device(device) if device is not None else torch.device("cuda", torch.cuda.current_device())
        for name, t in (("send", send), ("recv", recv)):
            if not t.is_cuda or t.device != dev:
                raise ValueError(f"RCCL {name} buffer lives on {t.device}, this rank's device is {dev}")


# ---- video mode: tracks are the unit of sharding (tester.py:368) ------------------------------
def shard_tracks(tracking_results: dict, rank: int, world: int) -> dict:
    """Greedy longest-first assignment of whole tracks to ranks (a track's frames stay together so that temporal
    smoothing needs no exchange); deterministic for a given dict, balanced to within the longest track."""
    order = sorted(tracking_results, key=lambda k: (-len(tracking_results[k]["frames"]), str(k)))
    load = [0] * world
    mine = {}
    for k in order:
        r = min(range(world), key=lambda i: (load[i], i))
        load[r] += len(tracking_results[k]["frames"])
        if r == rank:
            mine[k] = tracking_results[k]
    return mine


def collective_device(group=None) -> torch.device:
    """Where the buffers of a collective must live for the group's backend: RCCL ("nccl") moves device memory,
    gloo host memory."""
    import torch.distributed as dist
    if dist.get_backend(group) == "nccl":
        return torch.device("cuda", torch.cuda.current_device())
    return torch.device("cpu")


def gather_track_records(local: Dict[str, torch.Tensor], group=None, device=None, width: int = REC) -> Dict[str, torch.Tensor]:
    """All ranks end up with every track's [T, width] records.  `local`: {person_id: [T,width] tensor} of this rank
    (width = 254 for the plain SMPL record, 352 = VIDEO_REC when the 49 raw 2-D joints ride along).
    One all-gather of the concatenated rows (padded to the largest rank) + one object gather of the (id, T) index.
    The collective's buffers are allocated on `device` (default: what the group's backend needs), also on a rank that
    owns no track at all (more ranks than people is the common video case)."""
    import torch.distributed as dist
    world = dist.get_world_size(group)
    dev = torch.device(device) if device is not None else collective_device(group)
    ids = sorted(local, key=str)
    index = [(k, int(local[k].shape[0])) for k in ids]
    all_index: List[list] = [None] * world
    dist.all_gather_object(all_index, index, group=group)
    nmax = max(1, max(sum(t for _, t in idx) for idx in all_index))
    send = torch.zeros(nmax, width, device=dev, dtype=torch.float32)
    if ids:
        rows = torch.cat([local[k].to(dev) for k in ids], 0)
        assert rows.shape[1] == width, (rows.shape, width)
        send[: rows.shape[0]] = rows
    buf = torch.empty(world * nmax, width, device=dev, dtype=torch.float32)
    dist.all_gather_into_tensor(buf, send, group=group)
    out = {}
    for r, idx in enumerate(all_index):
        off = r * nmax
        for k, t in idx:
            out[k] = buf[off: off + t].clone()
            off += t
    return out
