"""Process-group bootstrap and small collective helpers: one process per GPU, ``torch.distributed`` over RCCL
(backend ``"nccl"`` on ROCm) across xGMI, gloo on hosts without a GPU.

Replaces the reference's oneCCL bootstrap (python/llm/src/ipex_llm/transformers/pipeline_parallel.py:108-112
``init_pipeline_parallel`` and the DeepSpeed / mpirun environment handling of its TP examples,
python/llm/example/GPU/Deepspeed-AutoTP/deepspeed_autotp.py).

xGMI is point-to-point (7 links per GPU, no switch): small latency-bound messages (the [tokens, hidden] all-reduce of
a decode step) are dominated by per-collective launch + ring latency, so decode-time tensor parallelism wants few,
fused collectives; large gradient buckets are bandwidth-bound per link, so buckets are sized in tens of MB.
"""
from __future__ import annotations

import datetime
import os
from typing import Optional

import torch
import torch.distributed as dist

DEFAULT_BUCKET_MB = 64


def _env_int(*names, default=None):
    for n in names:
        v = os.environ.get(n)
        if v is not None:
            return int(v)
    return default


def get_rank_world_from_env():
    """RANK / WORLD_SIZE from torchrun, or the MPI launchers' variables (mpirun setups of the reference)."""
    rank = _env_int("RANK", "PMI_RANK", "OMPI_COMM_WORLD_RANK", default=0)
    world = _env_int("WORLD_SIZE", "PMI_SIZE", "OMPI_COMM_WORLD_SIZE", default=1)
    local = _env_int("LOCAL_RANK", "MPI_LOCALRANKID", "OMPI_COMM_WORLD_LOCAL_RANK", default=rank)
    return rank, world, local


def init_distributed(backend: Optional[str] = None, timeout_s: int = 1800):
    """Idempotent ``init_process_group``.  Returns (rank, world_size, device)."""
    rank, world, local = get_rank_world_from_env()
    use_gpu = torch.cuda.device_count() > 0 and os.environ.get("BIGDL_FORCE_CPU") != "1"
    if use_gpu:
        torch.cuda.set_device(local % torch.cuda.device_count())
        device = torch.device("cuda", local % torch.cuda.device_count())
    else:
        device = torch.device("cpu")
    if not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # dmabuf IPC for RCCL peer buffers
        backend = backend or ("nccl" if use_gpu else "gloo")
        kw = {}
        if backend == "nccl":
            kw["device_id"] = device
        dist.init_process_group(backend=backend, rank=rank, world_size=world,
                                timeout=datetime.timedelta(seconds=timeout_s), **kw)
        if world > 1 and os.environ.get("BIGDL_NO_CONTROL_GROUP") != "1":
            control_group()          # every rank passes here: the side group exists before anyone needs it
    return dist.get_rank(), dist.get_world_size(), device


def staggered_wait(local_rank: Optional[int] = None):
    """``RANK_WAIT_TIME`` seconds (default 0) times the local rank before a rank starts loading its copy of the
    checkpoint, so that the ranks of one host do not hold full-precision state dicts in host memory at the same time
    (reference: example/GPU/Deepspeed-AutoTP/deepspeed_autotp.py:69-73)."""
    import time
    wait = float(os.environ.get("RANK_WAIT_TIME", 0))
    if wait > 0:
        if local_rank is None:
            local_rank = get_rank_world_from_env()[2]
        time.sleep(wait * local_rank)


def rank() -> int:
    return dist.get_rank() if dist.is_initialized() else 0


def world_size() -> int:
    return dist.get_world_size() if dist.is_initialized() else 1


def new_groups(tp: int = 1, pp: int = 1):
    """Split the world into a (dp, pp, tp) grid, tp fastest (TP peers are neighbours on the xGMI mesh).
    Returns dict(tp=group, pp=group, dp=group, coords=(dp_idx, pp_idx, tp_idx))."""
    world, me = world_size(), rank()
    assert world % (tp * pp) == 0, "world size must be divisible by tp * pp"
    dp = world // (tp * pp)
    grid = torch.arange(world).view(dp, pp, tp)
    out = {}
    for name, dim in (("tp", 2), ("pp", 1), ("dp", 0)):
        lines = grid.movedim(dim, -1).reshape(-1, grid.shape[dim])
        mine = None
        for ranks in lines.tolist():
            g = dist.new_group(ranks) if world > 1 else None
            if me in ranks:
                mine = g
        out[name] = mine
    pos = (grid == me).nonzero()[0].tolist()
    out["coords"] = tuple(pos)
    return out


def all_reduce_(t: torch.Tensor, group=None):
    if world_size() > 1:
        dist.all_reduce(t, group=group)
    return t


def all_reduce_coalesced_(tensors, group=None, bucket_mb: int = DEFAULT_BUCKET_MB, average: bool = False):
    """Sum (or average) a list of same-dtype tensors in flat buckets: few large collectives instead of one per
    tensor (LoRA training has hundreds of small gradients)."""
    if world_size() <= 1 or not tensors:
        return tensors
    n = dist.get_world_size(group)
    limit = bucket_mb * 1024 * 1024
    bucket, size = [], 0

    def flush():
        nonlocal bucket, size
        if not bucket:
            return
        flat = torch.cat([t.reshape(-1) for t in bucket])
        dist.all_reduce(flat, group=group)
        if average:
            flat.div_(n)
        off = 0
        for t in bucket:
            t.copy_(flat[off:off + t.numel()].view_as(t))
            off += t.numel()
        bucket, size = [], 0

    by_dtype = {}
    for t in tensors:
        by_dtype.setdefault(t.dtype, []).append(t)
    for group_t in by_dtype.values():
        for t in group_t:
            nb = t.numel() * t.element_size()
            if size + nb > limit:
                flush()
            bucket.append(t)
            size += nb
        flush()
    return tensors


_CONTROL = {}       # default group id -> gloo side group


def control_group():
    """A gloo group over the same ranks as the default group, for host-side control messages (pickled requests, stop
    flags, shapes).  On an RCCL job ``broadcast_object_list`` would otherwise stage every pickle through the GPU: a
    device allocation, two collectives and a stream synchronisation per message, queued behind whatever compute the
    stream holds.  Created on first use by ALL ranks together (call it once at start-up, e.g. right after
    ``init_distributed``); on a gloo job the default group already is one."""
    if world_size() <= 1:
        return None
    if dist.get_backend() == "gloo":
        return dist.group.WORLD
    key = id(dist.group.WORLD)
    if key not in _CONTROL:
        _CONTROL[key] = dist.new_group(backend="gloo", timeout=datetime.timedelta(seconds=1800))
    return _CONTROL[key]


def broadcast_object(obj, src=0, group=None):
    """Pickle-broadcast over the control (gloo) group unless a group is given."""
    if world_size() <= 1:
        return obj
    box = [obj]
    dist.broadcast_object_list(box, src=src, group=group if group is not None else control_group(),
                               device=torch.device("cpu") if group is None else None)
    return box[0]


def barrier():
    if world_size() > 1:
        dist.barrier()
