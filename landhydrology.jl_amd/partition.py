"""Block partition of an ensemble of independent columns over ranks (one process
per GPU), and the one collective of the path: the global stable-dt minimum.

The reference has one column and no parallelism; columns are
fully independent, so rank r of W owns the contiguous block
[r*N/W, (r+1)*N/W) with no halo and no data-path collective.  Only the
build-defined adaptive step needs communication: a single `min` all-reduce of
one FT value per step (RCCL over xGMI on GPUs: backend "nccl"; gloo on CPU).
With a user-supplied fixed dt -- what the reference does -- nothing is exchanged
and every rank's results equal the single-process run bit for bit.
"""
from __future__ import annotations

from typing import Tuple


def block_range(ncols_global: int, rank: int, world: int) -> Tuple[int, int]:
    """Half-open column range [lo, hi) owned by `rank`; the first N % W ranks get
    one extra column."""
    if not (0 <= rank < world):
        raise ValueError(f"rank {rank} outside [0, {world})")
    if ncols_global < world:
        raise ValueError("fewer columns than ranks")
    q, r = divmod(ncols_global, world)
    lo = rank * q + min(rank, r)
    return lo, lo + q + (1 if rank < r else 0)


def global_min_dt(local_dt_tensor, group=None):
    """All-reduce (min) of the local stable dt, in place, on whatever device the
    tensor lives on: a device tensor written by `lh_stable_dt_device` stays on the
    GPU (no host round trip).  No-op without an initialised process group.  This is the
    host-owned form of the collective (torch.distributed; the only one available to ranks that
    share a GPU, e.g. a gloo rehearsal); with `attach_native_comm` the library does it itself
    over RCCL and this call is not needed."""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(local_dt_tensor, op=dist.ReduceOp.MIN, group=group)
    return local_dt_tensor


def attach_native_comm(ctx, rank: int, world: int, group=None):
    """Attach the library's own RCCL communicator to this rank's context (lh_comm_init): rank 0
    draws the id (lh_comm_unique_id = ncclGetUniqueId), torch.distributed only ships its 128
    bytes, every rank then joins.  From here on `lh_rhs_stable_dt` / `lh_stable_dt[_device]`
    deliver the GLOBAL minimum -- ncclAllReduce(count = 1, min) enqueued by the library on the
    context's stream -- and the host owns no collective any more.  A Julia host does the same
    with MPI.Bcast for the id (julia/LandHydrologyHIP.jl, `attach_comm!`)."""
    import ctypes as C

    import torch
    import torch.distributed as dist

    from . import _ffi as F
    L = F.lib()
    buf = (C.c_ubyte * F.LH_COMM_ID_BYTES)()
    if rank == 0:
        F.check(L.lh_comm_unique_id(buf), None)
    t = torch.tensor(list(buf), dtype=torch.uint8)
    on_gpu = dist.get_backend(group) == "nccl"
    if on_gpu:
        t = t.cuda()
    dist.broadcast(t, src=0, group=group)
    ident = C.create_string_buffer(bytes(t.cpu().tolist()), F.LH_COMM_ID_BYTES)
    F.check(L.lh_comm_init(ctx, int(rank), int(world), ident), ctx)


def detach_native_comm(ctx):
    from . import _ffi as F
    F.check(F.lib().lh_comm_destroy(ctx), ctx)
