"""Multi-GPU plumbing of the hot path: reads are independent for a fixed kit, so a
batch is sharded by contiguous read ranges, one process per GPU, with NO data-path collective; the
only exchange is one all-reduce (SUM, int64) of the per-barcode / per-kit count vector.

`torch.distributed` is used purely as the RCCL front end (backend "nccl" on the GPU box; "gloo" in
the CPU tests).  Nothing here touches the kernels.
"""
import numpy as np


def shard_range(n_items, rank, world_size):
    """Contiguous [begin, end) of rank `rank`; sizes differ by at most one, order preserved."""
    if world_size < 1 or not (0 <= rank < world_size):
        raise ValueError("bad rank/world_size: {}/{}".format(rank, world_size))
    base, extra = divmod(n_items, world_size)
    begin = rank * base + min(rank, extra)
    return begin, begin + base + (1 if rank < extra else 0)


class _DevArray(object):
    """Minimal __cuda_array_interface__ view of device memory owned by the native library."""

    def __init__(self, ptr, n, typestr="<i8"):
        self.__cuda_array_interface__ = {"shape": (n,), "typestr": typestr, "data": (int(ptr), False),
                                         "version": 2, "strides": None}


def allreduce_counts(counts, dist=None, device=None):
    """Sum an int64 count vector over all ranks.  `counts`: numpy array (host) or a
    (device_pointer, n) pair describing the library's device-resident count vector.  Returns a
    numpy int64 array with the global counts (every rank gets the same result)."""
    import torch
    if dist is None:
        import torch.distributed as dist
    if isinstance(counts, tuple):
        ptr, n = counts
        try:
            t = torch.as_tensor(_DevArray(ptr, n), device=device or "cuda").clone()
        except Exception:            # pragma: no cover - depends on the torch build
            raise RuntimeError("cannot view the native count vector as a torch tensor")
    else:
        t = torch.from_numpy(np.ascontiguousarray(counts, dtype=np.int64).copy())
        if device is not None:
            t = t.to(device)
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return t.cpu().numpy()
