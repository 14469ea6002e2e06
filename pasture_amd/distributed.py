"""Multi-GPU: points shard by contiguous index range, one process per GPU; the only exchange is ONE all-reduce of the
6-scalar AABB record (RCCL over xGMI when the backend is "nccl"; "gloo" on CPU for the tests).

The reference has no distributed code.  Its own index-range processing (convert_into_range buffer_conversion.rs:292,
1 MiB chunks raw_readers.rs:309-349) is what makes the path shard without any data-path collective.
"""
from __future__ import annotations

from typing import Optional, Tuple

F64_MAX = 1.7976931348623157e308


def shard_range(n: int, rank: int, world_size: int) -> range:
    """GPU g of G owns [g*ceil(n/G), min(n, (g+1)*ceil(n/G)))  (SURVEY.md 8(e))."""
    per = -(-n // world_size)
    return range(min(n, rank * per), min(n, (rank + 1) * per))


def allreduce_bounds_record(rec, group=None):
    """In-place global AABB of a 6-element f64 tensor {min xyz, max xyz} living on the rank's device.

    Encoded as [min xyz, -max xyz] so that a single MIN all-reduce of count 6 suffices.  Empty shards contribute the
    reference's seeds (+f64::MAX / f64::MIN, bounds.rs:31-32), i.e. the identities.
    """
    import torch.distributed as dist
    rec[3:].neg_()
    dist.all_reduce(rec, op=dist.ReduceOp.MIN, group=group)
    rec[3:].neg_()
    return rec


def bounds_from_record(rec) -> Optional[Tuple[Tuple[float, float, float], Tuple[float, float, float]]]:
    """AABB::from_min_max on the reduced record: None if every shard was empty, panics like math/bounds.rs:21-26 if min > max."""
    from ._capi import ERR_BOUNDS_INVALID, PasturePanic
    v = [float(x) for x in rec.tolist()]
    mn, mx = v[:3], v[3:]
    if all(a == F64_MAX for a in mn) and all(b == -F64_MAX for b in mx):
        return None
    if any(a > b for a, b in zip(mn, mx)):
        raise PasturePanic(ERR_BOUNDS_INVALID, "AABB::from_min_max: Minimum position must be <= maximum position!")
    return tuple(mn), tuple(mx)
