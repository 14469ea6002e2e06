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


class PipelinedBoundsReduce:
    """The same all-reduce for a stream of steps, overlapped with compute: step i writes its local {min xyz, max xyz} into
    `current()`, `submit()` starts the reduction (MIN of the minima, MAX of the maxima) ASYNCHRONOUSLY (the collective runs on the backend's own stream
    behind an event of the compute stream), and the next step's kernels are launched without waiting for it.  A ring of
    `depth` record buffers keeps a record alive until its all-reduce has finished; `finish()` returns the last global record."""

    def __init__(self, make_record, depth: int = 4, group=None):
        self.recs = [make_record() for _ in range(depth)]
        self.work = [None] * depth
        self.group = group
        self.i = 0

    def current(self):
        b = self.i % len(self.recs)
        if self.work[b] is not None:  # the buffer's previous all-reduce (depth steps ago) must be complete before reuse
            for w in self.work[b]:
                w.wait()
            self.work[b] = None
        return self.recs[b]

    def exposed_us(self):
        return None  # (torch's own collectives: no event pair of ours around them)

    def submit(self, timed: bool = False) -> None:
        import torch.distributed as dist
        b = self.i % len(self.recs)
        rec = self.recs[b]
        # MIN over the three minima + MAX over the three maxima: two 24-byte collectives on the backend's stream and NOTHING
        # on the compute stream (the [min, -max] encoding of allreduce_bounds_record costs two extra kernels there)
        self.work[b] = (dist.all_reduce(rec[:3], op=dist.ReduceOp.MIN, group=self.group, async_op=True),
                        dist.all_reduce(rec[3:], op=dist.ReduceOp.MAX, group=self.group, async_op=True))
        self.i += 1

    def wait(self) -> None:
        """Waits for every pending all-reduce (what a timed region ends with; `finish()` afterwards only hands out the record)."""
        for k in range(len(self.recs)):
            if self.work[k] is not None:
                for w in self.work[k]:
                    w.wait()
                self.work[k] = None

    def finish(self):
        """Waits for every pending all-reduce; returns the global record of the last submitted step (decoded), or None."""
        last = None
        self.wait()
        if self.i:
            last = self.recs[(self.i - 1) % len(self.recs)]
        return last


class CapiTransport:
    """The boundary's own collective as the transport of BoundsExchange: `pst_comm_init_rank` bootstrapped from the launcher's
    rendezvous (rank 0's 128-byte id broadcast over the torch group), then `pst_bounds_allreduce` per record -- ONE
    ncclAllReduce(6 x f64, ncclMin) over {min, -max}, in place, ordered on the stream it is given."""
    name = "pst_bounds_allreduce (C ABI: one ncclAllReduce of 6 x f64 with ncclMin over {min, -max}, RCCL)"
    # BoundsExchange registers its ring records with pst_bounds_record_set_form: the kernels that produce them write {min, -max} themselves and the
    # exchange is the collective ALONE (round 6; before: a one-block negation kernel on either side of it -- three launches of latency per step)
    encoded_records = True

    def __init__(self, group=None, api=None):
        from ._capi import product_api
        self.api = api or product_api()
        self.comm = Communicator.from_torch_group(group, self.api)

    def set_record_form(self, rec, encoded: bool) -> None:
        import ctypes as C
        self.api.bounds_record_set_form(C.c_void_p(rec.data_ptr()), 1 if encoded else 0)

    def size(self) -> int:
        return self.comm.size()

    def allreduce(self, rec, stream_handle=None, restore_handle=None) -> None:
        import ctypes as C
        if stream_handle is not None:  # the library launches on the calling thread's stream: point it at the exchange stream for this call
            self.api.set_stream(C.c_void_p(stream_handle))
        try:
            self.comm.allreduce_bounds(rec.data_ptr())
        finally:
            if stream_handle is not None:
                self.api.set_stream(C.c_void_p(restore_handle))

    def close(self) -> None:
        self.comm.destroy()


class TorchTransport:
    """The same exchange through torch.distributed (RCCL when the backend is "nccl", gloo on CPU): the seam the CPU tests swap in."""
    name = "torch.distributed.all_reduce (MIN over {min, -max})"

    def __init__(self, group=None):
        self.group = group

    def size(self) -> int:
        import torch.distributed as dist
        return dist.get_world_size(self.group)

    def allreduce(self, rec, stream_handle=None, restore_handle=None) -> None:
        allreduce_bounds_record(rec, self.group)

    def close(self) -> None:
        pass


class BoundsExchange:
    """The per-step exchange of the sharded path for a stream of steps: step i writes its local {min xyz, max xyz}
    into `current()`; `submit()` reduces that record in place over the ranks through `transport.allreduce`.  On a GPU the reduction
    is ordered behind the step's kernels by an event and runs on its own high-priority stream, so the next step's kernels are
    launched without waiting for it; a ring of `depth` records keeps a record alive until its reduction has finished, and before a
    record is written again the compute stream waits for that reduction.  `finish()` returns the last global record.
    The transport is the only thing that differs between bench.py on RCCL (CapiTransport) and the gloo tests (TorchTransport)."""

    def __init__(self, make_record, transport, depth: int = 4):
        self.recs = [make_record() for _ in range(depth)]
        self.transport = transport
        self.i = 0
        self._cuda = bool(self.recs[0].is_cuda)
        self._done = [None] * depth
        # records the producing kernels leave as {min, -max} (CapiTransport on a GPU): registered for the ring's lifetime, decoded by finish()
        self.encoded = bool(self._cuda and getattr(transport, "encoded_records", False))
        if self.encoded:
            for rec in self.recs:
                transport.set_record_form(rec, True)
        if self._cuda:
            import torch
            self._main = torch.cuda.current_stream()
            self._side = torch.cuda.Stream(priority=-1)
            # every record goes through the transport once, now (COLLECTIVE: every rank constructs its exchange at the same point): the first
            # reduction of a buffer RCCL has not seen costs tens of milliseconds (measured at one rank: 20 timed steps 2.7 ms each with a cold
            # fourth record, 0.80 with all four warm), and that must not land in a timed region
            for rec in self.recs:
                rec.copy_(torch.tensor([F64_MAX] * 3 + [F64_MAX if self.encoded else -F64_MAX] * 3, dtype=rec.dtype, device=rec.device))
                self._side.wait_stream(self._main)
                self.transport.allreduce(rec, self._side.cuda_stream, self._main.cuda_stream)
            self._side.synchronize()

    def current(self):
        b = self.i % len(self.recs)
        if self._cuda and self._done[b] is not None:
            self._main.wait_event(self._done[b])  # the record's previous reduction (depth steps ago) before it is overwritten
            self._done[b] = None
        return self.recs[b]

    def submit(self, timed: bool = False) -> None:
        """timed: keep a timing event pair around this reduction (`exposed_us()` after `finish()`: from the moment the step's kernels were
        done to the moment the global record was there -- for the LAST step of a run that is the part of the exchange nothing hides)."""
        b = self.i % len(self.recs)
        rec = self.recs[b]
        if self._cuda:
            import torch
            ready = torch.cuda.Event(enable_timing=timed)
            ready.record(self._main)
            self._side.wait_event(ready)
            self.transport.allreduce(rec, self._side.cuda_stream, self._main.cuda_stream)
            done = torch.cuda.Event(enable_timing=timed)
            done.record(self._side)
            self._done[b] = done
            if timed:
                self._timed = (ready, done)
        else:
            self.transport.allreduce(rec)
        self.i += 1

    def wait(self) -> None:
        """Waits for every pending reduction (the exchange stream runs dry): what a timed region ends with.  `finish()` afterwards only decodes."""
        if self._cuda:
            self._side.synchronize()
            self._done = [None] * len(self.recs)

    def finish(self):
        """Waits for every pending reduction; returns the last global record as {min xyz, max xyz} (a decoded copy when the ring's records are kept
        as {min, -max}), or None."""
        self.wait()
        if not self.i:
            return None
        rec = self.recs[(self.i - 1) % len(self.recs)]
        if self.encoded:
            # decoded on the HOST (a copy of 48 bytes): the first launch of a torch elementwise kernel loads its code object -- ten milliseconds
            # that would land inside the caller's timed region
            rec = rec.cpu()
            rec[3:] = -rec[3:]
        return rec

    def close(self) -> None:
        """Forgets the ring's record addresses (a later allocation at the same address must not inherit the {min, -max} form)."""
        if self.encoded:
            for rec in self.recs:
                self.transport.set_record_form(rec, False)
            self.encoded = False

    def __del__(self):
        try:
            self.close()
        except Exception:  # noqa: BLE001
            pass

    def exposed_us(self):
        """Microseconds between "the timed step's kernels are done" and "its global record is there" (None without a timed submit)."""
        t = getattr(self, "_timed", None)
        if t is None:
            return None
        t[1].synchronize()
        return 1e3 * t[0].elapsed_time(t[1])


def verify_global_bounds(local_source_rec, global_result_rec, scale=(1.0, 1.0, 1.0), offset=(0.0, 0.0, 0.0), group=None) -> dict:
    """Self-check of a sharded convert + AABB run, through a path that shares nothing with the exchange under test.

    Every rank contributes its LOCAL SOURCE bounds {min xyz, max xyz} (6 f64, seeds for an empty shard); they are all-gathered with
    torch.distributed (not the AABB all-reduce), their union is the global source AABB, and -- the affine map x -> (x * scale) + offset
    (two roundings) being monotone per component -- bounds(affine(P)) == affine(bounds(P)) EXACTLY, so the all-reduced result record must
    equal affine(union) bit for bit, on every rank, and must be the same on every rank.  Raises AssertionError with the records otherwise;
    returns {"verified": True, "ranks": N, "expected": [...]}.  Used by bench.py at N > 1 (the first multi-GPU runs must not be silently
    wrong) and by the gloo tests."""
    import numpy as np
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    dev = "cuda" if dist.get_backend(group) == "nccl" else "cpu"
    mine = torch.cat([local_source_rec.detach().to(dev, torch.float64).reshape(6), global_result_rec.detach().to(dev, torch.float64).reshape(6)])
    everyone = [torch.zeros(12, dtype=torch.float64, device=dev) for _ in range(world)]
    dist.all_gather(everyone, mine, group=group)
    rows = np.array([t.cpu().numpy() for t in everyone])
    src, res = rows[:, :6], rows[:, 6:]
    src_min, src_max = src[:, :3].min(axis=0), src[:, 3:].max(axis=0)
    empty = bool((src_min == F64_MAX).all() and (src_max == -F64_MAX).all())
    s, o = np.array(scale, dtype=np.float64), np.array(offset, dtype=np.float64)
    if empty:
        expected = np.array([F64_MAX] * 3 + [-F64_MAX] * 3)
    else:
        lo, hi = (src_min * s) + o, (src_max * s) + o  # numpy: a product and a sum, each rounded -- the kernel's two roundings
        expected = np.concatenate([np.where(s >= 0, lo, hi), np.where(s >= 0, hi, lo)])
    same_everywhere = bool((res == res[0]).all())
    ok = same_everywhere and bool((res[0] == expected).all())
    if not ok:
        raise AssertionError(f"global AABB self-check failed over {world} ranks: all-reduced records per rank {res.tolist()}, expected affine(union of the "
                             f"local source bounds) {expected.tolist()}, local source bounds per rank {src.tolist()}")
    return {"verified": True, "ranks": world, "expected": expected.tolist()}


def shard_output_offsets(local_count: int, group=None):
    """Compaction (`filter`) over index-range shards: rank r keeps its matches in order, so its slice of the global result
    starts at the sum of the match counts of ranks < r.  One all-gather of one int64 per rank.  Returns (offset, total)."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    backend = dist.get_backend(group)
    dev = "cuda" if backend == "nccl" else "cpu"
    mine = torch.tensor([int(local_count)], dtype=torch.int64, device=dev)
    counts = [torch.zeros(1, dtype=torch.int64, device=dev) for _ in range(world)]
    dist.all_gather(counts, mine, group=group)
    counts = [int(c.item()) for c in counts]
    rank = dist.get_rank(group)
    return sum(counts[:rank]), sum(counts)


def allreduce_las_header(bounds6, points_by_return, group=None):
    """The LAS writer's header side effects over shards (raw_writers.rs:28-83): header bounds = MIN / MAX of the per-shard
    bounds, points-by-return = SUM of the per-shard histograms.  `bounds6` = [min xyz, max xyz] (6 floats), `points_by_return`
    a list of ints.  Returns the global (bounds6, points_by_return)."""
    import torch
    import torch.distributed as dist
    dev = "cuda" if dist.get_backend(group) == "nccl" else "cpu"
    b = torch.tensor(list(bounds6), dtype=torch.float64, device=dev)
    allreduce_bounds_record(b, group)
    c = torch.tensor(list(points_by_return), dtype=torch.int64, device=dev)
    dist.all_reduce(c, op=dist.ReduceOp.SUM, group=group)
    return [float(x) for x in b.tolist()], [int(x) for x in c.tolist()]


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


class Communicator:
    """pst_comm: the C ABI's own RCCL communicator (include/pasture_amd.h, "multi-GPU"), for hosts that do not run torch.distributed.

    `Communicator.from_unique_id` is the one-process-per-GPU form (rank 0 creates the id with `Communicator.unique_id()` and ships the 128
    bytes to the other ranks over any channel -- `from_torch_group` uses a torch.distributed broadcast for that); `single_process(n)`
    drives GPUs 0..n-1 from one process.  `allreduce_bounds(ptr)` is ONE ncclAllReduce of the 6-double record, in place, stream-ordered."""

    def __init__(self, handle, api):
        self._h, self.api = handle, api

    @staticmethod
    def unique_id(api=None) -> bytes:
        import ctypes as C
        from ._capi import product_api
        api = api or product_api()
        buf = (C.c_uint8 * 128)()
        api.comm_unique_id(buf)
        return bytes(buf)

    @classmethod
    def from_unique_id(cls, n_ranks: int, rank: int, uid: bytes, api=None) -> "Communicator":
        import ctypes as C
        from ._capi import product_api
        api = api or product_api()
        assert len(uid) == 128
        buf = (C.c_uint8 * 128).from_buffer_copy(uid)
        h = C.c_void_p()
        api.comm_init_rank(n_ranks, rank, buf, C.byref(h))
        return cls(h, api)

    @classmethod
    def from_torch_group(cls, group=None, api=None) -> "Communicator":
        """Bootstraps over an existing torch.distributed group (any backend): rank 0's id is broadcast as 128 bytes."""
        import torch
        import torch.distributed as dist
        rank, world = dist.get_rank(group), dist.get_world_size(group)  # group-local rank: the rank of the new communicator
        dev = "cuda" if dist.get_backend(group) == "nccl" else "cpu"  # "cuda" = torch's CURRENT device: the caller has set the rank's GPU
        # 128 id bytes + one flag byte.  Rank 0 ALWAYS takes part in the broadcast: if it cannot create the id (RCCL not loadable is the case
        # bench.py falls back from) it sends flag = 1 and every rank raises AFTER the broadcast, so all ranks leave this function the same
        # way and their next collective (the caller's agreement all-reduce) matches -- raising before the broadcast on rank 0 alone left the
        # other ranks blocked in it.
        t = torch.zeros(129, dtype=torch.uint8, device=dev)
        err0 = None
        if rank == 0:
            try:
                t = torch.tensor(list(cls.unique_id(api)) + [0], dtype=torch.uint8, device=dev)
            except Exception as e:  # noqa: BLE001 -- whatever it is, the other ranks must hear about it
                err0 = e
                t[128] = 1
        # broadcast's `src` is a GLOBAL rank: the group's rank 0 is not global rank 0 for a sub-group
        dist.broadcast(t, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
        raw = bytes(t.cpu().tolist())
        if raw[128]:
            raise RuntimeError(f"pst_comm_unique_id failed on rank 0 of the group: {err0 if err0 is not None else 'see rank 0'}")
        return cls.from_unique_id(world, rank, raw[:128], api)

    @classmethod
    def single_process(cls, n_gpus: int, api=None) -> "Communicator":
        import ctypes as C
        from ._capi import product_api
        api = api or product_api()
        h = C.c_void_p()
        api.comm_init(n_gpus, C.byref(h))
        return cls(h, api)

    def size(self) -> int:
        import ctypes as C
        n = C.c_int()
        self.api.comm_size(self._h, C.byref(n))
        return n.value

    def allreduce_bounds(self, device_rec6_ptr: int) -> None:
        import ctypes as C
        self.api.bounds_allreduce(self._h, C.c_void_p(device_rec6_ptr))

    def allreduce_bounds_multi(self, device_rec_ptrs, streams=None) -> None:
        import ctypes as C
        n = len(device_rec_ptrs)
        recs = (C.c_void_p * n)(*device_rec_ptrs)
        st = (C.c_void_p * n)(*streams) if streams is not None else None
        self.api.bounds_allreduce_multi(self._h, recs, st)

    def destroy(self) -> None:
        if self._h is not None:
            self.api.comm_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.destroy()
        except Exception:
            pass
