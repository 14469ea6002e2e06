"""Device memory runs out: the boundary reports PST_ERR_OUT_OF_MEMORY (the Rust reference aborts on a failed allocation; a C ABI must not), the
failure does not leak into the next call, and memory the library has freed can be handed back to the other allocators of the process.
Found by tools/exp_oom.py at the end of round 6: a failed allocation left HIP's per-thread last error set and the NEXT call (any call that checks
its launches) reported "out of memory"; blocks of destroyed buffers stayed in the stream-ordered pool (release threshold = never), out of reach of
hipMalloc -- torch could not allocate memory that hipMemGetInfo's caller had long freed."""
import numpy as np
import pytest

from pasture_amd import las
from pasture_amd._capi import PastureError
from pasture_amd.algorithms import calculate_bounds, compute_normals, voxelgrid_filter
from pasture_amd.buffers import HashMapBuffer, VectorBuffer
from pasture_amd.conversion import BufferLayoutConverter
from pasture_amd.layout import PointLayout, attributes as A

pytestmark = pytest.mark.gpu
GIB = 1 << 30


def _free():
    import torch
    torch.cuda.synchronize()
    return torch.cuda.mem_get_info()[0]


def _small_round_trip():
    layout = PointLayout.from_attributes([A.POSITION_3D])
    src = HashMapBuffer.new_from_layout(layout)
    src.resize(100_003)
    src.synth_fill(42, 0)
    out = BufferLayoutConverter.for_layouts(layout, layout).convert(src, VectorBuffer)
    a = src.get_attribute_range(A.POSITION_3D, range(0, 100_003))
    assert out.get_attribute_range(A.POSITION_3D, range(0, 100_003)).tobytes() == a.tobytes()
    b = calculate_bounds(out)
    assert b.min() == tuple(a.min(axis=0)) and b.max() == tuple(a.max(axis=0))


def test_a_failed_allocation_is_a_status_and_does_not_poison_the_next_call(hip):
    layout = PointLayout.from_attributes([A.POSITION_3D])
    b = HashMapBuffer.new_from_layout(layout)
    b.resize(1000)
    b.synth_fill(1, 0)
    before = b.get_attribute_range(A.POSITION_3D, range(0, 1000)).tobytes()
    with pytest.raises(PastureError, match=r"status 22.*out of memory"):
        b.resize(200_000_000_000)  # 4.8 TB
    assert b.len() == 1000 and b.get_attribute_range(A.POSITION_3D, range(0, 1000)).tobytes() == before  # the buffer is what it was
    for count in (1 << 61, (1 << 64) // 24 + 1, (1 << 64) - 1):  # count * 24 wraps around 2^64: a SMALL allocation if nobody looks (Vec::resize: "capacity overflow")
        for buf in (b, VectorBuffer.new_from_layout(layout)):
            with pytest.raises(PastureError, match=r"status 22.*capacity overflow"):
                buf.resize(count)
    assert b.len() == 1000
    _small_round_trip()  # (this call failed with "hipGetLastError(): out of memory" before the fix)


def test_release_scratch_hands_the_pool_back(hip):
    layout = PointLayout.from_attributes([A.POSITION_3D])
    hip.release_scratch()
    f0 = _free()
    if f0 < 20 * GIB:
        pytest.skip("needs 20 GiB of free HBM")
    big = HashMapBuffer.new_from_layout(layout)
    big.resize(400_000_000)  # 9.6 GB
    f1 = _free()
    del big
    f2 = _free()
    assert f0 - f1 > 8 * GIB and f0 - f2 > 8 * GIB  # the pool keeps the block for the library's next allocation ...
    hip.release_scratch()
    assert f0 - _free() < 1 * GIB  # ... until it is asked to give it back


def test_full_device_every_allocating_call_reports_out_of_memory_and_recovers():
    """In a process of its own, which fills the device and then calls everything that allocates.  Whether those calls are REFUSED cannot be arranged
    from outside while another process (the suite's own) holds memory on the device: the driver then refuses 12 MiB to torch's hipMalloc and still
    serves 2.4 GB to the library's hipMallocAsync (three whole-suite runs ended here: 1754 passed, this one "did not raise", although the same body
    alone on a box is refused every time).  So the body checks what holds either way -- whatever ran is right, a refusal is a PastureError, everything
    works once the memory is back -- and is STRICT (status 22 on every refusal, the 2.4-GB request and at least one more call refused) only under
    PST_STRICT_OOM=1, which tools/r06_calls/r06_gpu66.sh sets for the stand-alone run."""
    import os
    import subprocess
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    prog = f"import sys; sys.path.insert(0, {os.path.dirname(here)!r}); sys.path.insert(0, {here!r}); import test_out_of_memory as t; t._full_device_body(); print('BODY-OK')"
    r = subprocess.run([sys.executable, "-c", prog], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "BODY-OK" in r.stdout, (r.stdout[-1500:], r.stderr[-3000:])


def _full_device_body():
    import torch
    from pasture_amd import product_api
    hip = product_api()
    layout = PointLayout.from_attributes([A.POSITION_3D])
    n = 4_000_000
    cloud = HashMapBuffer.new_from_layout(layout)
    cloud.resize(n)
    cloud.synth_fill(42, 0)
    pts = HashMapBuffer.new_from_layout(las.point_layout_from_las_point_format(las.Format(0), False))
    pts.resize(n)
    pts.synth_fill(7, 0)
    mask = (torch.arange(n, device="cuda") % 2 == 0).to(torch.uint8)
    want_normals = compute_normals(cloud, 16)
    # fill the device: one block for all but 64 MiB, then 12-MiB blocks until the driver refuses (hipMemGetInfo's "free" and what an allocation can
    # still get differ by tens of MiB); before that, whatever earlier tests of this process left in the pool or in Python's garbage is handed back --
    # a block freed into the pool after the fill would serve the library's next request
    import gc
    gc.collect()
    hip.release_scratch()
    torch.cuda.empty_cache()
    import os
    strict = os.environ.get("PST_STRICT_OOM") == "1"
    print("strict" if strict else "tolerant", f"({_free() / GIB:.0f} GiB free before the fill)")
    hog = [torch.empty(max(1, _free() - 64 * (1 << 20)), dtype=torch.uint8, device="cuda")]
    try:
        while len(hog) < 64:
            hog.append(torch.empty(12 * (1 << 20), dtype=torch.uint8, device="cuda"))
    except torch.OutOfMemoryError:
        pass
    try:
        if strict:
            with pytest.raises(PastureError, match=r"status 22.*out of memory"):
                HashMapBuffer.new_from_layout(layout).resize(100_000_000)  # 2.4 GB

        # requests of tens of MiB: the stream-ordered pool may still serve them when the driver refuses a plain allocation of 12 MiB (seen in a process
        # that had run the > 4 GiB tests before; a refused request seems to leave what it gathered in the pool).  The contract is "the right result, or
        # status 22" -- never another status, a wrong result, or a failure that the NEXT call reports
        def right_or_oom(call):  # the call's result (checked after the device has room again: reading a result back allocates too), or None
            try:
                return call()
            except PastureError as e:
                assert not strict or ("status 22" in str(e) and "out of memory" in str(e)), str(e)
                return None
        vox = HashMapBuffer.new_from_layout(layout)
        under_pressure = {
            "convert": right_or_oom(lambda: BufferLayoutConverter.for_layouts(layout, layout).convert(cloud, VectorBuffer)),
            "filter": right_or_oom(lambda: pts.filter(HashMapBuffer, (mask.data_ptr(), "device"))),
            "normals": right_or_oom(lambda: compute_normals(cloud, 16)),
            "voxels": right_or_oom(lambda: (voxelgrid_filter(cloud, 2.5, 2.5, 2.5, vox), vox)[1]),
        }
        assert not strict or any(v is None for v in under_pressure.values())  # (compute_normals needs ~ 55 bytes per point of scratch: 220 MB)
    finally:
        del hog
        torch.cuda.empty_cache()
    _small_round_trip()
    got = compute_normals(cloud, 16)
    assert np.array_equal(got[0], want_normals[0]) and np.array_equal(got[1], want_normals[1])
    out = HashMapBuffer.new_from_layout(layout)
    voxelgrid_filter(cloud, 2.5, 2.5, 2.5, out)
    kept = pts.filter(HashMapBuffer, (mask.data_ptr(), "device"))
    assert out.len() > 0 and kept.len() == n // 2
    # whatever did get its memory under pressure must be right as well
    xyz = cloud.get_attribute_range(A.POSITION_3D, range(0, n)).tobytes()
    r = under_pressure["convert"]
    assert r is None or r.get_attribute_range(A.POSITION_3D, range(0, n)).tobytes() == xyz
    r = under_pressure["filter"]
    assert r is None or (r.len() == n // 2 and r.get_attribute_range(A.INTENSITY, range(0, n // 2)).tobytes() == kept.get_attribute_range(A.INTENSITY, range(0, n // 2)).tobytes())
    r = under_pressure["normals"]
    assert r is None or (np.array_equal(r[0], want_normals[0]) and np.array_equal(r[1], want_normals[1]))
    r = under_pressure["voxels"]
    assert r is None or (r.len() == out.len() and r.get_attribute_range(A.POSITION_3D, range(0, out.len())).tobytes() == out.get_attribute_range(A.POSITION_3D, range(0, out.len())).tobytes())
