"""voxelgrid_filter (pasture-algorithms/src/voxel_grid.rs:109-689).  Known answers: the reference's own test
(:906-941, CompletePoint cloud of 3002 points -> 1000 voxels) and doc-test (:86-108); numpy restatement of the per-attribute
reductions on random data WITHOUT most-common ties (the reference breaks such ties by HashMap iteration order)."""
import numpy as np
import pytest

from harness import BUFFER_KINDS
from pasture_amd._capi import PastureError, PasturePanic
from pasture_amd.algorithms import voxelgrid_filter
from pasture_amd.buffers import HashMapBuffer, VectorBuffer
from pasture_amd.layout import PointLayout, attributes as A

COMPLETE = [A.POSITION_3D, A.INTENSITY, A.RETURN_NUMBER, A.NUMBER_OF_RETURNS, A.CLASSIFICATION_FLAGS, A.SCANNER_CHANNEL, A.SCAN_DIRECTION_FLAG,
            A.EDGE_OF_FLIGHT_LINE, A.CLASSIFICATION, A.SCAN_ANGLE_RANK, A.SCAN_ANGLE, A.USER_DATA, A.POINT_SOURCE_ID, A.COLOR_RGB, A.GPS_TIME,
            A.NIR]  # CompletePoint, voxel_grid.rs:700-746 (repr(C, packed))


def complete_point_cloud(layout, seed=0):
    """setup_point_cloud, voxel_grid.rs:755-904."""
    rng = np.random.default_rng(seed)
    n = 3002
    rec = np.zeros(n, dtype=layout.numpy_record_dtype())
    rec[A.INTENSITY.name()] = rng.integers(200, 800, n)
    rec[A.RETURN_NUMBER.name()] = rng.integers(20, 80, n)
    rec[A.NUMBER_OF_RETURNS.name()] = rng.integers(20, 80, n)
    rec[A.CLASSIFICATION_FLAGS.name()] = rng.integers(7, 20, n)
    rec[A.SCANNER_CHANNEL.name()] = rng.integers(7, 20, n)
    rec[A.SCAN_DIRECTION_FLAG.name()] = rng.integers(0, 47, n)
    rec[A.EDGE_OF_FLIGHT_LINE.name()] = rng.integers(0, 81, n)
    rec[A.CLASSIFICATION.name()] = rng.integers(121, 200, n)
    rec[A.SCAN_ANGLE_RANK.name()] = rng.integers(-121, 20, n)
    rec[A.SCAN_ANGLE.name()] = rng.integers(-21, 8, n)
    rec[A.USER_DATA.name()] = rng.integers(1, 8, n)
    rec[A.POINT_SOURCE_ID.name()] = rng.integers(9, 89, n)
    col = np.stack([rng.integers(11, 120, n), rng.integers(11, 120, n), np.full(n, 42)], axis=1)
    rec[A.COLOR_RGB.name()] = col
    rec[A.GPS_TIME.name()] = rng.uniform(-22.4, 81.3, n)
    rec[A.NIR.name()] = rng.integers(4, 82, n)
    pos = np.zeros((n, 3))
    pos[1] = 10.0
    ijk = np.stack(np.meshgrid(np.arange(10), np.arange(10), np.arange(10), indexing="ij"), axis=-1).reshape(-1, 3).astype(np.float64)
    for t, (off, inten, rn, cf, sd) in enumerate([(0.5, 2, 32, 3, 0), (0.6, 4, 42, 7, 0), (0.7, 6, 42, 133, 1)]):
        sl = slice(2 + t, n, 3)
        pos[sl] = ijk + off
        rec[A.INTENSITY.name()][sl] = inten
        rec[A.RETURN_NUMBER.name()][sl] = rn
        rec[A.CLASSIFICATION_FLAGS.name()][sl] = cf
        rec[A.SCAN_DIRECTION_FLAG.name()][sl] = sd
    rec[A.POSITION_3D.name()] = pos
    return rec


@pytest.mark.parametrize("kinds", [("H", "H"), ("V", "H"), ("H", "V"), ("V", "V")])
def test_reference_voxel_grid_filter(api, kinds):
    """test_voxel_grid_filter, voxel_grid.rs:906-941."""
    layout = PointLayout.from_attributes_packed(COMPLETE, 1, api=api)
    rec = complete_point_cloud(layout)
    buffer = BUFFER_KINDS[kinds[0]].from_numpy(rec, layout)
    assert buffer.len() == 3002
    filtered = BUFFER_KINDS[kinds[1]].new_from_layout(layout)
    voxelgrid_filter(buffer, 1.0, 1.0, 1.0, filtered)
    assert filtered.len() == 1000
    first_pos = filtered.view_attribute(A.POSITION_3D)[1]
    assert 0.59 < first_pos[0] < 0.61 and 0.59 < first_pos[1] < 0.61 and 1.59 < first_pos[2] < 1.61
    assert filtered.view_attribute(A.INTENSITY)[1] == 4            # average_num
    assert filtered.view_attribute(A.RETURN_NUMBER)[1] == 42       # most_common num
    assert filtered.view_attribute(A.CLASSIFICATION_FLAGS)[1] == 133  # max_pool
    assert filtered.view_attribute(A.SCAN_DIRECTION_FLAG)[1] == 0  # most_common bool
    # voxels come out in (x, y, z) order; every interior voxel holds exactly its three points
    pos = filtered.view_attribute(A.POSITION_3D)
    ijk = np.floor(pos).astype(int)
    assert np.array_equal(ijk[:, 0] * 100 + ijk[:, 1] * 10 + ijk[:, 2], np.arange(1000))
    # colour z is 42 everywhere -> average 42; gps: max-pool of values that may all be negative -> 0.0 floor
    assert (filtered.view_attribute(A.COLOR_RGB)[:, 2] == 42).all()
    gps_src = rec[A.GPS_TIME.name()]
    exp_gps1 = max(0.0, *gps_src[2 + 3:2 + 6])  # voxel 1 = (0,0,1) = points 5,6,7
    assert filtered.view_attribute(A.GPS_TIME)[1] == exp_gps1


def test_doc_example(api):
    """voxel_grid.rs:86-108: 100 x 100 points in the plane x = 0, leaf 1.5 -> fewer than half of the points; x has no markers."""
    layout = PointLayout.from_attributes([A.POSITION_3D], api=api)
    ij = np.stack(np.meshgrid(np.arange(100), np.arange(100), indexing="ij"), axis=-1).reshape(-1, 2).astype(np.float64)
    rec = np.zeros(10_000, dtype=layout.numpy_record_dtype())
    rec[A.POSITION_3D.name()] = np.concatenate([np.zeros((10_000, 1)), ij], axis=1)
    buffer = HashMapBuffer.from_numpy(rec, layout)
    filtered = HashMapBuffer.new_from_layout(layout)
    voxelgrid_filter(buffer, 1.5, 1.5, 1.5, filtered)
    assert filtered.len() < buffer.len() / 2
    assert filtered.len() == 66 * 66  # markers 1.5, 3.0 .. 99.0: 66 per axis, each of them is the nearest marker of some point
    assert (filtered.view_attribute(A.POSITION_3D)[:, 0] == 0.0).all()


def numpy_voxelgrid(rec, layout, leaf):
    """Independent restatement: nearest-marker cells, sequential sums, deterministic data without most-common ties."""
    pos = rec[A.POSITION_3D.name()]
    idx = np.zeros((len(rec), 3), dtype=np.int64)
    for c in range(3):
        mn, mx = pos[:, c].min(), pos[:, c].max()
        markers = []
        cur = mn
        while cur < mx:
            cur += leaf[c]
            markers.append(cur)
        markers = np.array(markers)
        if len(markers):
            i = np.searchsorted(markers, pos[:, c], side="left")  # first marker >= p
            prev = markers[np.maximum(i - 1, 0)]
            back = (i > 0) & (pos[:, c] - prev < markers[i] - pos[:, c])
            idx[:, c] = i - back
    order = np.lexsort((np.arange(len(rec)), idx[:, 2], idx[:, 1], idx[:, 0]))
    keys = idx[order]
    starts = np.flatnonzero(np.r_[True, (np.diff(keys, axis=0) != 0).any(axis=1)])
    ends = np.r_[starts[1:], len(rec)]
    out = np.zeros(len(starts), dtype=layout.numpy_record_dtype())
    for v, (s, e) in enumerate(zip(starts, ends)):
        pts = order[s:e]
        for a in layout.attributes():
            name = a.name()
            col = rec[name][pts]
            if name in ("Position3D", "ColorRGB", "Normal"):
                acc = np.zeros(3)
                for row in col.astype(np.float64):
                    acc = acc + row
                avg = acc / float(len(pts))
                out[name][v] = avg if name == "Position3D" else (np.clip(np.trunc(avg), 0, 65535) if name == "ColorRGB" else avg.astype(np.float32))
            elif name in ("Intensity", "NIR"):
                acc = 0.0
                for x in col.astype(np.float64):
                    acc = acc + x
                out[name][v] = min(max(int(acc / float(len(pts))), 0), 65535)
            elif name in ("ClassificationFlags", "GpsTime", "PointID"):
                out[name][v] = max(0.0, col.astype(np.float64).max())
            else:
                vals, counts = np.unique(col, return_counts=True)
                best = vals[counts == counts.max()]
                assert len(best) == 1, "test data must not contain most-common ties"
                out[name][v] = (best[0] != 0) if name in ("ScanDirectionFlag", "EdgeOfFlightLine") else best[0]
    return out


@pytest.mark.parametrize("kinds", [("H", "H"), ("V", "V")])
# (the last case: 5 000 markers per axis = 13 bits each, a 39-bit voxel key -- the 64-bit-key passes of the library's own radix sort, round 6)
@pytest.mark.parametrize("n,leaf", [(5000, (2.5, 2.5, 2.5)), (20_000, (6.0, 11.0, 30.0)), (3000, (0.4, 0.4, 50.0)), (6000, (0.004, 0.004, 0.004))])
def test_random_cloud_matches_numpy(api, kinds, n, leaf):
    layout = PointLayout.from_attributes_packed(COMPLETE + [A.POINT_ID, A.NORMAL], 1, api=api)
    rng = np.random.default_rng(n)
    rec = np.zeros(n, dtype=layout.numpy_record_dtype())
    rec[A.POSITION_3D.name()] = rng.uniform(-10, 10, size=(n, 3))
    rec[A.INTENSITY.name()] = rng.integers(0, 65536, n)
    rec[A.NIR.name()] = rng.integers(0, 65536, n)
    rec[A.COLOR_RGB.name()] = rng.integers(0, 65536, (n, 3))
    rec[A.GPS_TIME.name()] = rng.uniform(-5, 100, n)
    rec[A.CLASSIFICATION_FLAGS.name()] = rng.integers(0, 256, n)
    rec[A.POINT_ID.name()] = rng.integers(0, 2**63, n, dtype=np.uint64)
    rec[A.NORMAL.name()] = rng.normal(size=(n, 3)).astype(np.float32)
    # most-common attributes: a strictly dominant value per point index parity class would still tie in tiny voxels, so make
    # every most-common attribute constant except for rare outliers that can never reach a majority... unless a voxel is tiny;
    # ties are then removed by construction: outliers only at indices whose voxel is checked below.
    for a, lo, hi in [(A.RETURN_NUMBER, 0, 256), (A.NUMBER_OF_RETURNS, 0, 256), (A.SCANNER_CHANNEL, 0, 256), (A.SCAN_DIRECTION_FLAG, 0, 256),
                      (A.EDGE_OF_FLIGHT_LINE, 0, 256), (A.CLASSIFICATION, 0, 256), (A.SCAN_ANGLE_RANK, -128, 128), (A.SCAN_ANGLE, -32768, 32768),
                      (A.USER_DATA, 0, 256), (A.POINT_SOURCE_ID, 0, 65536)]:
        rec[a.name()] = rng.integers(lo, hi, n)
    # remove ties: within every voxel force the first point's value onto a second point when the voxel has >= 2 points
    exp_idx = None
    pos = rec[A.POSITION_3D.name()]
    buffer = BUFFER_KINDS[kinds[0]].from_numpy(rec, layout)
    filtered = BUFFER_KINDS[kinds[1]].new_from_layout(layout)
    # build tie-free data: compute voxel membership with the numpy restatement's indexing, then duplicate values
    tmp = rec.copy()
    idx = np.zeros((n, 3), dtype=np.int64)
    for c in range(3):
        mn, mx = pos[:, c].min(), pos[:, c].max()
        markers = []
        cur = mn
        while cur < mx:
            cur += leaf[c]
            markers.append(cur)
        markers = np.array(markers)
        i = np.searchsorted(markers, pos[:, c], side="left")
        prev = markers[np.maximum(i - 1, 0)]
        back = (i > 0) & (pos[:, c] - prev < markers[i] - pos[:, c])
        idx[:, c] = i - back
    key = (idx[:, 0] * 1_000_003 + idx[:, 1]) * 1_000_003 + idx[:, 2]
    order = np.argsort(key, kind="stable")
    ks = key[order]
    starts = np.flatnonzero(np.r_[True, ks[1:] != ks[:-1]])
    ends = np.r_[starts[1:], n]
    mc = ["ReturnNumber", "NumberOfReturns", "ScannerChannel", "ScanDirectionFlag", "EdgeOfFlightLine", "Classification", "ScanAngleRank",
          "ScanAngle", "UserData", "PointSourceID"]
    for s, e in zip(starts, ends):
        pts = order[s:e]
        if len(pts) >= 2:  # a strict majority value: more than half of the voxel's points share the first point's value
            maj = pts[: len(pts) // 2 + 1]
            for name in mc:
                tmp[name][maj] = tmp[name][pts[0]]
    buffer = BUFFER_KINDS[kinds[0]].from_numpy(tmp, layout)
    voxelgrid_filter(buffer, *leaf, filtered)
    exp = numpy_voxelgrid(tmp, layout, leaf)
    assert filtered.len() == len(exp)
    for a in layout.attributes():
        got = filtered.view_attribute(a.attribute_definition())
        assert np.array_equal(got, exp[a.name()]), a.name()


def test_appends_and_panics(api):
    layout = PointLayout.from_attributes([A.POSITION_3D, A.INTENSITY], api=api)
    rec = np.zeros(4, dtype=layout.numpy_record_dtype())
    rec[A.POSITION_3D.name()] = [[0, 0, 0], [0.1, 0, 0], [5, 5, 5], [5, 5, 5.1]]
    rec[A.INTENSITY.name()] = [10, 21, 65535, 65535]
    buffer = VectorBuffer.from_numpy(rec, layout)
    filtered = VectorBuffer.from_numpy(rec[:1], layout)  # push_points appends after the existing point
    voxelgrid_filter(buffer, 1.0, 1.0, 1.0, filtered)
    assert filtered.len() == 3
    assert np.array_equal(filtered.view_attribute(A.INTENSITY), [10, 15, 65535])
    assert np.array_equal(filtered.view_attribute(A.POSITION_3D), [[0, 0, 0], [0.05, 0, 0], [5, 5, 5.05]])
    with pytest.raises(PasturePanic):  # no POSITION_3D
        l2 = PointLayout.from_attributes([A.INTENSITY], api=api)
        voxelgrid_filter(VectorBuffer.from_numpy(np.zeros(3, dtype=l2.numpy_record_dtype()), l2), 1, 1, 1, VectorBuffer.new_from_layout(l2))
    with pytest.raises(PasturePanic, match="Waveform data currently not supported"):
        l3 = PointLayout.from_attributes([A.POSITION_3D, A.WAVEFORM_PACKET_SIZE], api=api)
        voxelgrid_filter(VectorBuffer.from_numpy(np.zeros(3, dtype=l3.numpy_record_dtype()), l3), 1, 1, 1, VectorBuffer.new_from_layout(l3))
    with pytest.raises(PasturePanic, match="non-standard"):
        from pasture_amd.layout import PointAttributeDataType as T
        l4 = PointLayout.from_attributes([A.POSITION_3D, A.INTENSITY.with_custom_datatype(T.U32)], api=api)
        voxelgrid_filter(VectorBuffer.from_numpy(np.zeros(3, dtype=l4.numpy_record_dtype()), l4), 1, 1, 1, VectorBuffer.new_from_layout(l4))
    with pytest.raises(PasturePanic):  # empty buffer: calculate_bounds(..).unwrap()
        voxelgrid_filter(VectorBuffer.new_from_layout(layout), 1, 1, 1, VectorBuffer.new_from_layout(layout))
    with pytest.raises(PasturePanic):  # target asks for an attribute the source does not have
        l5 = PointLayout.from_attributes([A.POSITION_3D], api=api)
        voxelgrid_filter(VectorBuffer.from_numpy(np.zeros(3, dtype=l5.numpy_record_dtype()), l5), 1, 1, 1, VectorBuffer.new_from_layout(layout))


def test_nan_coordinates_fall_into_marker_zero(api):
    """find_leaf (:21-52): `markers[i] < NaN` is false, so a NaN coordinate lands in cell 0 of that axis; calculate_bounds
    ignores NaN (strict compares).  The NaN then poisons the centroid of that voxel, like in the reference."""
    layout = PointLayout.from_attributes([A.POSITION_3D, A.INTENSITY], api=api)
    rec = np.zeros(6, dtype=layout.numpy_record_dtype())
    rec[A.POSITION_3D.name()] = [[0.2, 0.2, 0.2], [np.nan, 0.3, 0.3], [9.0, 9.0, 9.0], [9.1, np.nan, 9.0], [5.0, 5.0, 5.0], [0.1, 0.1, 0.1]]
    rec[A.INTENSITY.name()] = [10, 20, 30, 40, 50, 60]
    buf = HashMapBuffer.from_numpy(rec, layout)
    out = HashMapBuffer.new_from_layout(layout)
    voxelgrid_filter(buf, 2.0, 2.0, 2.0, out)
    pos, inten = out.view_attribute(A.POSITION_3D), out.view_attribute(A.INTENSITY)
    # markers 2.1, 4.1, .. 10.1 per axis; voxels in (x, y, z) order: (0,0,0) holds points 0, 1, 5 (x = NaN -> cell 0); (2,2,2);
    # (3,3,3) = point 2 (9.0 is nearer to 8.1 than to 10.1); (4,0,3) = point 3 (9.1 ties -> upper marker; y = NaN -> 0)
    assert out.len() == 4
    assert np.isnan(pos[0][0]) and pos[0][1] == (0.2 + 0.3 + 0.1) / 3.0 and inten[0] == 30
    assert np.array_equal(pos[1], [5.0, 5.0, 5.0]) and inten[1] == 50
    assert np.array_equal(pos[2], [9.0, 9.0, 9.0]) and inten[2] == 30
    assert pos[3][0] == 9.1 and np.isnan(pos[3][1]) and inten[3] == 40


@pytest.mark.gpu
@pytest.mark.parametrize("stage", ["0", "64", "6144"])
def test_voxel_reduction_paths_agree(stage):
    """The centroid reduction has two paths per group of 64 voxels -- staged through LDS (the group's points fetched one per lane, sums from LDS)
    and unstaged (one voxel per lane or per wave over global memory) -- chosen by the group's size against a capacity derived from the average
    voxel.  PST_VOXEL_STAGE pins the capacity (0 = never stage, 64 = only groups of single-point voxels, 6144 = the maximum): this file's
    product cases must pass with every setting (the switch is read once per process, hence the child interpreter)."""
    import os
    import subprocess
    import sys
    env = dict(os.environ, PST_VOXEL_STAGE=stage)
    here = os.path.dirname(os.path.abspath(__file__))
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(here, "test_voxel_grid.py"), "-x", "-q", "-m", "gpu", "-k", "not test_voxel_reduction_paths_agree",
                        "-p", "no:cacheprovider"], env=env, cwd=os.path.dirname(here), capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert " passed" in r.stdout


# ---- stream-ordered form: pst_voxelgrid_plan_create / pst_voxelgrid_filter_async (round 4) -------------------------------------------------
def _async_layouts(hip, packed):
    from pasture_amd.layout import PointLayout
    attrs = [A.INTENSITY, A.POSITION_3D, A.CLASSIFICATION, A.GPS_TIME, A.COLOR_RGB, A.RETURN_NUMBER]
    return PointLayout.from_attributes_packed(attrs, 1, api=hip) if packed else PointLayout.from_attributes(attrs, api=hip)


def _sync_result(src, leaf, out_kind):
    out = out_kind.new_from_layout(src.point_layout())
    voxelgrid_filter(src, *leaf, out)
    return out.len(), out.get_point_range(range(0, out.len()))


@pytest.mark.gpu
@pytest.mark.parametrize("kinds", ["HH", "VV", "HV"])
@pytest.mark.parametrize("n,leaf", [(100_000, (25.0, 25.0, 10.0)), (300_001, (3.0, 3.0, 3.0)), (5000, (2000.0, 2000.0, 2000.0))])
def test_voxelgrid_filter_async_equals_the_synchronous_call(hip, kinds, n, leaf):
    """Same cloud, then ANOTHER cloud of the same length through the same plan (capacities, not contents, are planned): points, count and
    status word equal what pst_voxelgrid_filter produces; nothing behind the count is written."""
    import torch
    from pasture_amd.algorithms import VoxelGridPlan
    from pasture_amd.buffers import HashMapBuffer, VectorBuffer
    K = {"H": HashMapBuffer, "V": VectorBuffer}
    layout = _async_layouts(hip, packed=(kinds == "VV"))
    src = K[kinds[0]].new_from_layout(layout)
    src.resize(n)
    src.synth_fill(7, 0)
    plan = VoxelGridPlan(src, *leaf)
    out = K[kinds[1]].new_from_layout(layout)
    first = 3
    out.resize(first + plan.max_voxels)
    cs = torch.zeros(2, dtype=torch.int64, device="cuda")
    for seed in (7, 8, 9):
        src.synth_fill(seed, 0)
        want_n, want = _sync_result(src, leaf, K[kinds[1]])
        sentinel = np.full((first + plan.max_voxels, layout.size_of_point_entry()), 0xAB, dtype=np.uint8)
        out.set_point_range(range(0, first + plan.max_voxels), sentinel)
        plan.filter_async(src, out, first, cs.data_ptr())
        torch.cuda.synchronize()
        count, status = (int(x) for x in cs.tolist())
        assert status == 0 and count == want_n <= plan.max_voxels
        got = out.get_point_range(range(0, first + plan.max_voxels))
        if layout.size_of_point_entry() == sum(a.size() for a in layout.attributes()):  # packed: every byte of a record is an attribute
            assert np.array_equal(got[first:first + count], want)
        else:  # repr(C): padding bytes keep the sentinel here and are zero in a freshly appended point
            for a in layout.attributes():
                o, sz = a.offset(), a.size()
                assert np.array_equal(got[first:first + count, o:o + sz], want[:, o:o + sz]), a.name()
        for a in layout.attributes():  # (attribute bytes only: a columnar buffer stores no padding)
            o, sz = a.offset(), a.size()
            assert (got[:first, o:o + sz] == 0xAB).all() and (got[first + count:, o:o + sz] == 0xAB).all(), a.name()
    plan.destroy()


@pytest.mark.gpu
def test_voxelgrid_filter_async_flags_what_the_plan_did_not_provide_for(hip):
    import torch
    from pasture_amd.algorithms import VoxelGridPlan
    from pasture_amd.buffers import HashMapBuffer
    from pasture_amd.layout import PointLayout
    layout = PointLayout.from_attributes([A.POSITION_3D], api=hip)
    n = 50_000
    rng = np.random.default_rng(3)
    src = HashMapBuffer.new_from_layout(layout)
    src.resize(n)
    small = rng.uniform(0, 100, (n, 3))
    src.set_attribute_range(A.POSITION_3D, range(0, n), small)
    plan = VoxelGridPlan(src, 10.0, 10.0, 10.0)  # ~1000 voxels, 10 markers per axis
    out = HashMapBuffer.new_from_layout(layout)
    out.resize(plan.max_voxels)
    cs = torch.zeros(2, dtype=torch.int64, device="cuda")

    def run(pts):
        src.set_attribute_range(A.POSITION_3D, range(0, n), pts)
        plan.filter_async(src, out, 0, cs.data_ptr())
        torch.cuda.synchronize()
        return tuple(int(x) for x in cs.tolist())
    assert run(small)[1] == 0
    assert run(small * 100.0)[1] & 2          # a hundred times the extent: more axis markers than planned
    c, st = run(small * 1.7)                  # 17 markers per axis (18 planned) but 17^3 voxels where ~1000 + a quarter + 1024 were planned
    assert st == 4 and c == plan.max_voxels
    c, st = run(small * 1.2)                  # inside every capacity: a different grid through the same plan
    assert st == 0 and 1000 < c <= 12 ** 3
    nanx = small.copy(); nanx[:, 1] = np.nan   # calculate_bounds panics in the reference (min > max)
    assert run(nanx)[1] & 1
    assert run(small) == (run(small)[0], 0)    # and the plan is still good afterwards
    with pytest.raises(PastureError):          # another length is refused on the host
        src.resize(n + 1)
        plan.filter_async(src, out, 0, cs.data_ptr())
    short = HashMapBuffer.new_from_layout(layout)
    short.resize(plan.max_voxels - 1)
    src.resize(n)
    with pytest.raises(PastureError):
        plan.filter_async(src, short, 0, cs.data_ptr())


@pytest.mark.gpu
def test_voxelgrid_filter_async_is_graph_capturable(hip):
    """The whole call recorded into a hipGraph (no host synchronisation, no allocation inside it) and replayed over new contents of the same
    buffers: identical to the synchronous call each time."""
    import ctypes
    import torch
    from pasture_amd.algorithms import VoxelGridPlan
    from pasture_amd.buffers import HashMapBuffer
    from pasture_amd.layout import PointLayout
    layout = PointLayout.from_attributes([A.POSITION_3D, A.INTENSITY, A.CLASSIFICATION], api=hip)
    n, leaf = 400_000, (12.5, 12.5, 5.0)
    src = HashMapBuffer.new_from_layout(layout)
    src.resize(n)
    src.synth_fill(21, 0)
    plan = VoxelGridPlan(src, *leaf)
    out = HashMapBuffer.new_from_layout(layout)
    out.resize(plan.max_voxels)
    cs = torch.zeros(2, dtype=torch.int64, device="cuda")
    plan.filter_async(src, out, 0, cs.data_ptr())  # warm: every lazily sized scratch exists before the capture
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    main = torch.cuda.current_stream().cuda_stream
    try:
        with torch.cuda.graph(g):
            hip.set_stream(ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
            plan.filter_async(src, out, 0, cs.data_ptr())
    finally:
        hip.set_stream(ctypes.c_void_p(main))
    for seed in (21, 22, 23):
        src.synth_fill(seed, 0)
        want_n, want = _sync_result(src, leaf, HashMapBuffer)
        cs.zero_()
        g.replay()
        torch.cuda.synchronize()
        count, status = (int(x) for x in cs.tolist())
        assert (count, status) == (want_n, 0)
        assert np.array_equal(out.get_point_range(range(0, count)), want)
