"""SliceBuffer::slice (pasture-core/src/containers/slice.rs:16-43), compute_centroid (pasture-algorithms/src/normal_estimation.rs:198-237,
known answer :503-528) and view_attribute_with_conversion (point_buffer.rs:322-330, buffer_views.rs:533-650).
Expectations are numpy / the reference's literals; the CPU suite pins the oracle, the GPU suite the HIP path (and the two against each other)."""
import numpy as np
import pytest

from harness import BUFFER_KINDS
from pasture_amd._capi import PastureError, PasturePanic
from pasture_amd.algorithms import calculate_bounds, compute_centroid, compute_normals, minmax_attribute, transform_attribute
from pasture_amd.conversion import BufferLayoutConverter, Transform
from pasture_amd.layout import PointAttributeDataType as T, PointAttributeDefinition, PointLayout, attributes as A
from rust_as_ref import rust_as_array

KINDS = ["V", "H"]


def cloud(api, kind, n, seed=3, packed=False):
    """Position3D (Vec3f64) + Intensity (u16) + GpsTime (f64) + Classification (u8), random values."""
    attrs = [A.POSITION_3D, A.INTENSITY, A.GPS_TIME, A.CLASSIFICATION]
    layout = PointLayout.from_attributes_packed(attrs, 1, api=api) if packed else PointLayout.from_attributes(attrs, api=api)
    buf = BUFFER_KINDS[kind].new_from_layout(layout)
    buf.resize(n)
    rng = np.random.default_rng(seed)
    cols = {
        "Position3D": rng.random((n, 3)) * np.array([1000.0, 1000.0, 100.0]) - 300.0,
        "Intensity": rng.integers(0, 65536, n).astype(np.uint16),
        "GpsTime": rng.random(n) * 1e6 - 5e5,
        "Classification": rng.integers(0, 256, n).astype(np.uint8),
    }
    if n:
        for a in attrs:
            buf.set_attribute_range(a, range(0, n), cols[a.name()])
    return buf, cols


# ---- slices ----------------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("kind", KINDS)
@pytest.mark.parametrize("packed", [False, True])
def test_slice_is_a_view_with_the_parents_layout(api, kind, packed):
    buf, cols = cloud(api, kind, 1000, packed=packed)
    s = buf.slice(range(100, 350))
    assert s.len() == 250 and s.point_layout() == buf.point_layout()
    assert (s.as_columnar() is None) == (buf.as_columnar() is None)  # same memory-layout capabilities (slice.rs:76-79)
    assert np.array_equal(s.get_point_range(range(0, 250)), buf.get_point_range(range(100, 350)))
    assert np.array_equal(s.view_attribute(A.INTENSITY), cols["Intensity"][100:350])
    assert np.array_equal(s.get_attribute_range(A.POSITION_3D, range(10, 20)), cols["Position3D"][110:120])
    # slice of a slice
    s2 = s.slice(range(50, 60))
    assert np.array_equal(s2.view_attribute(A.GPS_TIME), cols["GpsTime"][150:160])
    # empty slices anywhere inside, including at the end
    assert buf.slice(range(1000, 1000)).len() == 0 and buf.slice(range(7, 7)).len() == 0


@pytest.mark.parametrize("kind", KINDS)
def test_slice_panics(api, kind):
    buf, _ = cloud(api, kind, 100)
    with pytest.raises(PasturePanic) as e:
        buf.slice(range(50, 101))
    assert e.value.code == 3
    s = buf.slice(range(10, 20))
    with pytest.raises(PasturePanic) as e:  # local index assertions, slice.rs:52-75
        s.get_attribute_range(A.INTENSITY, range(5, 11))
    assert e.value.code == 3
    with pytest.raises(PastureError) as e:  # not an OwningBuffer
        s.resize(20)
    assert e.value.code == 23


@pytest.mark.parametrize("kind", KINDS)
@pytest.mark.parametrize("packed", [False, True])
def test_swap_exchanges_two_points_and_panics_out_of_bounds(api, kind, packed):
    """BorrowedMutBuffer::swap, point_buffer.rs:229 (VectorBuffer :770-783, HashMapBuffer :1276-1292): expectations are numpy swaps."""
    n = 257
    buf, cols = cloud(api, kind, n, packed=packed)
    expect = {k: v.copy() for k, v in cols.items()}
    for i, j in [(0, n - 1), (5, 5), (17, 18), (n - 1, 100), (100, 0)]:
        buf.swap(i, j)
        for v in expect.values():
            v[[i, j]] = v[[j, i]]
    for a in (A.POSITION_3D, A.INTENSITY, A.GPS_TIME, A.CLASSIFICATION):
        assert np.array_equal(buf.view_attribute(a), expect[a.name()])
    view = buf.slice(range(10, 20))  # swap through a slice_mut exchanges points of the parent
    view.swap(0, 9)
    for v in expect.values():
        v[[10, 19]] = v[[19, 10]]
    assert np.array_equal(buf.view_attribute(A.POSITION_3D), expect["Position3D"])
    for i, j in [(n, 0), (0, n), (n + 5, n + 5)]:
        with pytest.raises(PasturePanic) as e:
            buf.swap(i, j)
        assert e.value.code == 3
    with pytest.raises(PasturePanic):
        view.swap(0, 10)


@pytest.mark.gpu
@pytest.mark.parametrize("kind", KINDS)
def test_stale_slice_is_an_error_not_a_read_of_freed_memory(hip, kind):
    """What Rust's borrow checker forbids (slice.rs:16-43: the slice borrows the parent) the C ABI has to catch at run time: a slice cut before
    its parent was resized or destroyed fails with PST_ERR_INVALID_ARGUMENT on every later use; a slice cut afterwards works."""
    buf, cols = cloud(hip, kind, 5000)
    s = buf.slice(range(1000, 3000))
    ss = s.slice(range(10, 20))
    assert np.array_equal(s.view_attribute(A.INTENSITY), cols["Intensity"][1000:3000])
    buf.resize(5000)  # same length: nothing moves, the slices stay valid
    assert np.array_equal(ss.view_attribute(A.INTENSITY), cols["Intensity"][1010:1020])
    buf.resize(200_000)  # the storage is reallocated
    for stale in (s, ss):
        for use in (lambda v: v.view_attribute(A.INTENSITY), lambda v: calculate_bounds(v), lambda v: v.slice(range(0, 5)), lambda v: len(v)):
            with pytest.raises(PastureError) as e:
                use(stale)
            assert e.value.code == 1 and "slice" in str(e.value)
    fresh = buf.slice(range(1000, 3000))
    assert np.array_equal(fresh.view_attribute(A.INTENSITY), cols["Intensity"][1000:3000])
    buf.resize(100)  # shrinking needs &mut too
    with pytest.raises(PastureError) as e:
        fresh.view_attribute(A.INTENSITY)
    assert e.value.code == 1
    # destroying the parent while a view is alive (the Python mirror keeps the parent alive for its views; the C ABI does not)
    raw_parent, _ = cloud(hip, kind, 300)
    view = raw_parent.slice(range(0, 100))
    view._keepalive = None
    hip.buffer_destroy(raw_parent._h)
    raw_parent._h = None
    with pytest.raises(PastureError) as e:
        calculate_bounds(view)
    assert e.value.code == 1


@pytest.mark.parametrize("kind", KINDS)
def test_bounds_of_slices_and_chunked_minmax(api, kind):
    """calculate_bounds(&buf.slice(a..b)) and the chunked min-max of pasture-tools/src/bin/info.rs:66-78 (per-chunk minmax_attribute folded
    with infimum / supremum) equal the whole-buffer answers."""
    n = 50_003
    buf, cols = cloud(api, kind, n)
    pos = cols["Position3D"]
    for a, b in [(0, n), (1, 2), (12345, 23456), (n - 1, n), (4097, 4097 + 1536)]:
        got = calculate_bounds(buf.slice(range(a, b)))
        assert got.min() == tuple(pos[a:b].min(axis=0)) and got.max() == tuple(pos[a:b].max(axis=0))
    assert calculate_bounds(buf.slice(range(5, 5))) is None
    chunk = 7001
    for attr, key in [(A.INTENSITY, "Intensity"), (A.GPS_TIME, "GpsTime"), (A.CLASSIFICATION, "Classification")]:
        acc = None
        for first in range(0, n, chunk):
            mm = minmax_attribute(buf.slice(range(first, min(n, first + chunk))), attr)
            acc = mm if acc is None else (min(acc[0], mm[0]), max(acc[1], mm[1]))
        whole = minmax_attribute(buf, attr)
        assert acc == whole == (cols[key].min(), cols[key].max())
    acc = None
    for first in range(0, n, chunk):
        mn, mx = minmax_attribute(buf.slice(range(first, min(n, first + chunk))), A.POSITION_3D)
        acc = (mn, mx) if acc is None else (np.minimum(acc[0], mn), np.maximum(acc[1], mx))
    assert np.array_equal(acc[0], pos.min(axis=0)) and np.array_equal(acc[1], pos.max(axis=0))


@pytest.mark.parametrize("kind", KINDS)
def test_transform_attribute_on_slice_mut_leaves_the_rest_untouched(api, kind):
    n = 10_000
    buf, cols = cloud(api, kind, n, packed=True)
    before = buf.get_point_range(range(0, n)).copy()
    scale, offset = (0.5, 2.0, -1.0), (10.0, -20.0, 30.0)
    a, b = 1234, 7777
    transform_attribute(buf.slice_mut(range(a, b)), A.POSITION_3D, Transform.affine(T.Vec3f64, scale, offset))
    after = buf.get_point_range(range(0, n))
    assert np.array_equal(after[:a], before[:a]) and np.array_equal(after[b:], before[b:])
    want = (cols["Position3D"][a:b] * np.array(scale)) + np.array(offset)  # two roundings
    assert np.array_equal(buf.get_attribute_range(A.POSITION_3D, range(a, b)), want)
    for attr, key in [(A.INTENSITY, "Intensity"), (A.GPS_TIME, "GpsTime"), (A.CLASSIFICATION, "Classification")]:
        assert np.array_equal(buf.view_attribute(attr), cols[key])


@pytest.mark.parametrize("src_kind", KINDS)
@pytest.mark.parametrize("dst_kind", KINDS)
def test_convert_from_and_into_slices(api, src_kind, dst_kind):
    n = 5000
    src, cols = cloud(api, src_kind, n)
    pos32 = A.POSITION_3D.with_custom_datatype(T.Vec3f32)
    tgt_layout = PointLayout.from_attributes([A.INTENSITY, pos32], api=api)
    dst = BUFFER_KINDS[dst_kind].new_from_layout(tgt_layout)
    dst.resize(n)
    conv = BufferLayoutConverter.for_layouts(src.point_layout(), tgt_layout)
    a, b = 777, 3210
    conv.convert_into(src.slice(range(a, b)), dst.slice_mut(range(a + 5, b + 5)))
    got_i = dst.view_attribute(A.INTENSITY)
    got_p = dst.view_attribute(pos32)
    assert np.array_equal(got_i[a + 5:b + 5], cols["Intensity"][a:b]) and not got_i[:a + 5].any() and not got_i[b + 5:].any()
    assert np.array_equal(got_p[a + 5:b + 5], cols["Position3D"][a:b].astype(np.float32)) and not got_p[:a + 5].any()


@pytest.mark.parametrize("kind", KINDS)
def test_compute_normals_of_a_slice_equals_normals_of_a_copy(api, kind):
    n = 3000
    buf, cols = cloud(api, kind, n, seed=11)
    a, b = 500, 2100
    copy = BUFFER_KINDS[kind].new_from_layout(buf.point_layout())
    copy.resize(b - a)
    copy.set_point_range(range(0, b - a), buf.get_point_range(range(a, b)))
    n1, c1, k1 = compute_normals(buf.slice(range(a, b)), 8, return_knn=True)
    n2, c2, k2 = compute_normals(copy, 8, return_knn=True)
    assert np.array_equal(k1, k2) and np.array_equal(n1, n2) and np.array_equal(c1, c2)


# ---- compute_centroid ------------------------------------------------------------------------------------------------------------------
def _pos_buffer(api, kind, pts):
    layout = PointLayout.from_attributes([A.POSITION_3D, A.INTENSITY], api=api)
    buf = BUFFER_KINDS[kind].new_from_layout(layout)
    pts = np.asarray(pts, dtype=np.float64).reshape(-1, 3)
    buf.resize(pts.shape[0])
    if pts.shape[0]:
        buf.set_attribute_range(A.POSITION_3D, range(0, pts.shape[0]), pts)
    return buf


@pytest.mark.parametrize("kind", KINDS)
def test_centroid_known_answer(api, kind):  # normal_estimation.rs:503-528: test_compute_normal_sub
    c = compute_centroid(_pos_buffer(api, kind, [[1, 0, 0], [0, 1, 0], [1, 1, 0], [-1, 0, 0]]))
    assert c == (0.25, 0.5, 0.0)


@pytest.mark.parametrize("kind", KINDS)
def test_centroid_nan_and_inf_semantics(api, kind):
    nan, inf = float("nan"), float("inf")
    # a NaN coordinate anywhere => the not-dense branch: mean over the FINITE points only (:218-233)
    c = compute_centroid(_pos_buffer(api, kind, [[1, 2, 3], [nan, 0, 0], [3, 4, 5], [inf, 1, 1], [5, 6, 10]]))
    assert c == (3.0, 4.0, 6.0)
    # inf without any NaN => still "dense" (is_dense tests is_nan only, :133-140): the plain mean, inf included
    c = compute_centroid(_pos_buffer(api, kind, [[1, 2, 3], [inf, 0, 0], [3, 4, 6]]))
    assert c == (inf, 2.0, 3.0)
    # no finite point at all: 0 / 0
    c = compute_centroid(_pos_buffer(api, kind, [[nan, 0, 0], [0, nan, 0]]))
    assert all(np.isnan(c))


@pytest.mark.parametrize("kind", KINDS)
def test_centroid_panics(api, kind):
    with pytest.raises(PasturePanic) as e:
        compute_centroid(_pos_buffer(api, kind, np.zeros((0, 3))))
    assert e.value.code == 11 and "The point cloud is empty!" in e.value.message
    layout = PointLayout.from_attributes([A.POSITION_3D.with_custom_datatype(T.Vec3f32)], api=api)
    buf = BUFFER_KINDS[kind].new_from_layout(layout)
    buf.resize(3)
    with pytest.raises(PasturePanic) as e:  # view_attribute::<Vector3<f64>> needs the exact datatype
        compute_centroid(buf)
    assert e.value.code == 4


@pytest.mark.parametrize("kind", KINDS)
@pytest.mark.parametrize("n", [1, 63, 64, 65, 257, 4096, 100_003, 1_000_000])
def test_centroid_random_within_1e9_of_the_sequential_sum(api, kind, n):
    """The reference adds left to right; a device sum is a tree.  Both are within n eps of the exact mean: 1e-9 relative to the
    coordinates' magnitude (the north star's f64 tolerance) with a wide margin; the oracle (sequential, like the reference) is exact
    against np.cumsum's order."""
    rng = np.random.default_rng(n)
    pts = rng.random((n, 3)) * np.array([1000.0, 1000.0, 100.0]) + np.array([500000.0, 5400000.0, 100.0])
    c = np.array(compute_centroid(_pos_buffer(api, kind, pts)))
    seq = np.array([np.cumsum(pts[:, k])[-1] / n for k in range(3)])  # cumsum = sequential left-to-right f64 additions
    if not api.is_product:
        assert np.array_equal(c, seq)
    assert np.all(np.abs(c - seq) <= 1e-9 * np.abs(seq))
    sl = _pos_buffer(api, kind, pts).slice(range(n // 3, n))
    cs = np.array(compute_centroid(sl))
    want = pts[n // 3:].mean(axis=0)
    assert np.all(np.abs(cs - want) <= 1e-9 * np.abs(want))


# ---- view_attribute_with_conversion ------------------------------------------------------------------------------------------------------
SCALARS = [T.U8, T.I8, T.U16, T.I16, T.U32, T.I32, T.U64, T.I64, T.F32, T.F64]


@pytest.mark.parametrize("kind", KINDS)
def test_converting_view_scalars_and_vec3(api, kind):
    n = 4099
    buf, cols = cloud(api, kind, n, packed=True)
    for dt in SCALARS:
        for name, key in [("GpsTime", "GpsTime"), ("Intensity", "Intensity"), ("Classification", "Classification")]:
            got = buf.view_attribute_with_conversion(PointAttributeDefinition.custom(name, dt))
            want = rust_as_array(cols[key], dt.numpy_dtype())
            assert got.dtype == dt.numpy_dtype() and got.tobytes() == want.tobytes(), (name, dt)
    for dt in [T.Vec3f32, T.Vec3i32, T.Vec3u16, T.Vec3u8, T.Vec3f64]:
        got = buf.view_attribute_with_conversion(A.POSITION_3D.with_custom_datatype(dt))
        want = rust_as_array(cols["Position3D"], dt.numpy_dtype())
        assert got.tobytes() == want.tobytes(), dt
    # a range, and through a slice
    got = buf.view_attribute_with_conversion(PointAttributeDefinition.custom("Intensity", T.F32), range(100, 1000))
    assert np.array_equal(got, cols["Intensity"][100:1000].astype(np.float32))
    got = buf.slice(range(100, 1000)).view_attribute_with_conversion(PointAttributeDefinition.custom("Intensity", T.F32))
    assert np.array_equal(got, cols["Intensity"][100:1000].astype(np.float32))


@pytest.mark.parametrize("kind", KINDS)
def test_converting_view_errors(api, kind):
    buf, _ = cloud(api, kind, 10)
    with pytest.raises(PasturePanic) as e:  # .expect("Attribute not found in PointLayout of buffer"), buffer_views.rs:549-552
        buf.view_attribute_with_conversion(PointAttributeDefinition.custom("Nope", T.F32))
    assert e.value.code == 4
    with pytest.raises(PasturePanic) as e:  # scalar <-> Vec3: no entry in the `as` table => Err(...) :553-561
        buf.view_attribute_with_conversion(PointAttributeDefinition.custom("Intensity", T.Vec3f32))
    assert e.value.code == 5
    with pytest.raises(PasturePanic) as e:
        buf.view_attribute_with_conversion(A.POSITION_3D.with_custom_datatype(T.F64))
    assert e.value.code == 5
    with pytest.raises(PasturePanic) as e:
        buf.view_attribute_with_conversion(PointAttributeDefinition.custom("Intensity", T.F32), range(5, 11))
    assert e.value.code == 3
