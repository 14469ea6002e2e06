"""Oracle-only pins: helper-level known answers of the reference's tests that have no counterpart in the C ABI."""
import ctypes as C

import numpy as np


def test_align_to(oracle):  # pasture-core/src/math/arithmetic.rs:78-85
    f = oracle.lib.orc_align_to
    f.restype = C.c_uint64
    f.argtypes = [C.c_uint64, C.c_uint64]
    assert [f(1, 0), f(1, 2), f(0, 2), f(4, 8), f(27, 8), f(8, 8), f(5, 8)] == [1, 2, 0, 8, 32, 8, 8]


def test_centroid_covariance_plane_known_answer(oracle):  # normal_estimation.rs:503-550
    pts = np.array([[1, 0, 0], [0, 1, 0], [1, 1, 0], [-1, 0, 0]], dtype=np.float64)
    cen, cov, ok = (C.c_double * 3)(), (C.c_double * 9)(), C.c_int()
    oracle.lib.orc_covariance.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.POINTER(C.c_int)]
    assert oracle.lib.orc_covariance(pts.ctypes.data, 4, cen, cov, C.byref(ok)) == 0 and ok.value == 1
    assert list(cen) == [0.25, 0.5, 0.0]
    assert list(cov) == [0.6875 * 4.0, 0.125 * 4.0, 0.0, 0.125 * 4.0, 0.25 * 4.0, 0.0, 0.0, 0.0, 0.0]
    normal, curv = (C.c_double * 3)(), C.c_double()
    oracle.lib.orc_plane_parameter.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(C.c_double)]
    assert oracle.lib.orc_plane_parameter(cov, normal, C.byref(curv)) == 0
    assert normal[0] == 0.0 and normal[1] == 0.0 and normal[2] != 0.0 and curv.value == 0.0


def test_covariance_error_on_nan(oracle):  # :552-578
    nan = float("nan")
    pts = np.array([[nan, 0, 0], [0, 1, nan], [1, 1, nan], [-1, nan, 0]], dtype=np.float64)
    cen, cov, ok = (C.c_double * 3)(), (C.c_double * 9)(), C.c_int(1)
    oracle.lib.orc_covariance.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.POINTER(C.c_int)]
    assert oracle.lib.orc_covariance(pts.ctypes.data, 4, cen, cov, C.byref(ok)) == 0 and ok.value == 0


def test_kdtree_matches_brute_force(oracle):
    """The stand-in for the un-vendored kd-tree crate returns exactly the k nearest by squared Euclidean distance."""
    from pasture_amd.algorithms import compute_normals
    from pasture_amd.buffers import VectorBuffer
    from pasture_amd.layout import PointLayout, attributes as A
    rng = np.random.default_rng(5)
    pts = rng.random((600, 3))
    buf = VectorBuffer.new_from_layout(PointLayout.from_attributes([A.POSITION_3D], api=oracle))
    buf.resize(600)
    buf.set_attribute_range(A.POSITION_3D, range(0, 600), pts)
    _, _, knn = compute_normals(buf, 9, return_knn=True)
    d = ((pts[:, None, :] - pts[None, :, :]) ** 2).sum(-1)
    want = np.argsort(d, axis=1, kind="stable")[:, :9]
    assert np.array_equal(np.sort(knn, axis=1), np.sort(want, axis=1))
    assert np.array_equal(knn[:, 0], np.arange(600))  # the query point itself comes first (distance 0)
