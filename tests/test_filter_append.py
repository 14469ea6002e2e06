"""OwningBufferExt::append (point_buffer.rs:419-489; reference test :2045-2075) and HashMapBuffer::filter / filter_into
(:1064-1136; reference test :2296-2330, bench benches/buffer_filter_bench.rs).  Expectations are numpy index expressions, so
the CPU suite pins the oracle and the GPU suite the HIP path against the same answers."""
import numpy as np
import pytest

from harness import BUFFER_KINDS, custom_point_type_big, random_records
from pasture_amd._capi import PasturePanic
from pasture_amd.buffers import HashMapBuffer, VectorBuffer
from pasture_amd.layout import PointAttributeDataType as T, PointAttributeDefinition, PointLayout, attributes as A


def records_of(buf) -> np.ndarray:
    """Attribute-wise content (padding excluded), comparable across storage kinds."""
    return {a.name(): buf.view_attribute(a.attribute_definition()) for a in buf.point_layout().attributes()}


def assert_same(buf, rec):
    got = records_of(buf)
    assert buf.len() == len(rec)
    for k, v in got.items():
        assert np.array_equal(v, rec[k]), k


@pytest.mark.parametrize("self_kind", ["V", "H"])
@pytest.mark.parametrize("other_kind", ["V", "H"])
def test_append_reference_case(api, self_kind, other_kind):
    """point_buffer.rs:2045-2075: append 16 CustomPointTypeBig points to an empty buffer, all four storage pairings."""
    layout = custom_point_type_big(api)
    rec = random_records(layout, 16, 1)
    other = BUFFER_KINDS[other_kind].from_numpy(rec, layout)
    buf = BUFFER_KINDS[self_kind].new_from_layout(layout)
    buf.append(other)
    assert_same(buf, rec)


@pytest.mark.parametrize("self_kind", ["V", "H"])
@pytest.mark.parametrize("other_kind", ["V", "H"])
def test_append_repeatedly_grows_and_keeps_old_points(api, self_kind, other_kind):
    layout = PointLayout.from_attributes([A.POSITION_3D, A.INTENSITY, A.CLASSIFICATION], api=api)  # repr(C) rules: padded records
    parts = [random_records(layout, n, 10 + n) for n in (5, 0, 1, 1000, 37)]
    buf = BUFFER_KINDS[self_kind].new_from_layout(layout)
    for p in parts:
        buf.append(BUFFER_KINDS[other_kind].from_numpy(p, layout))
    assert_same(buf, np.concatenate(parts))


def test_append_layout_mismatch_panics(api):
    a = VectorBuffer.new_from_layout(PointLayout.from_attributes([A.POSITION_3D], api=api))
    b = VectorBuffer.new_from_layout(PointLayout.from_attributes([A.POSITION_3D, A.INTENSITY], api=api))
    with pytest.raises(PasturePanic):
        a.append(b)


@pytest.mark.parametrize("out_kind", ["V", "H"])
def test_filter_reference_case(api, out_kind):
    """point_buffer.rs:2296-2330: even indices of 16 CustomPointTypeBig points into both buffer kinds."""
    layout = custom_point_type_big(api)
    rec = random_records(layout, 16, 2)
    src = HashMapBuffer.from_numpy(rec, layout)
    out = src.filter(BUFFER_KINDS[out_kind], lambda idx: idx % 2 == 0)
    assert_same(out, rec[::2])


@pytest.mark.parametrize("out_kind", ["V", "H"])
@pytest.mark.parametrize("n,density", [(4096, 0.5), (1, 1.0), (1, 0.0), (2047, 0.5), (2049, 0.03), (10_000, 0.97), (70_001, 0.5), (5000, 0.0), (5000, 1.0)])
def test_filter_random_masks(api, out_kind, n, density):
    """buffer_filter_bench.rs: random bool mask (4096 points there) — plus ragged sizes and extreme densities."""
    layout = custom_point_type_big(api)
    rec = random_records(layout, n, n)
    mask = np.random.default_rng(n + 1).random(n) < density
    src = HashMapBuffer.from_numpy(rec, layout)
    out = src.filter(BUFFER_KINDS[out_kind], mask)
    assert_same(out, rec[mask])


@pytest.mark.parametrize("out_kind", ["V", "H"])
def test_filter_every_datatype_and_padding(api, out_kind):
    """All attribute sizes (1..24 bytes, odd ByteArray) in a repr(C) layout with padding; interleaved target keeps its
    padding bytes (the reference writes attribute bytes only)."""
    attrs = [PointAttributeDefinition(f"a{k}", T(k)) for k in range(16)] + [PointAttributeDefinition("blob", T.ByteArray(5))]
    layout = PointLayout.from_attributes(attrs, api=api)
    n = 3001
    rec = random_records(layout, n, 9)
    mask = np.random.default_rng(3).random(n) < 0.4
    src = HashMapBuffer.from_numpy(rec, layout)
    out = src.filter(BUFFER_KINDS[out_kind], mask)
    assert_same(out, rec[mask])
    if out_kind == "V":
        dst = VectorBuffer.new_from_layout(layout)
        k = int(mask.sum())
        dst.resize(k + 3)
        raw = np.full((k + 3, layout.size_of_point_entry()), 0xAB, np.uint8)
        dst.set_point_range(range(0, k + 3), raw.view(layout.numpy_record_dtype()).reshape(-1))
        assert src.filter_into(dst, mask) == k
        got = np.ascontiguousarray(dst.get_point_range(range(0, k + 3))).view(np.uint8).reshape(k + 3, -1)
        covered = np.zeros(layout.size_of_point_entry(), bool)
        for a in layout.attributes():
            covered[a.offset():a.offset() + a.size()] = True
        assert (got[:, ~covered] == 0xAB).all() and (got[k:] == 0xAB).all()
        exp = np.ascontiguousarray(rec[mask]).view(np.uint8).reshape(k, -1)
        assert np.array_equal(got[:k][:, covered], exp[:, covered])


def test_filter_into_hint_and_panics(api):
    layout = custom_point_type_big(api)
    n = 1000
    rec = random_records(layout, n, 4)
    mask = np.arange(n) % 3 == 0
    k = int(mask.sum())
    src = HashMapBuffer.from_numpy(rec, layout)
    dst = HashMapBuffer.new_from_layout(layout)
    dst.resize(k)
    assert src.filter_into(dst, mask, k) == k
    assert_same(dst, rec[mask])
    short = HashMapBuffer.new_from_layout(layout)
    short.resize(k - 1)
    with pytest.raises(PasturePanic, match="at least as large as the number of predicate matches"):
        src.filter_into(short, mask)
    with pytest.raises(PasturePanic):  # hint smaller than the real number of matches: slice index panic in the reference
        src.filter_into(dst, mask, k - 5)
    from pasture_amd._capi import PastureError
    with pytest.raises(PastureError, match="filter is defined on HashMapBuffer"):  # the reference has no VectorBuffer::filter
        VectorBuffer.from_numpy(rec, layout).filter(HashMapBuffer, mask)
    other = HashMapBuffer.new_from_layout(PointLayout.from_attributes([A.POSITION_3D], api=api))
    other.resize(n)
    with pytest.raises(PasturePanic, match="PointLayouts must match"):
        src.filter_into(other, mask)
    # hint larger than the matches: only the real matches are written
    big = HashMapBuffer.new_from_layout(layout)
    big.resize(n)
    assert src.filter_into(big, mask, n) == k
    for a in layout.attributes():
        assert np.array_equal(big.view_attribute(a.attribute_definition())[:k], rec[a.name()][mask])


@pytest.mark.gpu
def test_filter_into_async_matches_the_synchronous_call(hip):
    """The stream-ordered filter_into (`Some(num_matches)`, device mask; point_buffer.rs:1082-1136): same target as the synchronous call, the
    hit count delivered in stream order; a hint below the hit count writes only `hint` points and reports the real count (the reference's
    slice panic :1103-1108 becomes the caller's check); a target shorter than the hint is refused on the host."""
    import torch
    api = hip
    layout = custom_point_type_big(api)
    n = 300_007
    rec = random_records(layout, n, 9)
    src = HashMapBuffer.from_numpy(rec, layout)
    mask_h = np.random.default_rng(5).random(n) < 0.37
    mask = torch.from_numpy(mask_h.astype(np.uint8)).cuda()
    k = int(mask_h.sum())
    hits = torch.full((1,), -1, dtype=torch.int64, device="cuda")
    for kind in (HashMapBuffer, VectorBuffer):
        got = kind.new_from_layout(layout)
        got.resize(k)
        src.filter_into_async(got, mask.data_ptr(), k, hits.data_ptr())
        src.filter_into_async(got, mask.data_ptr(), k, hits.data_ptr())  # back to back on one stream: the scratch is reused in stream order
        assert int(hits.item()) == k
        assert_same(got, rec[mask_h])
        # fewer points announced than the mask selects: only `hint` points are written, the count says what the mask holds
        short = kind.new_from_layout(layout)
        short.resize(k)
        hint = k - 1000
        src.filter_into_async(short, mask.data_ptr(), hint, hits.data_ptr())
        assert int(hits.item()) == k
        for a in layout.attributes():
            col = short.get_attribute_range(a.attribute_definition(), range(0, k))
            assert np.array_equal(col[:hint], rec[a.name()][mask_h][:hint]), a.name()
            assert not np.ascontiguousarray(col[hint:]).view(np.uint8).any(), a.name()
        tiny = kind.new_from_layout(layout)
        tiny.resize(k - 1)
        with pytest.raises(PasturePanic, match="at least as large as the number of predicate matches"):
            src.filter_into_async(tiny, mask.data_ptr(), k)


@pytest.mark.gpu
@pytest.mark.parametrize("density", [0.5, 1.0, 0.02, 0.0])
def test_filter_big_layout_takes_the_streaming_kernels(hip, density):
    """CustomPointTypeBig columns (buffer_filter_bench.rs:71-74) into both target kinds run on the streaming compaction kernels (plan family
    "static"): several tiles, a ragged last one, tiles with more matches than one LDS round holds (density 1.0), empty tiles (0.02, 0.0)."""
    from pasture_amd import conversion as cv
    layout = custom_point_type_big(hip)
    n = 300_007
    rec = random_records(layout, n, 21)
    mask = np.random.default_rng(22).random(n) < density
    src = HashMapBuffer.from_numpy(rec, layout)
    for kind in (HashMapBuffer, VectorBuffer):
        out = src.filter(kind, mask)
        if mask.any():
            assert cv.last_plan_kinds(hip) == ["static"], (kind.__name__, cv.last_plan_kinds(hip))
        assert out.len() == int(mask.sum())
        assert_same(out, rec[mask])


@pytest.mark.gpu
@pytest.mark.parametrize("out_kind", ["V", "H"])
@pytest.mark.parametrize("fmt", [0, 1, 2, 3, 6, 7, 8])
def test_filter_typed_las_points_takes_the_in_tree_streaming_kernels(hip, oracle, fmt, out_kind):
    """Typed LAS points of the formats without waveform packets (LasPointFormatN::layout(), las_types.rs) are compacted by kernels instantiated in the
    library (no run-time compilation): byte-identical to the oracle, plan family "static"."""
    from pasture_amd import conversion as cv
    from pasture_amd import las
    n = 70_001
    mask = np.random.default_rng(fmt).random(n) < 0.5

    def run(api):
        layout = las.point_layout_from_las_point_format(las.Format(fmt), False, api=api)
        src = HashMapBuffer.new_from_layout(layout)
        src.resize(n)
        src.synth_fill(40 + fmt, 0)
        out = src.filter(BUFFER_KINDS[out_kind], mask)
        kinds = cv.last_plan_kinds(api) if api is hip else None
        return out.len(), out.get_point_range(range(0, out.len())).tobytes(), kinds
    hn, hb, kinds = run(hip)
    on, ob, _ = run(oracle)
    assert hn == on == int(mask.sum()) and hb == ob
    assert kinds == ["static"], kinds
