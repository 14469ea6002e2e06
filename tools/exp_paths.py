"""Experiment: throughput of the secondary paths at 10^8 points (which kernels need work next)."""
import ctypes, sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import pasture_amd as pa
from pasture_amd import las
from pasture_amd.layout import attributes as A, PointLayout, PointAttributeDefinition, PointAttributeDataType as T

api = pa.product_api()
s = torch.cuda.current_stream()
api.set_stream(ctypes.c_void_p(s.cuda_stream))
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000_000

def timeit(name, fn, bytes_pp, reps=10):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(s)
    for _ in range(reps): fn()
    e1.record(s); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    print(f"{name:58s} {ms:8.3f} ms {n/ms/1e3:9.1f} Mpts/s {bytes_pp*n/ms/1e6:8.1f} GB/s ({bytes_pp} B/pt)", flush=True)

las0 = las.point_layout_from_las_point_format(las.Format(0), False)
aos = pa.VectorBuffer.new_from_layout(las0); aos.resize(n); aos.synth_fill(42, 0)
soa = pa.HashMapBuffer.new_from_layout(las0); soa.resize(n); soa.synth_fill(42, 0)
rec = torch.empty(6, dtype=torch.float64, device="cuda")

timeit("calculate_bounds interleaved LAS-0 (35 B stride)", lambda: pa.calculate_bounds_async(aos, rec.data_ptr()), 35)
timeit("calculate_bounds columnar", lambda: pa.calculate_bounds_async(soa, rec.data_ptr()), 24)
timeit("minmax_attribute columnar Intensity u16 (sync API)", lambda: pa.minmax_attribute(soa, A.INTENSITY), 2, reps=5)
timeit("minmax_attribute columnar Position3D (sync API)", lambda: pa.minmax_attribute(soa, A.POSITION_3D), 24, reps=5)
timeit("minmax_attribute interleaved Intensity u16 (sync API)", lambda: pa.minmax_attribute(aos, A.INTENSITY), 35, reps=5)

conv_id = pa.BufferLayoutConverter.for_layouts(las0, las0)
aos2 = pa.VectorBuffer.new_from_layout(las0); aos2.resize(n)
soa2 = pa.HashMapBuffer.new_from_layout(las0); soa2.resize(n)
r = range(0, n)
timeit("convert_into interleaved -> columnar (K3)", lambda: conv_id.convert_into_range_async(aos, r, soa2, r), 70)
timeit("convert_into columnar -> interleaved (K3')", lambda: conv_id.convert_into_range_async(soa, r, aos2, r), 70)
timeit("convert_into interleaved -> interleaved (identity)", lambda: conv_id.convert_into_range_async(aos, r, aos2, r), 70)
timeit("convert_into columnar -> columnar (10 column copies)", lambda: conv_id.convert_into_range_async(soa, r, soa2, r), 70)

# datatype-changing conversions (layout_conversion_bench.rs layouts: 35 B -> 25 B)
sl = PointLayout.from_attributes_packed([A.POSITION_3D, A.CLASSIFICATION, A.INTENSITY, A.GPS_TIME], 1)
tl = PointLayout.from_attributes_packed([A.GPS_TIME, A.POSITION_3D.with_custom_datatype(T.Vec3f32), A.CLASSIFICATION.with_custom_datatype(T.U32),
                                         A.INTENSITY.with_custom_datatype(T.U8)], 1)
conv_b = pa.BufferLayoutConverter.for_layouts(sl, tl)
bufs = {}
for kind, cls in (("V", pa.VectorBuffer), ("H", pa.HashMapBuffer)):
    b = cls.new_from_layout(sl); b.resize(n); b.synth_fill(1, 0); bufs["s" + kind] = b
    d = cls.new_from_layout(tl); d.resize(n); bufs["t" + kind] = d
for a in "VH":
    for b in "VH":
        timeit(f"bench layouts 35 B -> 25 B with `as` casts: {a} -> {b}", lambda a=a, b=b: conv_b.convert_into_range_async(bufs["s" + a], r, bufs["t" + b], r), 60)

pl = PointLayout.from_attributes([A.POSITION_3D])
p32 = PointLayout.from_attributes([A.POSITION_3D.with_custom_datatype(T.Vec3f32)])
ps = pa.HashMapBuffer.new_from_layout(pl); ps.resize(n); ps.synth_fill(42, 0)
pd = pa.HashMapBuffer.new_from_layout(p32); pd.resize(n)
conv_n = pa.BufferLayoutConverter.for_layouts(pl, p32)
timeit("columnar Vec3f64 -> Vec3f32 narrowing (K5)", lambda: conv_n.convert_into_range_async(ps, r, pd, r), 36)
timeit("convert() allocating columnar target from interleaved", lambda: conv_id.convert(aos, pa.HashMapBuffer), 70, reps=3)
