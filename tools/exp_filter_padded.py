"""Experiment: compaction of a repr(C) layout (padding inside the records) into a VectorBuffer and into columns, 10^8 points, density 0.5:
the streaming kernel (compiled before the timed region) against the gather kernel (PST_FILTER_STREAM=0).  Run once per setting."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import pasture_amd as pa
from pasture_amd import conversion as cv
from pasture_amd.layout import PointAttributeDataType as T, PointAttributeDefinition as D, PointLayout
api = pa.product_api(); s = torch.cuda.current_stream(); api.set_stream(ctypes.c_void_p(s.cuda_stream))
n = 100_000_000
if len(sys.argv) > 1 and sys.argv[1] == "narrow":  # the attributes of a typed LAS-0 point in a repr(C) record: 35 bytes written, 40 per record
    from pasture_amd import las
    layout = PointLayout.from_attributes([a.attribute_definition() for a in las.point_layout_from_las_point_format(las.Format(0), False).attributes()])
else:  # wide attributes: 52 bytes written, 64 per record
    layout = PointLayout.from_attributes([D("t", T.F64), D("i", T.U16), D("c", T.Vec3f32), D("k", T.U8), D("p", T.Vec3f64), D("b", T.ByteArray(5))])
written = sum(a.size() for a in layout.attributes())
src = pa.HashMapBuffer.new_from_layout(layout); src.resize(n); src.synth_fill(7, 0)
mask = (torch.rand(n, device="cuda") < 0.5).to(torch.uint8)
k = int(mask.sum().item())
hits = torch.zeros(1, dtype=torch.int64, device="cuda")
for kind in (pa.VectorBuffer, pa.HashMapBuffer):
    dst = kind.new_from_layout(layout); dst.resize(k)
    cv.jit_set_mode("sync"); src.filter_into_async(dst, mask.data_ptr(), k, hits.data_ptr()); cv.jit_set_mode("env")
    for _ in range(3): src.filter_into_async(dst, mask.data_ptr(), k, hits.data_ptr())
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(s)
    for _ in range(10): src.filter_into_async(dst, mask.data_ptr(), k, hits.data_ptr())
    e1.record(s); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    # bytes: 2 mask reads + the attributes read + written for half the points; a padded record target is also READ (its padding survives)
    stride = layout.size_of_point_entry()
    b = 2 + written + (0.5 * (2 * stride) if kind is pa.VectorBuffer else 0.5 * written)
    print(f"{kind.__name__}: record {stride} B ({written} written)  {ms:.4f} ms  {b * n / ms / 1e9 / 8:.4f} of peak  plan={cv.last_plan_kinds()}  PST_FILTER_STREAM={os.environ.get('PST_FILTER_STREAM', '1')}", flush=True)
    del dst
