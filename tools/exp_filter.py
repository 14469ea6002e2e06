"""Experiment: predicate compaction (HashMapBuffer::filter_into) at 10^8 points, mask resident in HBM."""
import ctypes, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import pasture_amd as pa
from pasture_amd import las
from pasture_amd.layout import attributes as A, PointLayout, PointAttributeDataType as T
api = pa.product_api(); s = torch.cuda.current_stream(); api.set_stream(ctypes.c_void_p(s.cuda_stream))
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000_000
big = PointLayout.from_attributes_packed([A.GPS_TIME, A.COLOR_RGB, A.POSITION_3D, A.CLASSIFICATION, A.INTENSITY.with_custom_datatype(T.I16)], 1)
las0 = las.point_layout_from_las_point_format(las.Format(0), False)
xyz = PointLayout.from_attributes([A.POSITION_3D])
for name, layout in (("CustomPointTypeBig 41 B", big), ("LAS-0 35 B", las0), ("XYZ 24 B", xyz)):
    src = pa.HashMapBuffer.new_from_layout(layout); src.resize(n); src.synth_fill(42, 0)
    sz = layout.size_of_point_entry()
    for density in (0.5, 0.1, 0.9):
        mask = (torch.rand(n, device="cuda") < density).to(torch.uint8)
        k = int(mask.sum().item())
        for kind, cls in (("H", pa.HashMapBuffer), ("V", pa.VectorBuffer)):
            dst = cls.new_from_layout(layout); dst.resize(k)
            for _ in range(2): src.filter_into(dst, (mask.data_ptr(), "device"), k)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(s)
            for _ in range(5): src.filter_into(dst, (mask.data_ptr(), "device"), k)
            e1.record(s); torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / 5
            b = 2 + sz + sz * k / n
            print(f"{name:24s} d={density:.1f} -> {kind}: {ms:7.3f} ms {n / ms / 1e6:7.1f} Gpts/s in  {n * b / ms / 1e9:6.2f} TB/s ({b:.1f} B/pt)", flush=True)
            del dst
    del src
