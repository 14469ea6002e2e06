"""Experiment: in-place transform_attribute (affine on POSITION_3D) for columnar and interleaved LAS-0 buffers."""
import ctypes, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import pasture_amd as pa
from pasture_amd import las
from pasture_amd.layout import attributes as A, PointAttributeDataType as T
api = pa.product_api(); s = torch.cuda.current_stream(); api.set_stream(ctypes.c_void_p(s.cuda_stream))
n = 100_000_000
las0 = las.point_layout_from_las_point_format(las.Format(0), False)
xf = pa.Transform.affine(T.Vec3f64, (1.0001, 0.9999, 1.0), (1.0, -2.0, 0.5))
for name, cls in (("columnar", pa.HashMapBuffer), ("interleaved", pa.VectorBuffer)):
    buf = cls.new_from_layout(las0); buf.resize(n); buf.synth_fill(42, 0)
    for _ in range(2): pa.transform_attribute(buf, A.POSITION_3D, xf)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(s)
    for _ in range(5): pa.transform_attribute(buf, A.POSITION_3D, xf)
    e1.record(s); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 5
    print(f"transform_attribute {name:12s}: {ms:7.3f} ms  {n / ms / 1e6:6.1f} Gpts/s  {48 * n / ms / 1e9:5.2f} TB/s algorithmic (24 R + 24 W)", flush=True)
