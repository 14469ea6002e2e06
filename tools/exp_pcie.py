"""Experiment: PCIe-inclusive rate of the headline step when the boundary hands over HOST buffers (DESIGN.md note)."""
import ctypes, sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import pasture_amd as pa
from pasture_amd.layout import attributes as A, PointLayout, PointAttributeDataType as T
api = pa.product_api(); s = torch.cuda.current_stream(); api.set_stream(ctypes.c_void_p(s.cuda_stream))
n = 100_000_000
layout = PointLayout.from_attributes([A.POSITION_3D])
host_in = torch.empty(n * 3, dtype=torch.float64).pin_memory(); host_in.uniform_(0, 1000)
host_out = torch.empty(n * 3, dtype=torch.float64).pin_memory()
dev_in = torch.empty(n * 3, dtype=torch.float64, device="cuda"); dev_out = torch.empty_like(dev_in)
src = pa.ExternalColumnsBuffer([dev_in], layout, n); dst = pa.ExternalColumnsBuffer([dev_out], layout, n)
conv = pa.BufferLayoutConverter.for_layouts(layout, layout)
conv.set_custom_mapping_with_transformation(A.POSITION_3D, A.POSITION_3D, pa.Transform.affine(T.Vec3f64, (0.001,) * 3, (5e5, 5.4e6, 100.0)), False)
rec = torch.empty(6, dtype=torch.float64, device="cuda")
def step():
    dev_in.copy_(host_in, non_blocking=True)
    conv.convert_into_with_bounds_async(src, dst, rec.data_ptr())
    host_out.copy_(dev_out, non_blocking=True)
for _ in range(2): step()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(5): step()
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
print(f"PCIe-inclusive (pinned H2D 2.4 GB + kernel + pinned D2H 2.4 GB): {dt*1e3:.1f} ms/step = {n/dt/1e6:.1f} Mpoints/s; H2D+D2H effective {4.8/dt:.1f} GB/s")
