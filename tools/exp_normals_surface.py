"""Experiment: kNN normals on a SURFACE-like cloud (LiDAR-like 2.5D), where a volume-density cell size is wrong."""
import ctypes, sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import pasture_amd as pa
from pasture_amd.layout import attributes as A, PointLayout, PointAttributeDefinition, PointAttributeDataType as T
api = pa.product_api(); s = torch.cuda.current_stream(); api.set_stream(ctypes.c_void_p(s.cuda_stream))
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
noise = float(sys.argv[2]) if len(sys.argv) > 2 else 0.02
g = torch.Generator(device="cuda"); g.manual_seed(1)
xy = torch.rand(n, 2, device="cuda", dtype=torch.float64, generator=g) * 1000.0
z = 10.0 * torch.sin(xy[:, 0] / 50.0) * torch.cos(xy[:, 1] / 80.0) + 50.0 + noise * torch.randn(n, device="cuda", dtype=torch.float64, generator=g)
pos = torch.cat([xy, z[:, None]], dim=1).contiguous()
layout = PointLayout.from_attributes([A.POSITION_3D])
src = pa.ExternalColumnsBuffer([pos], layout, n)
out = pa.HashMapBuffer.new_from_layout(PointLayout.from_attributes([A.NORMAL, PointAttributeDefinition("Curvature", T.F64)])); out.resize(n)
pa.compute_normals_into(src, 16, out); torch.cuda.synchronize()
t0 = time.perf_counter(); pa.compute_normals_into(src, 16, out); torch.cuda.synchronize(); dt = time.perf_counter() - t0
nrm = torch.as_tensor(out.view_attribute(A.NORMAL)[:5])
print(f"surface cloud n={n} noise={noise}: {dt*1e3:.2f} ms  {n/dt/1e6:.1f} Mpts/s; first normals (unnormalised) {nrm.tolist()[:2]}", flush=True)
