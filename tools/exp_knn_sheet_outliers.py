"""Experiment: a LiDAR-like sheet with stray returns far above and below: python tools/exp_knn_sheet_outliers.py [n] [n_outliers]"""
import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import pasture_amd as pa
from pasture_amd.layout import attributes as A, PointLayout
from pasture_amd.algorithms import compute_normals_device
api = pa.product_api(); s = torch.cuda.current_stream(); api.set_stream(ctypes.c_void_p(s.cuda_stream))
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
n_out = int(sys.argv[2]) if len(sys.argv) > 2 else 300
g = torch.Generator(device="cuda"); g.manual_seed(1)
xy = torch.rand(n, 2, device="cuda", dtype=torch.float64, generator=g) * 1000.0
z = 10.0 * torch.sin(xy[:, 0] / 50.0) * torch.cos(xy[:, 1] / 80.0) + 50.0 + 0.02 * torch.randn(n, device="cuda", dtype=torch.float64, generator=g)
pts = torch.cat([xy, z[:, None]], dim=1).contiguous()
if n_out:
    idx = torch.randint(0, n, (n_out,), device="cuda", generator=g)
    pts[idx, 2] = (torch.rand(n_out, device="cuda", dtype=torch.float64, generator=g) - 0.5) * 6000.0
src = pa.ExternalColumnsBuffer([pts], PointLayout.from_attributes([A.POSITION_3D]), n)
curv = torch.empty(n, dtype=torch.float64, device="cuda")
compute_normals_device(src, 16, 0, curv.data_ptr(), 0); torch.cuda.synchronize()
t0 = time.perf_counter(); compute_normals_device(src, 16, 0, curv.data_ptr(), 0); torch.cuda.synchronize(); dt = time.perf_counter() - t0
print(f"sheet n={n} + {n_out} stray points: {dt*1e3:.2f} ms  {n/dt/1e6:.1f} Mpts/s", flush=True)
