"""Experiment: kNN normals on a volume cloud plus a few far outliers (they stretch the bounding box): python tools/exp_knn_outliers.py [n] [n_outliers] [distance]"""
import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import pasture_amd as pa
from pasture_amd.layout import attributes as A, PointLayout
from pasture_amd.algorithms import compute_normals_device
api = pa.product_api(); s = torch.cuda.current_stream(); api.set_stream(ctypes.c_void_p(s.cuda_stream))
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
n_out = int(sys.argv[2]) if len(sys.argv) > 2 else 64
dist = float(sys.argv[3]) if len(sys.argv) > 3 else 20000.0
g = torch.Generator(device="cuda"); g.manual_seed(5)
pts = torch.rand(n, 3, device="cuda", dtype=torch.float64, generator=g) * torch.tensor([1000.0, 1000.0, 100.0], device="cuda", dtype=torch.float64)
if n_out:
    o = (torch.rand(n_out, 3, device="cuda", dtype=torch.float64, generator=g) - 0.5) * 2.0 * dist
    pts[torch.randint(0, n, (n_out,), device="cuda", generator=g)] = o
src = pa.ExternalColumnsBuffer([pts], PointLayout.from_attributes([A.POSITION_3D]), n)
curv = torch.empty(n, dtype=torch.float64, device="cuda")
compute_normals_device(src, 16, 0, curv.data_ptr(), 0); torch.cuda.synchronize()
t0 = time.perf_counter(); compute_normals_device(src, 16, 0, curv.data_ptr(), 0); torch.cuda.synchronize(); dt = time.perf_counter() - t0
print(f"volume cloud n={n} + {n_out} outliers within +-{dist}: {dt*1e3:.2f} ms  {n/dt/1e6:.1f} Mpts/s", flush=True)
