"""Experiment: kNN normals on two far-apart clusters (the bounding box's volume says nothing about their scale): python tools/exp_knn_clusters.py [n] [k]"""
import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import pasture_amd as pa
from pasture_amd.layout import attributes as A, PointLayout
api = pa.product_api(); s = torch.cuda.current_stream(); api.set_stream(ctypes.c_void_p(s.cuda_stream))
n = int(sys.argv[1]) if len(sys.argv) > 1 else 600_000
k = int(sys.argv[2]) if len(sys.argv) > 2 else 16
g = torch.Generator(device="cuda"); g.manual_seed(3)
a = torch.rand(n // 2, 3, device="cuda", dtype=torch.float64, generator=g) * 5.0
b = torch.rand(n - n // 2, 3, device="cuda", dtype=torch.float64, generator=g) * 3.0 + torch.tensor([4000.0, 2500.0, 900.0], device="cuda", dtype=torch.float64)
pts = torch.cat([a, b])[torch.randperm(n, device="cuda", generator=g)].contiguous()
src = pa.ExternalColumnsBuffer([pts], PointLayout.from_attributes([A.POSITION_3D]), n)
curv = torch.empty(n, dtype=torch.float64, device="cuda")
from pasture_amd.algorithms import compute_normals_device
compute_normals_device(src, k, 0, curv.data_ptr(), 0); torch.cuda.synchronize()
t0 = time.perf_counter(); compute_normals_device(src, k, 0, curv.data_ptr(), 0); torch.cuda.synchronize(); dt = time.perf_counter() - t0
print(f"two clusters n={n} k={k}: {dt*1e3:.2f} ms  {n/dt/1e6:.1f} Mpts/s", flush=True)
