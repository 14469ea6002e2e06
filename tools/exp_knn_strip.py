"""Experiment: a diagonal flight strip (a long narrow LiDAR sheet at an angle to the axes): python tools/exp_knn_strip.py [n] [angle_deg] [length] [width]"""
import ctypes, math, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import pasture_amd as pa
from pasture_amd.layout import attributes as A, PointLayout
from pasture_amd.algorithms import compute_normals_device
api = pa.product_api(); s = torch.cuda.current_stream(); api.set_stream(ctypes.c_void_p(s.cuda_stream))
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
ang = math.radians(float(sys.argv[2]) if len(sys.argv) > 2 else 40.0)
L = float(sys.argv[3]) if len(sys.argv) > 3 else 8000.0
W = float(sys.argv[4]) if len(sys.argv) > 4 else 300.0
g = torch.Generator(device="cuda"); g.manual_seed(1)
a = torch.rand(n, device="cuda", dtype=torch.float64, generator=g) * L
b = torch.rand(n, device="cuda", dtype=torch.float64, generator=g) * W
z = 10.0 * torch.sin(a / 50.0) * torch.cos(b / 80.0) + 50.0 + 0.02 * torch.randn(n, device="cuda", dtype=torch.float64, generator=g)
x = math.cos(ang) * a - math.sin(ang) * b + 500000.0
y = math.sin(ang) * a + math.cos(ang) * b + 5400000.0
pts = torch.stack([x, y, z], dim=1).contiguous()
src = pa.ExternalColumnsBuffer([pts], PointLayout.from_attributes([A.POSITION_3D]), n)
curv = torch.empty(n, dtype=torch.float64, device="cuda")
compute_normals_device(src, 16, 0, curv.data_ptr(), 0); torch.cuda.synchronize()
t0 = time.perf_counter(); compute_normals_device(src, 16, 0, curv.data_ptr(), 0); torch.cuda.synchronize(); dt = time.perf_counter() - t0
print(f"diagonal strip n={n} {L:g} x {W:g} at {math.degrees(ang):g} deg: {dt*1e3:.2f} ms  {n/dt/1e6:.1f} Mpts/s", flush=True)
