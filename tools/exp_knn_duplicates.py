"""Experiment: many coincident points (overlapping scans, a degenerate export): python tools/exp_knn_duplicates.py [n] [n_distinct]"""
import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import pasture_amd as pa
from pasture_amd.layout import attributes as A, PointLayout
from pasture_amd.algorithms import compute_normals_device
api = pa.product_api(); s = torch.cuda.current_stream(); api.set_stream(ctypes.c_void_p(s.cuda_stream))
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
m = int(sys.argv[2]) if len(sys.argv) > 2 else 1
g = torch.Generator(device="cuda"); g.manual_seed(2)
base = torch.rand(m, 3, device="cuda", dtype=torch.float64, generator=g) * 100.0
pts = base[torch.randint(0, m, (n,), device="cuda", generator=g)].contiguous()
src = pa.ExternalColumnsBuffer([pts], PointLayout.from_attributes([A.POSITION_3D]), n)
curv = torch.empty(n, dtype=torch.float64, device="cuda")
t0 = time.perf_counter()
try:
    compute_normals_device(src, 16, 0, curv.data_ptr(), 0)
except Exception as e:
    print("raised:", str(e)[:100])
torch.cuda.synchronize(); dt = time.perf_counter() - t0
print(f"{n} points on {m} distinct positions: {dt*1e3:.1f} ms", flush=True)
