"""Experiment: kNN normals at other k (the bench workload is k = 16): python tools/exp_knn_k.py [n] [k ...]"""
import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import pasture_amd as pa
from pasture_amd.layout import attributes as A, PointLayout, PointAttributeDefinition, PointAttributeDataType as T
api = pa.product_api(); s = torch.cuda.current_stream(); api.set_stream(ctypes.c_void_p(s.cuda_stream))
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
ks = [int(v) for v in sys.argv[2:]] or [4, 8, 12, 16, 24, 32, 40]
src = pa.HashMapBuffer.new_from_layout(PointLayout.from_attributes([A.POSITION_3D])); src.resize(n); src.synth_fill(42, 0)
out = pa.HashMapBuffer.new_from_layout(PointLayout.from_attributes([A.NORMAL, PointAttributeDefinition("Curvature", T.F64)])); out.resize(n)
for k in ks:
    pa.compute_normals_into(src, k, out); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3): pa.compute_normals_into(src, k, out)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 3
    print(f"k={k:3d} n={n}: {dt*1e3:8.2f} ms  {n/dt/1e6:8.1f} Mpts/s", flush=True)
