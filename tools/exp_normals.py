"""Experiment: kNN normal estimation throughput (configs[4]) for synthetic volume points."""
import ctypes, sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import pasture_amd as pa
from pasture_amd.layout import attributes as A, PointLayout, PointAttributeDefinition, PointAttributeDataType as T

api = pa.product_api()
s = torch.cuda.current_stream()
api.set_stream(ctypes.c_void_p(s.cuda_stream))
k = int(sys.argv[2]) if len(sys.argv) > 2 else 16
for n in [int(x) for x in sys.argv[1].split(",")]:
    src = pa.HashMapBuffer.new_from_layout(PointLayout.from_attributes([A.POSITION_3D])); src.resize(n); src.synth_fill(42, 0)
    out = pa.HashMapBuffer.new_from_layout(PointLayout.from_attributes([A.NORMAL, PointAttributeDefinition("Curvature", T.F64)])); out.resize(n)
    pa.compute_normals_into(src, k, out); torch.cuda.synchronize()
    t0 = time.perf_counter()
    reps = 3 if n <= 10_000_000 else 1
    for _ in range(reps): pa.compute_normals_into(src, k, out)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    print(f"normals n={n} k={k}: {dt*1e3:9.2f} ms  {n/dt/1e6:8.2f} Mpts/s  lower-bound traffic 44 B/pt -> {44*n/dt/1e9:7.1f} GB/s", flush=True)
