"""Per-call wall time of compute_normals_into at 10^8 points (are there host-side stalls?)."""
import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import pasture_amd as pa
from pasture_amd.layout import attributes as A, PointLayout, PointAttributeDefinition, PointAttributeDataType as T
api = pa.product_api(); s = torch.cuda.current_stream(); api.set_stream(ctypes.c_void_p(s.cuda_stream))
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000_000
layout = PointLayout.from_attributes([A.POSITION_3D])
src = pa.HashMapBuffer.new_from_layout(layout); src.resize(n); src.synth_fill(42, 0)
dst = pa.HashMapBuffer.new_from_layout(PointLayout.from_attributes([A.NORMAL, PointAttributeDefinition("Curvature", T.F64)])); dst.resize(n)
ts = []
for i in range(int(sys.argv[2]) if len(sys.argv) > 2 else 30):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    pa.compute_normals_into(src, 16, dst)
    torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
print(" ".join(f"{t:.0f}" for t in ts))
