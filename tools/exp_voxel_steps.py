"""Per-call wall time of voxelgrid_filter at 10^8 points (host-side stalls?)."""
import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import pasture_amd as pa
from pasture_amd.layout import attributes as A, PointLayout
api = pa.product_api(); s = torch.cuda.current_stream(); api.set_stream(ctypes.c_void_p(s.cuda_stream))
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000_000
layout = PointLayout.from_attributes([A.POSITION_3D])
src = pa.HashMapBuffer.new_from_layout(layout); src.resize(n); src.synth_fill(42, 0)
ts = []
for i in range(int(sys.argv[2]) if len(sys.argv) > 2 else 60):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    out = pa.HashMapBuffer.new_from_layout(layout)
    pa.voxelgrid_filter(src, 2.5, 2.5, 2.5, out)
    torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
print(" ".join(f"{t:.0f}" for t in ts))
