"""Does the relative placement of the source and target arrays matter to the headline kernel?  One process, one 6 GB torch allocation; source at
its start, target at (2.4 GB rounded up to 2 MiB) + skew; convert + affine + bounds timed with HIP events for a list of skews, each three times.
(Round 3: single bench runs spread over several percent between processes -- is it the virtual offset or the physical pages?)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import pasture_amd as pa
from pasture_amd.layout import attributes as A, PointAttributeDataType as T
from pasture_amd.buffers import ExternalColumnsBuffer

n = 100_000_000
layout = pa.PointLayout.from_attributes([A.POSITION_3D])
big = torch.empty(6 * 1024**3, dtype=torch.uint8, device="cuda")
base = big.data_ptr()
src = ExternalColumnsBuffer([base], layout, n)
src.synth_fill(7, 0)
rec = torch.zeros(6, dtype=torch.float64, device="cuda")
conv = pa.BufferLayoutConverter.for_layouts(layout, layout)
conv.set_custom_mapping_with_transformation(A.POSITION_3D, A.POSITION_3D, pa.Transform.affine(T.Vec3f64, (0.001, 0.001, 0.001), (10.0, 20.0, 30.0)), False)
span = ((n * 24 + (1 << 21) - 1) >> 21) << 21
for skew in [0, 256, 4096, 65536, 1 << 20, (1 << 21) + 4096, 3 << 20, 12345 * 256]:
    dst = ExternalColumnsBuffer([base + span + skew], layout, n)
    for _ in range(3):
        conv.convert_into_with_bounds_async(src, dst, rec.data_ptr())
    torch.cuda.synchronize()
    ts = []
    for rep in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            conv.convert_into_with_bounds_async(src, dst, rec.data_ptr())
        e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / 10)
    print(f"skew {skew:>9}: " + " ".join(f"{t:.4f}" for t in ts) + f"  ms  ({4.8e9 / min(ts) / 1e9:.0f} GB/s)")
