"""Experiment (round 4, review item 6): does the placement of the source / target columns move the headline kernel?
Same process, same kernel: the fused convert + affine + AABB step over 10^8 points, with the two columns carved out of ONE big allocation at
chosen byte offsets (ExternalColumnsBuffer), so that base alignment (256 B ... 2 MiB) and the src - dst distance are the only things that
change.  Prints TB/s of the 48 B/point per configuration, three rounds (run-to-run noise shows as the spread between rounds)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import pasture_amd as pa
from pasture_amd.layout import attributes as A, PointAttributeDataType as T

api = pa.product_api(); s = torch.cuda.current_stream(); api.set_stream(ctypes.c_void_p(s.cuda_stream))
n = int(os.environ.get("N", 100_000_000))
col = 24 * n
MiB = 1 << 20
pool = torch.empty(2 * col + 64 * MiB, dtype=torch.uint8, device="cuda")
base = pool.data_ptr()
align_up = lambda a, m: (a + m - 1) // m * m
b2 = align_up(base, 2 * MiB) - base  # offset of the first 2 MiB boundary inside the pool
layout = pa.PointLayout.from_attributes([A.POSITION_3D])
rec = torch.empty(6, dtype=torch.float64, device="cuda")
conv = pa.BufferLayoutConverter.for_layouts(layout, layout)
conv.set_custom_mapping_with_transformation(A.POSITION_3D, A.POSITION_3D, pa.Transform.affine(T.Vec3f64, (0.001,) * 3, (500000.0, 5400000.0, 100.0)), False)

def run(src_off, dst_off, steps=20):
    src = pa.ExternalColumnsBuffer([base + src_off], layout, n)
    dst = pa.ExternalColumnsBuffer([base + dst_off], layout, n)
    src.synth_fill(42, 0)
    for _ in range(3): conv.convert_into_with_bounds_async(src, dst, rec.data_ptr())
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(s)
    for _ in range(steps): conv.convert_into_with_bounds_async(src, dst, rec.data_ptr())
    e1.record(s); torch.cuda.synchronize()
    return 48 * n * steps / (e0.elapsed_time(e1) * 1e-3) / 1e12

gap = align_up(col, 2 * MiB)  # dst starts at the next 2 MiB boundary behind src
configs = [
    ("src 2MiB-aligned, dst 2MiB-aligned", b2, b2 + gap),
    ("src +256 B, dst 2MiB-aligned", b2 + 256, b2 + gap + 2 * MiB),
    ("src 2MiB, dst +256 B", b2, b2 + gap + 256),
    ("src 2MiB, dst +4 KiB", b2, b2 + gap + 4096),
    ("src 2MiB, dst +64 KiB", b2, b2 + gap + 65536),
    ("src 2MiB, dst +1 MiB", b2, b2 + gap + MiB),
    ("src +256 B, dst +256 B", b2 + 256, b2 + gap + 2 * MiB + 256),
    ("src +4 KiB + 256, dst +36 KiB + 512", b2 + 4096 + 256, b2 + gap + 2 * MiB + 36864 + 512),
    ("dst directly behind src (256-B rounded)", b2, b2 + align_up(col, 256)),
]
for rnd in range(3):
    for name, so, do in configs:
        print(f"round {rnd}  {name:42s} {run(so, do):6.3f} TB/s  ({run(so, do) / 8.0:5.3f} of peak)", flush=True)
# and the library's own allocations (the stream-ordered pool), for reference
srcb = pa.HashMapBuffer.new_from_layout(layout); srcb.resize(n); srcb.synth_fill(42, 0)
dstb = pa.HashMapBuffer.new_from_layout(layout); dstb.resize(n)
for rnd in range(3):
    for _ in range(3): conv.convert_into_with_bounds_async(srcb, dstb, rec.data_ptr())
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(s)
    for _ in range(20): conv.convert_into_with_bounds_async(srcb, dstb, rec.data_ptr())
    e1.record(s); torch.cuda.synchronize()
    print(f"round {rnd}  library pool buffers (src % 2MiB = {srcb.column_ptr(A.POSITION_3D) % (2*MiB)}, dst % 2MiB = {dstb.column_ptr(A.POSITION_3D) % (2*MiB)}) "
          f"{48 * n * 20 / (e0.elapsed_time(e1) * 1e-3) / 1e12:6.3f} TB/s", flush=True)
