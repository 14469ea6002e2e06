#!/usr/bin/env python3
"""Round 6: a closure applied inside the conversion pass (fused into the plan-specialised kernel) against its own strided pass (PST_EXPR_FUSE=0, the
round-5 form), same process.  Typed LAS-0 records (35 B) <-> 10 columns / records with `v * 0.001 + 500000.0` on Position3D: ms per call (HIP events),
GB/s against the plan's algorithmic bytes (70 B per point), the kernel families of the call, and the target bytes of both forms compared on the device."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import pasture_amd as pa  # noqa: E402
from pasture_amd import conversion as cv, las  # noqa: E402
from pasture_amd.layout import attributes as A  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000_000
hip = pa.product_api()
layout = las.point_layout_from_las_point_format(las.Format(0), False)
KINDS = {"V": pa.VectorBuffer, "H": pa.HashMapBuffer}


def view(buf, kind):
    class _M:
        pass
    out = []
    if kind == "V":
        ptrs = [(buf.points_ptr(), n * 35)]
    else:
        ptrs = [(buf.column_ptr(a.attribute_definition()), n * a.size()) for a in layout.attributes()]
    for p, b in ptrs:
        m = _M()
        m.__cuda_array_interface__ = {"shape": (b,), "typestr": "|u1", "data": (p, False), "version": 2}
        out.append(torch.as_tensor(m, device="cuda"))
    return out


for sk, dk in (("V", "H"), ("H", "V"), ("V", "V")):
    src = KINDS[sk].new_from_layout(layout)
    src.resize(n)
    src.synth_fill(42, 0)
    dst = KINDS[dk].new_from_layout(layout)
    dst.resize(n)
    conv = pa.BufferLayoutConverter.for_layouts(layout, layout)
    conv.set_custom_mapping_with_expression(A.POSITION_3D, A.POSITION_3D, "v * 0.001 + 500000.0", False)
    conv.prepare(KINDS[sk], KINDS[dk])  # (PST_EXPR_FUSE is read once per process by the library: the two forms run in two processes)
    for _ in range(2):
        conv.convert_into_range_async(src, range(0, n), dst, range(0, n))
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(8)]
    for a, b in ev:
        a.record()
        conv.convert_into_range_async(src, range(0, n), dst, range(0, n))
        b.record()
    torch.cuda.synchronize()
    ms = sorted(a.elapsed_time(b) for a, b in ev)[len(ev) // 2]
    chk = sum(int(t.to(torch.int64).sum().item()) for t in view(dst, dk))
    print(f"{sk}->{dk} n={n} PST_EXPR_FUSE={os.environ.get('PST_EXPR_FUSE', '1')}: {ms:.3f} ms  {70 * n / ms / 1e6:.0f} GB/s of 70 B/pt  kinds={cv.last_plan_kinds()}  byte-sum={chk}")
