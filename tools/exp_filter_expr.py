#!/usr/bin/env python3
"""Round 6: HashMapBuffer::filter with the predicate INSIDE the compaction kernels (count pass + streaming scatter pass, no byte mask) against the round-5
form (predicate -> byte mask -> count -> scatter; PST_EXPR_FUSE=0, read once per process).  Typed LAS-0 points, 10^8, `Classification < 128` (density 0.5):
wall time of the synchronous call (count read-back and target allocation included), median of 7, and the sum of the selected Intensity values."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import pasture_amd as pa  # noqa: E402
from pasture_amd import conversion as cv, las  # noqa: E402
from pasture_amd.layout import attributes as A  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000_000
layout = las.point_layout_from_las_point_format(las.Format(0), False)
src = pa.HashMapBuffer.new_from_layout(layout)
src.resize(n)
src.synth_fill(42, 0)
for name, kind in (("columns", pa.HashMapBuffer), ("records", pa.VectorBuffer)):
    for text in ("Classification < 128", "Classification < 128 && Position3D.z < 50.0"):
        ts = []
        for it in range(8):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            out = src.filter_expr(kind, text)
            ts.append((time.perf_counter() - t0) * 1e3)
            m = out.len()
            chk = int(np.asarray(out.get_attribute_range(A.INTENSITY, range(0, min(m, 1 << 20)))).astype(np.int64).sum()) if it == 0 else chk
            del out
        ms = sorted(ts[1:])[3]
        print(f"filter -> {name} `{text}` PST_EXPR_FUSE={os.environ.get('PST_EXPR_FUSE', '1')}: {ms:.3f} ms per call, {m} of {n} selected, kinds={cv.last_plan_kinds()}, intensity-sum(first 2^20)={chk}")
