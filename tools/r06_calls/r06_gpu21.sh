cd $GRAFT_REPO_ROOT
timeout 2400 python -m pytest tests/test_expressions.py tests/test_jit.py tests/test_buffer_converter.py tests/test_las_golden.py -x -q -m gpu 2>&1 | tail -5
python - <<'P'
import os, sys, time
sys.path.insert(0, '.')
import torch, numpy as np
import pasture_amd as pa
from pasture_amd import las, conversion as cv
from pasture_amd.algorithms import transform_attribute_expr
from pasture_amd.layout import attributes as A
layout = las.point_layout_from_las_point_format(las.Format(0), False)
n = 100_000_000
for kind in (pa.VectorBuffer, pa.HashMapBuffer):
    buf = kind.new_from_layout(layout); buf.resize(n); buf.synth_fill(42, 0)
    ts = []
    for it in range(6):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        transform_attribute_expr(buf, A.POSITION_3D, "v * 1.0000001 + (double)(i & 1)")
        ts.append((time.perf_counter() - t0) * 1e3)
    print(kind.__name__, "transform_attribute_expr 1e8 LAS-0 points: ms per call", [round(t, 3) for t in ts[1:]], "kinds", cv.last_plan_kinds())
P
