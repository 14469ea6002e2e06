cd $GRAFT_REPO_ROOT
for f in 1 0; do PST_EXPR_FUSE=$f python - <<'P'
import os, sys, time
sys.path.insert(0, '.')
import torch
import pasture_amd as pa
from pasture_amd import las, conversion as cv
from pasture_amd.algorithms import transform_attribute_expr
from pasture_amd.layout import attributes as A
layout = las.point_layout_from_las_point_format(las.Format(0), False)
n = 100_000_000
buf = pa.VectorBuffer.new_from_layout(layout); buf.resize(n); buf.synth_fill(42, 0)
ts = []
for it in range(6):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    transform_attribute_expr(buf, A.POSITION_3D, "v * 1.0000001 + (double)(i & 1)")
    ts.append((time.perf_counter() - t0) * 1e3)
print("PST_EXPR_FUSE=" + os.environ["PST_EXPR_FUSE"], "VectorBuffer transform_attribute_expr, 1e8 typed LAS-0 records in place: ms per call", [round(t, 3) for t in ts[1:]], cv.last_plan_kinds())
P
done
