cd $GRAFT_REPO_ROOT
python /dev/stdin <<'PY' 2>&1 | grep -v amdgpu.ids | tail -3
import sys, os, numpy as np
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import test_gpu_parity as gp
from conftest import _load_oracle
import pasture_amd as pa
from pasture_amd.algorithms import compute_normals
from pasture_amd.buffers import HashMapBuffer
from pasture_amd.layout import PointLayout, attributes as A
hip, orc = pa.product_api(), _load_oracle()
(c, kind, pts, k), = gp._fuzz_module().cases(50_000 + 85, 1, more_kinds=True)
n = len(pts)
def run(api):
    buf = HashMapBuffer.new_from_layout(PointLayout.from_attributes([A.POSITION_3D], api=api))
    buf.resize(n); buf.set_attribute_range(A.POSITION_3D, range(0, n), pts)
    return compute_normals(buf, k, return_knn=True)
hn, hc, hk = run(hip); on, oc, ok = run(orc)
bad, cbad = gp._compare_normals(hn, hc, on, oc, scales=gp._cov_scales(pts, ok))
print(kind, "default fit: bad curvatures", cbad.sum(), "bad normals", bad.sum())
PY
for w in normals_knn16 normals_knn16_sheet normals_knn16_async; do
  python bench.py --workload $w --steps 8 --warmup 2 --no-cpu-baseline 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$w', 'ms_per_step', d['ms_per_step'])"
done
( time timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed" ) 2>&1 | tail -5
