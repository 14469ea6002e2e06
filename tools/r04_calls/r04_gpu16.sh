cd $GRAFT_REPO_ROOT
cat > /tmp/repro.py <<'PY'
import sys, os, importlib.util, numpy as np
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
print(kind, n, k, "lists equal", np.array_equal(hk, ok), "fit", os.environ.get("PST_KNN_FIT"), "debug follows")
scales = gp._cov_scales(pts, ok)
bad, cbad = gp._compare_normals(hn, hc, on, oc, scales=scales)
for q in np.nonzero(cbad)[0]:
    print("query", q, "hip curv", hc[q], "oracle", oc[q], "diff", abs(hc[q]-oc[q]), "scale", scales[q], "floor", max(1e-12, 1e-13*scales[q]), "normal rel", np.linalg.norm(hn[q]-on[q])/np.linalg.norm(on[q]))
print("bad curvatures", cbad.sum())
PY
PST_KNN_DEBUG=1 python /tmp/repro.py 2>&1 | grep -v amdgpu.ids | tail -12
PST_KNN_FIT=seq python /tmp/repro.py 2>&1 | grep -v amdgpu.ids | tail -4
