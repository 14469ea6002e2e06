import sys, os, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import _load_oracle
import pasture_amd as pa
from pasture_amd.algorithms import compute_normals
from pasture_amd.buffers import HashMapBuffer
from pasture_amd.layout import PointLayout, attributes as A
hip, orc = pa.product_api(), _load_oracle()
rng = np.random.default_rng(3)
cases = {
  "flat_plane": np.column_stack([rng.random((20000, 2)) * 300.0, np.full(20000, 7.25)]),
  "line_x": np.column_stack([rng.random(6000) * 1000.0, np.full(6000, 1.0), np.full(6000, -2.0)]),
  "two_planes": np.concatenate([np.column_stack([rng.random((15000, 2)) * 200.0, np.zeros(15000)]), np.column_stack([rng.random((15000, 2)) * 200.0, np.full(15000, 150.0)])]),
  "huge_offsets": rng.random((30000, 3)) * np.array([300.0, 300.0, 30.0]) + np.array([5.4e6, 5.0e5, 100.0]),
  "tiny_cluster_plus_far_point": np.concatenate([rng.random((9000, 3)) * 1e-3, np.array([[1e6, 1e6, 1e6]])]),
}
for name, pts in cases.items():
    res = {}
    for tag, api in (("hip", hip), ("orc", orc)):
        buf = HashMapBuffer.new_from_layout(PointLayout.from_attributes([A.POSITION_3D], api=api)); buf.resize(len(pts))
        buf.set_attribute_range(A.POSITION_3D, range(0, len(pts)), pts)
        try:
            res[tag] = compute_normals(buf, 16, return_knn=True)
        except Exception as e:
            res[tag] = repr(e)[:120]
    if isinstance(res["hip"], str) or isinstance(res["orc"], str):
        print(name, "hip:", res["hip"] if isinstance(res["hip"], str) else "ok", "| orc:", res["orc"] if isinstance(res["orc"], str) else "ok")
        continue
    (hn, hc, hk), (on, oc, ok) = res["hip"], res["orc"]
    d_h = ((pts[hk] - pts[:, None, :]) ** 2).sum(2); d_o = ((pts[ok] - pts[:, None, :]) ** 2).sum(2)
    same_idx = np.array_equal(hk, ok)
    print(name, "knn identical:", same_idx, "| neighbour distances identical:", np.array_equal(d_h, d_o), "| max |dn|:", np.abs(hn - on).max(), "| max |dc|:", np.abs(hc - oc).max())
