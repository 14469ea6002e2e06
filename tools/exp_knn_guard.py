#!/usr/bin/env python3
"""How often the per-query conditioning guard of the one-pass plane fit fires, and what it changes: the structured volume of
tests/test_gpu_parity.py (_structured_volume) and a uniform cloud, PST_KNN_FIT=pivot with PST_KNN_FIT_GUARD=1 / 0 and PST_KNN_FIT=seq:
queries whose result differs between guarded and unguarded (= lanes that took the fallback and got another value), and the worst
differences of both against the reference-order instance."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import pasture_amd as pa  # noqa: E402
from pasture_amd.algorithms import compute_normals_device, reload_tuning  # noqa: E402
from pasture_amd.buffers import ExternalColumnsBuffer  # noqa: E402
from pasture_amd.layout import PointLayout, attributes as A  # noqa: E402
import test_gpu_parity as gp  # noqa: E402

hip = pa.product_api()


def run(pts, k, env):
    for key in ("PST_KNN_FIT", "PST_KNN_FIT_GUARD"):
        os.environ.pop(key, None)
    os.environ.update(env)
    reload_tuning(hip)
    n = pts.shape[0]
    src = ExternalColumnsBuffer([pts], PointLayout.from_attributes([A.POSITION_3D], api=hip), n)
    nrm = torch.empty((n, 3), dtype=torch.float64, device="cuda")
    cur = torch.empty(n, dtype=torch.float64, device="cuda")
    knn = torch.empty((n, k), dtype=torch.int32, device="cuda")
    compute_normals_device(src, k, nrm.data_ptr(), cur.data_ptr(), knn.data_ptr())
    torch.cuda.synchronize()
    return nrm, cur, knn


for name, pts in (("structured volume", gp._structured_volume(1_200_000, 105, None)), ("structured volume, quantised 0.001", gp._structured_volume(1_200_000, 105, 0.001)),
                  ("uniform 4 10^6", torch.rand(4_000_000, 3, device="cuda", dtype=torch.float64) * torch.tensor([1000.0, 1000.0, 100.0], device="cuda", dtype=torch.float64))):
    for k in (5, 16):
        sn, sc, sk = run(pts, k, {"PST_KNN_FIT": "seq"})
        if k == 16:
            os.environ["PST_KNN_DEBUG"] = "1"  # which search this cloud takes under the DEFAULT dispatch
            run(pts, k, {})
            os.environ.pop("PST_KNN_DEBUG", None)
        gn, gc, gk = run(pts, k, {"PST_KNN_FIT": "pivot"})
        un, uc, uk = run(pts, k, {"PST_KNN_FIT": "pivot", "PST_KNN_FIT_GUARD": "0"})
        # (a query the guard hands to the exact search gets its list from THAT kernel: the same distances, possibly another order among exact ties)
        same = (sk == gk).all(dim=1) & (sk == uk).all(dim=1)
        changed = ((gn != un).any(dim=1) | (gc != uc)).sum().item()
        same_as_seq = ((gn == sn).all(dim=1) & (gc == sc)).sum().item()
        # the parity window of tests/test_gpu_parity.py::_compare_normals per query: 1e-9 |c| + max(1e-12, 1e-13 scale), scale = max |entry| of the covariance
        nb = pts[sk.long().reshape(-1)].view(pts.shape[0], k, 3)
        d = nb - nb.mean(dim=1, keepdim=True)
        scale = torch.einsum("nki,nkj->nij", d, d).abs().reshape(-1, 9).max(dim=1).values
        window = 1e-9 * sc.abs() + torch.clamp(1e-13 * scale, min=1e-12)
        del nb, d

        def worst(n_, c_):
            rn = ((n_ - sn).norm(dim=1) / sn.norm(dim=1).clamp_min(1e-300))[same].max().item()
            ratio = ((c_ - sc).abs() / window)[same]
            i = int(ratio.argmax())
            return f"rel. normal {rn:.2e}, curvature difference / window {ratio[i].item():.3f} (|diff| {(c_ - sc).abs()[same][i].item():.2e}, scale {scale[same][i].item():.3g}, curvature {sc[same][i].item():.3g})"
        print(f"{name}, n = {pts.shape[0]}, k = {k}: lists identical on {int(same.sum())}; guard changed {changed} queries; guarded == reference-order bit for bit on {same_as_seq}\n"
              f"    guarded   vs reference-order: {worst(gn, gc)}\n    unguarded vs reference-order: {worst(un, uc)}", flush=True)
