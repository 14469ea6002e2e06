"""Differential fuzz of the kNN search on clouds that are not a filled box (run on the GPU box): random planes / strips / clusters / halos /
outliers at random orientations, sizes and k, every neighbour list compared with the CPU oracle.  python tools/fuzz_knn_sparse.py [cases] [seed] [only,these,cases]   (FUZZ_KINDS=1: also filled boxes, sheets, lattices,
density contrasts; FUZZ_ONLY=kind,kind: only those; FUZZ_KS=40,64: these k in turn; FUZZ_BIG=1: 1.05 - 1.6 million points)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np


def rot(rng):
    q, _ = np.linalg.qr(rng.normal(size=(3, 3)))
    return q


def cloud(rng, big=False, more_kinds=False):
    n = int(rng.integers(1_050_000, 1_600_000)) if big else int(rng.integers(66_000, 220_000))
    kinds = ["slab", "strip", "clusters", "halo", "outliers", "line"] + (["volume", "sheet", "lattice", "dense_core"] if more_kinds else [])
    if os.environ.get("FUZZ_ONLY"):
        kinds = os.environ["FUZZ_ONLY"].split(",")
    kind = rng.choice(kinds)
    R, off = rot(rng), rng.choice([0.0, 1e3, 5e5, 5e6]) * rng.normal(size=3)
    if kind == "slab":
        p = np.column_stack([rng.random(n) * 800, rng.random(n) * 400, rng.normal(0, rng.choice([0.0, 0.05, 2.0]), n)])
    elif kind == "strip":
        a = rng.random(n) * 4000; b = rng.random(n) * 100
        p = np.column_stack([a, b, 5 * np.sin(a / 37) + rng.normal(0, 0.02, n)])
    elif kind == "clusters":
        m = int(rng.integers(2, 6)); parts = []
        for i in range(m):
            parts.append(rng.random((n // m, 3)) * rng.uniform(1, 50) + rng.normal(size=3) * rng.uniform(100, 50000))
        p = np.concatenate(parts)
    elif kind == "halo":
        p = rng.random((n, 3)) * np.array([300, 300, 40.0]); m = int(n * rng.uniform(0.002, 0.03))
        p[rng.integers(0, n, m)] = (rng.random((m, 3)) - 0.5) * rng.uniform(3000, 40000)
    elif kind == "outliers":
        p = rng.random((n, 3)) * np.array([500, 500, 50.0]); m = int(rng.integers(1, 400))
        p[rng.integers(0, n, m)] = (rng.random((m, 3)) - 0.5) * rng.uniform(2000, 1e6)
    elif kind == "volume":      # a filled box of random aspect
        p = rng.random((n, 3)) * rng.uniform(5, 2000, 3)
    elif kind == "sheet":       # an undulating surface with noise
        xy = rng.random((n, 2)) * rng.uniform(100, 3000, 2)
        p = np.column_stack([xy, rng.uniform(1, 30) * np.sin(xy[:, 0] / rng.uniform(20, 200)) * np.cos(xy[:, 1] / rng.uniform(20, 200)) + rng.normal(0, rng.choice([0.0, 0.02, 0.5]), n)])
    elif kind == "lattice":     # coordinates on a coarse lattice: exact distance ties and coincident points
        p = np.round(rng.random((n, 3)) * np.array([300, 300, rng.choice([0.0, 4.0, 60.0])])) * rng.choice([0.25, 0.5, 1.0])
    elif kind == "dense_core":  # a density contrast of a few hundred inside one box
        p = rng.random((n, 3)) * np.array([800, 800, 80.0]); m = n // 2
        p[:m] = rng.random((m, 3)) * np.array([60, 60, 20.0]) + np.array([300, 500, 30.0])
    else:
        t = rng.random(n) * 3000
        p = np.column_stack([t, rng.normal(0, 0.3, n), rng.normal(0, 0.3, n)])
    p = p[rng.permutation(len(p))]
    return kind, p @ R.T + off


def cases(seed, count, big=False, more_kinds=False):
    """(index, kind, points, k) of the first `count` cases of a seed -- a fixed sequence (PCG64), so a failing case can be named.
    (big / more_kinds change the sequence: tests name cases of the default one.)"""
    rng = np.random.default_rng(seed)
    for c in range(count):
        kind, pts = cloud(rng, big, more_kinds)
        k = int(rng.choice([5, 8, 12, 16, 16, 16, 24, 30]))
        if os.environ.get("FUZZ_KS"):  # (other k: does not change the sequence of clouds)
            ks = [int(v) for v in os.environ["FUZZ_KS"].split(",")]
            k = ks[c % len(ks)]
        yield c, kind, pts, k


def main():
    import tests.conftest as tc
    from pasture_amd._capi import product_api
    from pasture_amd.buffers import HashMapBuffer
    from pasture_amd.layout import PointLayout, attributes as A
    from pasture_amd.algorithms import compute_normals
    hip, orc = product_api(), tc._load_oracle()
    count = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    only = {int(v) for v in sys.argv[3].split(",")} if len(sys.argv) > 3 else None  # evaluate only these cases (the others are generated and skipped)
    bad = 0
    for c, kind, pts, k in cases(seed, count, big=bool(os.environ.get("FUZZ_BIG")), more_kinds=bool(os.environ.get("FUZZ_KINDS"))):
        if only is not None and c not in only:
            continue
        n = len(pts)
        out = []
        for api in (hip, orc):
            b = HashMapBuffer.new_from_layout(PointLayout.from_attributes([A.POSITION_3D], api=api)); b.resize(n)
            b.set_attribute_range(A.POSITION_3D, range(0, n), pts)
            t0 = time.time(); out.append(compute_normals(b, k, return_knn=True) + (time.time() - t0,))
        (hn, hc, hk, th), (on, oc, ok, to) = out
        diff = (hk != ok).any(axis=1)
        real = 0
        if diff.any():
            # equal distances may be listed in either order: compare the distances of the differing lists
            d_h = ((pts[hk[diff]] - pts[diff.nonzero()[0], None, :]) ** 2).sum(-1); d_o = ((pts[ok[diff]] - pts[diff.nonzero()[0], None, :]) ** 2).sum(-1)
            real = int((d_h != d_o).any(axis=1).sum())
        bad += int(real > 0)
        if real and only is not None:
            qs = diff.nonzero()[0][(d_h != d_o).any(axis=1)]
            for q in qs[:6]:
                dh = np.sqrt(((pts[hk[q]] - pts[q]) ** 2).sum(-1)); do = np.sqrt(((pts[ok[q]] - pts[q]) ** 2).sum(-1))
                print(f"  query {q} at {pts[q].tolist()}\n    hip    idx {hk[q].tolist()}\n           d   {np.round(dh, 4).tolist()}\n    oracle idx {ok[q].tolist()}\n           d   {np.round(do, 4).tolist()}", flush=True)
        print(f"case {c:3d} {kind:9s} n={n:6d} k={k:2d}: lists differing {int(diff.sum()):5d} (not by ties: {real})  hip {th*1e3:7.1f} ms  oracle {to:5.1f} s", flush=True)
    print("FAILED" if bad else "all identical", flush=True)


if __name__ == "__main__":
    main()
