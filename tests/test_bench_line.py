"""The ONE JSON line of bench.py on a GPU box, at a reduced size: the headline with `roofline`, `cpu_baseline` and `verified`, and the legs the
round-4 review asked to make driver-visible -- configs[2] (typed LAS-0 records -> 10 columns) and configs[4] (kNN(16) normals) with their in-run
oracle spot checks, and the north-star leg."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_bench_line_carries_every_leg_and_verifies_them():
    n = 3_000_000
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--points", str(n), "--steps", "4", "--warmup", "1", "--north-star-points", str(2 * n),
                        "--cpu-sample-points", str(n)], capture_output=True, text=True, env=env, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert line["n_gpus"] == 1 and line["unit"] == "Mpoints/s" and line["dtype"] == "f64" and line["vs_baseline"] is None
    assert line["verified"] is True  # GPU AABB of the converted points == the oracle's over the same synthetic points
    rf = line["roofline"]
    assert rf["bound"] == "hbm" and rf["peak"] == 8000.0 and 0 < rf["frac"] < 1 and rf["algorithmic_bytes_per_point"] == 48
    cb = line["cpu_baseline"]
    assert cb["kind"] == "port" and cb["cores"] == 1 and cb["value"] > 0
    c2 = line["configs2_las0_to_columns"]
    assert c2["verified"] is True and c2["points"] == n and c2["algorithmic_bytes_per_point"] == 70 and 0 < c2["frac"] < 1 and c2["plan"]
    c4 = line["configs4_knn16"]
    assert c4["verified"] is True and c4["k"] == 16 and c4["ms_per_call"] > 0
    assert c4["checks"] == {"columns_equal_f64_results_narrowed": True, "neighbour_lists_checked": 48, "neighbour_lists_exact": 48}
    sc = cb["spot_checks"]
    assert sc["configs2"]["columns_compared"] == 10 and sc["configs2"]["points_compared"] == 100_000 and sc["configs4"]["fits_compared"] == 48
    ns = line["north_star_1e9"]
    assert ns["points"] == 2 * n and ns["bounds"]
    # the legs can be switched off, and the line stays whole
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--points", str(n), "--steps", "2", "--warmup", "1", "--no-extra-legs", "--no-north-star",
                        "--no-cpu-baseline"], capture_output=True, text=True, env=env, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert "configs2_las0_to_columns" not in line and "configs4_knn16" not in line and "verified" not in line and line["roofline"]["frac"] > 0
