"""The planning code of the kNN call (pasture_amd/csrc/normals_plan.hpp) is host-only and free of HIP: its decisions -- does the cloud fill
its box, which box, which frame, how many fine cells, which kernel instance, all points or another level -- are pure functions of measured
statistics.  tests/cpp/test_knn_plan.cpp feeds it the statistics an MI355X measured for the cloud kinds of the differential fuzz and asserts
the chosen path; this test builds and runs it (g++, no GPU)."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_knn_plan_decisions(tmp_path):
    gxx = shutil.which("g++")
    if not gxx:
        pytest.skip("no g++")
    exe = str(tmp_path / "test_knn_plan")
    subprocess.check_call([gxx, "-std=c++17", "-O1", "-Wall", "-Werror", "-I", os.path.join(ROOT, "pasture_amd", "csrc"),
                           os.path.join(ROOT, "tests", "cpp", "test_knn_plan.cpp"), "-o", exe])
    env = {k: v for k, v in os.environ.items() if not k.startswith("PST_")}
    r = subprocess.run([exe], capture_output=True, text=True, env=env)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "all checks passed" in r.stdout


def test_knn_tuning_is_read_from_the_environment_once(tmp_path):
    """KnnTuning::from_env: every switch lands in the struct (a second program reads a full set of variables)."""
    gxx = shutil.which("g++")
    if not gxx:
        pytest.skip("no g++")
    src = tmp_path / "t.cpp"
    src.write_text('#include "normals_plan.hpp"\n#include <cstdio>\nint main() { const pstk::KnnTuning& t = pstk::knn_tuning(); '
                   'std::printf("%g %g %g %d %ld %d%d%d%d%d%d%d %d %d%d%d %d %u,%u,%u %u %lld\\n", t.cell, t.per_cell, t.tau_m, t.rx, t.cell_budget, t.debug, t.trace, '
                   't.no_scale, t.no_trim, t.no_rotate, t.no_tile, t.force_tile, t.dense, t.direct_out, t.box_list, t.rounds, (int)t.variant, t.tile[0], t.tile[1], t.tile[2], '
                   't.flush_at, t.scratch_max); return 0; }\n')
    exe = str(tmp_path / "t")
    subprocess.check_call([gxx, "-std=c++17", "-I", os.path.join(ROOT, "pasture_amd", "csrc"), str(src), "-o", exe])
    env = {k: v for k, v in os.environ.items() if not k.startswith("PST_")}
    assert subprocess.run([exe], capture_output=True, text=True, env=env).stdout.split() == ["0", "0", "0", "0", "20", "0000000", "-1", "111", "0", "0,0,0", "48", str(16 << 30)]
    env.update({"PST_KNN_CELL": "2.5", "PST_KNN_PER_CELL": "3", "PST_KNN_TAU_M": "30", "PST_KNN_RX": "2", "PST_KNN_CELL_BUDGET": "7", "PST_KNN_DEBUG": "1",
                "PST_KNN_NO_TRIM": "1", "PST_KNN_FORCE_TILE": "1", "PST_KNN_DENSE": "0", "PST_KNN_DIRECT": "0", "PST_KNN_ROUNDS": "0", "PST_KNN_VAR": "G",
                "PST_KNN_TILE": "12,4,3", "PST_KNN_FLUSH_AT": "32", "PST_SCRATCH_MAX_BYTES": "1024"})
    out = subprocess.run([exe], capture_output=True, text=True, env=env).stdout.split()
    assert out == ["2.5", "3", "30", "2", "7", "1001001", "0", "010", str(ord("G")), "12,4,3", "32", "1024"], out
