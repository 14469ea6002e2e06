"""Deep differential fuzz as part of the driver-run GPU suite (round-3 review: the deep runs existed only as a sentence in DESIGN.md).

The same fuzz bodies as tests/test_gpu_parity.py and tests/test_jit.py, on seeds BEYOND the ones those modules run (seed >= their x1
range), in chunks so that a few hundred extra cases cost few pytest items.  PST_DEEP_FUZZ (default 3) = how many extra multiples of each
suite's x1 case count run here; 0 skips the module; PST_FUZZ_SCALE still scales the base suites themselves.  Every case compares the HIP
path with the oracle byte for byte (neighbour lists identically, normals / curvature within the windows of _compare_normals)."""
import os

import pytest

import test_expressions as te
import test_gpu_parity as gp
import test_jit as tj

pytestmark = pytest.mark.gpu

DEEP = int(os.environ.get("PST_DEEP_FUZZ", "3"))
CHUNK = 20

# (name, body(hip, oracle, seed), x1 case count of the base suite)
SUITES = [
    ("conversions", lambda hip, oracle, seed: gp.test_random_conversions_vs_oracle(hip, oracle, seed), 400),
    ("filter_append", lambda hip, oracle, seed: gp.test_random_filter_append_vs_oracle(hip, oracle, seed), 60),
    ("las_round_trips", lambda hip, oracle, seed: gp.test_random_las_round_trips_vs_oracle(hip, oracle, seed), 40),
    ("voxelgrid", lambda hip, oracle, seed: gp.test_random_voxelgrid_vs_oracle(hip, oracle, seed), 40),
    ("knn_normals", lambda hip, oracle, seed: gp.test_random_knn_normals_vs_oracle(hip, oracle, seed), 24),
    ("knn_sparse_clouds", lambda hip, oracle, seed: gp.test_random_sparse_clouds_knn_vs_oracle(hip, oracle, seed), 8),
    # round 6: expression mappings inside the plan-specialised kernels, random layouts (seeds far from test_expressions' own)
    ("expression_mappings", lambda hip, oracle, seed: te.random_expression_conversion(hip, 1000 + seed), 8),
    ("expression_predicates", lambda hip, oracle, seed: te.random_predicate_filter(hip, 1000 + seed), 8),
]


def _chunks():
    out = []
    for name, _, base in SUITES:
        first, last = base * gp.FUZZ, base * gp.FUZZ + base * DEEP
        for lo in range(first, last, CHUNK):
            out.append(pytest.param(name, lo, min(last, lo + CHUNK), id=f"{name}-{lo}"))
    return out


@pytest.mark.parametrize("suite,lo,hi", _chunks())
def test_deep_fuzz(hip, oracle, suite, lo, hi):
    body = {n: b for n, b, _ in SUITES}[suite]
    for seed in range(lo, hi):
        try:
            body(hip, oracle, seed)
        except AssertionError as e:
            raise AssertionError(f"deep fuzz {suite}, seed {seed}: {e}") from e


def _jit_chunks():
    first, last = 90 * tj.FUZZ, 90 * tj.FUZZ + 90 * DEEP
    return [pytest.param(lo, min(last, lo + CHUNK), id=f"specialised-{lo}") for lo in range(first, last, CHUNK)]


@pytest.mark.parametrize("lo,hi", _jit_chunks())
def test_deep_fuzz_specialised_conversions(hip, oracle, jit_sync, lo, hi):
    """The plan-specialised (hipRTC) kernels on seeds beyond test_jit's own: byte-identical to the oracle."""
    for seed in range(lo, hi):
        try:
            tj.test_specialised_conversions_vs_oracle(hip, oracle, jit_sync, seed)
        except AssertionError as e:
            raise AssertionError(f"deep fuzz specialised conversions, seed {seed}: {e}") from e


def _compaction_chunks():
    first, last = 16 * tj.FUZZ, 16 * tj.FUZZ + 16 * DEEP
    return [pytest.param(lo, min(last, lo + CHUNK), id=f"compaction-{lo}") for lo in range(first, last, CHUNK)]


@pytest.mark.parametrize("lo,hi", _compaction_chunks())
def test_deep_fuzz_specialised_compaction(hip, oracle, jit_sync, lo, hi):
    """The run-time compiled streaming compaction kernels (filter_stream.hpp) on seeds beyond test_jit's own: byte-identical to the oracle."""
    for seed in range(lo, hi):
        try:
            tj.test_specialised_compaction_vs_oracle(hip, oracle, jit_sync, seed)
        except AssertionError as e:
            raise AssertionError(f"deep fuzz specialised compaction, seed {seed}: {e}") from e


from test_jit import jit_sync  # noqa: E402,F401  (the fixture: PST_JIT=sync for the duration of a test)
