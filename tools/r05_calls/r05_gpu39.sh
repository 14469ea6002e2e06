cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r05
timeout 150 python -m pytest tests/test_gpu_parity.py tests/test_algorithms.py -m gpu -q -x --tb=short -k "structured_volume or one_pass or quantised or cross_lane or normals or knn" 2>&1 | grep -v amdgpu.ids | tail -4 | cut -c1-300
python bench.py --steps 20 --warmup 5 2>/dev/null | tail -1 > gpurun_out/r05/r05_bench_line_final2.json; python -c "
import json; d=json.load(open('gpurun_out/r05/r05_bench_line_final2.json')); print(d['roofline']['frac'], d['configs4_knn16']['ms_per_call'], d['configs4_knn16']['ms_per_call_planned'], d['configs4_knn16']['verified'], d['configs2_las0_to_columns']['frac'], d['north_star_1e9']['frac'], d.get('verified'))"
