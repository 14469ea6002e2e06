cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r02/prof_summ2
cp profiles/hbm_traffic.json gpurun_out/r02/prof_summ2/hbm_traffic.json
tools/profile_round.sh convert_affine_bounds > /dev/null 2>&1
python tools/rocprof_summary.py --round r02 --workload convert_affine_bounds --kernel "vec3f64_stream_kernel<true, true, true" --out gpurun_out/r02/prof_summ2 --kt gpurun_out/prof/convert_affine_bounds/kt/bench_results.db --fetch gpurun_out/prof/convert_affine_bounds/fetch/bench_results.db --write gpurun_out/prof/convert_affine_bounds/write/bench_results.db --cmd "python bench.py --no-cpu-baseline --no-north-star --workload convert_affine_bounds" | head -8
rm -rf gpurun_out/prof
