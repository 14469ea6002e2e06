cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06
timeout 1200 python bench.py > gpurun_out/r06/bench_line_final_b.json 2> gpurun_out/r06/bench_line_final_b.err
echo "bench exit $?"
for w in filter_las0_columnar las0_encode; do timeout 600 python bench.py --no-cpu-baseline --no-north-star --no-extra-legs --workload $w --steps 20 2>/dev/null | tail -1 | cut -c1-400; done
