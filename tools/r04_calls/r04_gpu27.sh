cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -m gpu -x -q -k "filter or static" 2>&1 | tail -2
SPECS_FILE=tools/r04_specs_final3.txt bash tools/run_profiles_r04.sh 2>&1 | tail -12
for w in filter_big_columnar filter_big_interleaved; do
  python bench.py --no-cpu-baseline --no-north-star --workload $w --plan specialised --steps 20 --warmup 5 2>/dev/null | tail -1 >> gpurun_out/r04/filter_lines.jsonl
done
cut -c1-300 gpurun_out/r04/filter_lines.jsonl
