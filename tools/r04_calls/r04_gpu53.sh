cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04
( time timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed|FAILED|Error" ) 2>&1 | tail -5
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
python bench.py --steps 20 --warmup 5 2>/dev/null | tail -1 > gpurun_out/r04/bench_line_final.json; cut -c1-330 gpurun_out/r04/bench_line_final.json
SPECS_FILE=tools/r04_specs_final3.txt bash tools/run_profiles_r04.sh 2>&1 | grep -v simple_timer | tail -6
rm -f gpurun_out/r04/filter_lines.jsonl
for w in filter_big_columnar filter_big_interleaved filter_las0_columnar filter_las0_interleaved filter_las3_columnar filter_las3_interleaved filter_las8_columnar filter_las8_interleaved filter_las9_interleaved; do
  for plan in interpreted specialised; do
    python bench.py --no-cpu-baseline --no-north-star --workload $w --plan $plan --steps 20 --warmup 5 2>/dev/null | tail -1 >> gpurun_out/r04/filter_lines.jsonl
  done
done
python - <<'PY'
import json
for l in open("gpurun_out/r04/filter_lines.jsonl"):
    d = json.loads(l); print(d["config"]["workload"].split(":")[0], d["ms_per_step"], d["roofline"]["frac"], d["config"].get("plan"))
PY
