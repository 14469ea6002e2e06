cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04
timeout 900 python -m pytest tests/test_jit.py -m gpu -x -q -k "compaction" 2>&1 | tail -5
SPECS_FILE=tools/r04_specs_final3.txt bash tools/run_profiles_r04.sh 2>&1 | grep -v simple_timer | tail -8
rm -f gpurun_out/r04/filter_lines.jsonl
for w in filter_big_columnar filter_big_interleaved filter_las0_columnar filter_las0_interleaved filter_las3_columnar filter_las3_interleaved; do
  for plan in interpreted specialised; do
    python bench.py --no-cpu-baseline --no-north-star --workload $w --plan $plan --steps 20 --warmup 5 2>/dev/null | tail -1 >> gpurun_out/r04/filter_lines.jsonl
  done
done
python - <<'PY'
import json
for l in open("gpurun_out/r04/filter_lines.jsonl"):
    d = json.loads(l); print(d["config"]["workload"].split(":")[0], d["ms_per_step"], d["roofline"]["frac"], d["config"].get("plan"))
PY
