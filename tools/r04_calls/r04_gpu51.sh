cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04
timeout 900 python -m pytest tests/test_jit.py tests/test_filter_append.py tests/test_deep_fuzz.py -m gpu -x -q -k "filter or padding or tiny or compaction" 2>&1 | tail -2
PST_FUZZ_SCALE=10 timeout 900 python -m pytest tests/test_jit.py -m gpu -x -q -k "specialised_compaction_vs" 2>&1 | tail -2
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
