cd $GRAFT_REPO_ROOT
# Where do the waves of the kernels that stay below 0.75 of peak spend their cycles?  One SQ pass per workload (counters only, with --kernel-trace).
export TMPDIR=/tmp
mkdir -p gpurun_out/r04
for spec in "las0_encode:las" "columns_to_las0:convert" "filter_big_interleaved:filter" "filter_big_columnar:filter" "convert_affine_bounds:vec3f64_stream" "las0_to_columns:convert"; do
  w=${spec%%:*}; kern=${spec##*:}
  d=gpurun_out/prof_sq/$w; mkdir -p $d
  timeout 600 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVES SQ_BUSY_CYCLES -d $d -o bench -- \
    python bench.py --no-cpu-baseline --no-north-star --workload $w --steps 3 --warmup 1 > $d/bench.log 2>&1
  echo "$w rc=$?"
  python - "$w" "$kern" "$d/bench_results.db" <<'PY'
import json, sqlite3, sys
w, kern, db = sys.argv[1:4]
cur = sqlite3.connect(db).cursor()
rows = list(cur.execute("select kernel_name, counter_name, avg(value), count(*) from counters_collection group by kernel_name, counter_name"))
by = {}
for name, counter, val, cnt in rows:
    short = name.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:90]
    by.setdefault(short, {"launches": cnt})[counter] = round(val)
# the kernels with the most wave cycles first
top = sorted(by.items(), key=lambda kv: -kv[1].get("SQ_WAVE_CYCLES", 0))[:4]
out = {"workload": w, "kernels": {k: v for k, v in top}}
for k, v in top:
    wc = v.get("SQ_WAVE_CYCLES", 0) or 1
    v["frac_wait_any"] = round(v.get("SQ_WAIT_ANY", 0) / wc, 3)
    v["frac_wait_inst_any"] = round(v.get("SQ_WAIT_INST_ANY", 0) / wc, 3)
    v["frac_active_inst_any"] = round(v.get("SQ_ACTIVE_INST_ANY", 0) / wc, 3)
    v["frac_active_inst_valu"] = round(v.get("SQ_ACTIVE_INST_VALU", 0) / wc, 3)
print(json.dumps(out))
open("gpurun_out/r04/sq_cycles.jsonl", "a").write(json.dumps(out) + "\n")
PY
  rm -rf $d
done
