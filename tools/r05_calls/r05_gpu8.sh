cd $GRAFT_REPO_ROOT
bash tools/exp_resident.sh "narrow_f64_f32" "0 8 6 4 3 2 1" gpurun_out/r05/resident_sweep_columns.jsonl
python bench.py > gpurun_out/r05/bench_c.json 2> gpurun_out/r05/bench_c.err; echo "bench rc=$?"
python - <<'PY'
import json
d=json.load(open('gpurun_out/r05/bench_c.json'))
for k in ('value','ms_per_step','verified'): print(k, d.get(k))
print('roofline', d['roofline']['frac'], d['roofline']['kernel_ms_avg'])
for k in ('configs2_las0_to_columns','configs4_knn16','north_star_1e9'):
    v=dict(d.get(k,{})); v.pop('note',None); print(k, json.dumps(v)[:900])
cb=d.get('cpu_baseline',{}); print('cpu', cb.get('value'), cb.get('wall_s'), cb.get('spot_checks_wall_s'), json.dumps(cb.get('spot_checks'))[:1200])
PY
tail -5 gpurun_out/r05/bench_c.err
