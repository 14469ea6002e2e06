cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06
( time timeout 900 python bench.py > gpurun_out/r06/bench_line_box3.json 2> gpurun_out/r06/bench_line_box3.err ) 2>&1 | tail -3
python - <<'P'
import json
l=json.loads(open('gpurun_out/r06/bench_line_box3.json').read().strip().splitlines()[-1])
print(json.dumps(l['roofline'])[:900])
print(l['ms_per_step'], l.get('verified'), {k: (v.get('verified'), v.get('ms_per_step', v.get('ms_per_call', v.get('us_per_chunk_synchronous')))) for k, v in l.items() if isinstance(v, dict) and 'verified' in v})
P
