cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04
timeout 900 python -m pytest tests -m gpu -x -q -k "filter or static" 2>&1 | tail -3
PST_JIT=sync timeout 900 python -m pytest tests/test_filter_append.py tests/test_gpu_parity.py -m gpu -x -q -k "filter" 2>&1 | tail -3
line() { python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1', d['ms_per_step'], d['roofline']['frac'], d['config'].get('plan'))"; }
for rep in 1 2; do
 for w in filter_las0_columnar filter_las3_columnar filter_big_columnar; do
  for plan in interpreted specialised; do
    python bench.py --no-cpu-baseline --no-north-star --workload $w --plan $plan --steps 20 --warmup 5 2>gpurun_out/r04/err.txt | tail -1 | line "$w $plan" || tail -5 gpurun_out/r04/err.txt
  done
 done
done
