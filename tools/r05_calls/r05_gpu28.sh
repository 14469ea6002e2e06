cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r05
B="--no-cpu-baseline --no-north-star --no-extra-legs"
for rep in 1 2; do
for f in 48 8 16 24 32 40 56 48; do
  echo -n "flush_at=$f  "; PST_KNN_FLUSH_AT=$f python bench.py --workload normals_knn16 --steps 5 --warmup 2 $B 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['roofline']['kernel_ms_avg'])"
done
done | tee gpurun_out/r05/flush_at_sweep.txt
