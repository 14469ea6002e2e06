cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r05
echo "== full suite"
timeout 1200 python -m pytest tests -m gpu -q --tb=short 2>&1 | grep -vE "amdgpu.ids" | tail -12 | cut -c1-1200
cp gpurun_out/curvature_floor_use.json gpurun_out/r05/r05_curvature_floor_use.json 2>/dev/null
echo "== smoke"; python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
echo "== bench line"
python bench.py --steps 20 --warmup 5 2>/dev/null | tail -1 > gpurun_out/r05/r05_bench_line_final.json; cut -c1-400 gpurun_out/r05/r05_bench_line_final.json
