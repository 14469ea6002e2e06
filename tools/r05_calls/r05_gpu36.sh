cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r05
timeout 1000 python -m pytest tests -m gpu -q --tb=short > /tmp/suite.log 2>&1; echo "rc=$?"
grep -v amdgpu.ids /tmp/suite.log | tail -8 | cut -c1-600
cp gpurun_out/curvature_floor_use.json gpurun_out/r05/r05_curvature_floor_use.json 2>/dev/null
