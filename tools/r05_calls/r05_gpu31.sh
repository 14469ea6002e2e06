cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
echo "== tests"; timeout 900 python -m pytest tests/test_jit.py tests/test_las_golden.py tests/test_buffer_converter.py tests/test_gpu_parity.py -m gpu -q --tb=short -k "family or las or convert or full_size" 2>&1 | grep -v amdgpu.ids | tail -8 | cut -c1-600
for i in 1 2; do python bench.py --workload columns_to_las0 --plan specialised --steps 20 --warmup 3 --no-cpu-baseline --no-north-star --no-extra-legs 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['roofline']['kernel_ms_avg'], d['roofline']['frac'], d['config']['plan'], d['config'].get('family_measured'))"; done
