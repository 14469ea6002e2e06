cd $GRAFT_REPO_ROOT
( time PST_JIT=sync timeout 3000 python -m pytest tests/test_gpu_parity.py tests/test_buffer_converter.py tests/test_las_golden.py tests/test_las_encode.py tests/test_slices_centroid_views.py tests/test_filter_append.py tests/test_jit.py -m gpu -q -x 2>&1 | grep -E "passed|failed|FAILED|Error" ) 2>&1 | tail -8
python - <<'PY'
import sys, os
sys.path.insert(0, os.getcwd())
import pasture_amd as pa
from pasture_amd import conversion as cv
print("jit stats of this process (none expected):", cv.jit_stats(pa.product_api()))
PY
ls ~/.cache/pasture_amd 2>/dev/null | wc -l
