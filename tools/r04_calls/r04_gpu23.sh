cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_jit.py tests/test_las_golden.py tests/test_las_encode.py -m gpu -x -q 2>&1 | grep -E "passed|failed|FAILED" | tail -3
timeout 1200 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "las or LAS or raw" 2>&1 | grep -E "passed|failed|FAILED" | tail -3
python - <<'PY'
import ctypes, torch, sys, os
sys.path.insert(0, os.getcwd())
import pasture_amd as pa
from pasture_amd import las, conversion as cv
api = pa.product_api(); s = torch.cuda.current_stream(); api.set_stream(ctypes.c_void_p(s.cuda_stream))
n = 50_000_000
for f in (1, 3, 6, 7):
    raw = las.point_layout_from_las_point_format(las.Format(f), True); typed = las.point_layout_from_las_point_format(las.Format(f), False)
    src = pa.VectorBuffer.new_from_layout(raw); src.resize(n); src.synth_fill(1, 0)
    dst = pa.VectorBuffer.new_from_layout(typed); dst.resize(n)
    conv = las.get_default_las_converter(raw, typed, (0.001,)*3, (0.0,)*3)
    for _ in range(2): conv.convert_into_range_async(src, range(0, n), dst, range(0, n))
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(s)
    for _ in range(5): conv.convert_into_range_async(src, range(0, n), dst, range(0, n))
    e1.record(s); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 5
    b = raw.size_of_point_entry() + typed.size_of_point_entry()
    print(f"raw LAS-{f} -> typed records: {cv.last_plan_kinds(api)} {b * n / ms / 1e9 / 8000:.3f} of peak")
PY
