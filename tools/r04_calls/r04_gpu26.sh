cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04
timeout 900 python bench.py --gpus 8 --rehearse-on-one-gpu --steps 5 --warmup 2 --no-cpu-baseline --no-north-star --points 12500000 --configs3-points 1000000000 2>gpurun_out/r04/rehearsal_8.err | tail -1 > gpurun_out/r04/r04_rehearsal_8ranks_one_gpu.json
tail -3 gpurun_out/r04/rehearsal_8.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/r04/r04_rehearsal_8ranks_one_gpu.json'))
print(d['n_gpus'], d['config'].get('rehearsal'), d['self_check'], d['per_rank'])
print(d['configs3_1e9']['self_check'], d['configs3_1e9']['points_rank0'], d['configs3_1e9']['bounds'])
PY
