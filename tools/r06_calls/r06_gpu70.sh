cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06
# the driver's own command on the final tree
SECONDS=0; python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r06/bench_driver_cmd.json 2> gpurun_out/r06/bench_driver_cmd.err; echo "rc=$?"
echo "wall ${SECONDS}s"
python3 -c "
import json; d=json.loads(open('gpurun_out/r06/bench_driver_cmd.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['traffic'], d['north_star_1e9']['frac'], d['configs2_las0_to_columns']['frac'], d['configs4_knn16']['ms_per_call'], d['cpu_baseline']['value'], d['verified'])"
python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
