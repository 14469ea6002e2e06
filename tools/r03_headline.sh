#!/bin/bash
# Round 3, north-star evidence (runs on the GPU box through gpurun): the 10^9-point fused convert + AABB under rocprofv3 (kernel trace, then
# separate FETCH_SIZE / WRITE_SIZE passes), summarised on the box; then a same-box A/B of the placement switches at 10^9 points.
set -u
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
out=gpurun_out/r03/headline
mkdir -p $out
cp profiles/hbm_traffic.json $out/hbm_traffic.json
N=1000000000
for pass in kt fetch write; do
  d=gpurun_out/prof1e9/$pass; mkdir -p $d
  case $pass in
    kt) args="--kernel-trace --stats"; steps=10 ;;
    fetch) args="--kernel-trace --pmc FETCH_SIZE"; steps=3 ;;
    write) args="--kernel-trace --pmc WRITE_SIZE"; steps=3 ;;
  esac
  timeout 900 rocprofv3 $args -d $d -o bench -- python bench.py --no-cpu-baseline --no-north-star --points $N --steps $steps --warmup 2 > $d/bench.log 2>&1
  echo "$pass rc=$? $(tail -c 400 $d/bench.log | tr '\n' ' ' | cut -c1-300)"
done
python tools/rocprof_summary.py --round r03 --workload convert_affine_bounds --key convert_affine_bounds_1e9 --points $N --kernel "vec3f64_stream_kernel<true, true, true" \
  --out $out --kt gpurun_out/prof1e9/kt/bench_results.db --fetch gpurun_out/prof1e9/fetch/bench_results.db --write gpurun_out/prof1e9/write/bench_results.db \
  --cmd "python bench.py --no-cpu-baseline --no-north-star --points $N --steps 10 --warmup 2" | head -12
rm -rf gpurun_out/prof1e9
# same-box A/B at 10^9 points: tile numbering, allocator
for rep in 1 2 3; do
  for v in "default" "PST_STREAM_XCD=0" "PST_NO_POOL=1"; do
    e=""; [ "$v" != default ] && e="$v"
    env $e python bench.py --no-cpu-baseline --no-north-star --points $N --steps 10 --warmup 2 2>/dev/null | tail -1 | \
      python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('AB1e9 $v', r['frac'], r['kernel_ms_avg'], r['kernel_ms_min'])"
  done
done | tee $out/ab_1e9.txt
# the default driver line (10^8 timed region + the 10^9 leg + CPU baseline)
python bench.py 2>/dev/null | tail -1 > $out/bench_default.json
python -c "import json; d=json.load(open('$out/bench_default.json')); print(d['value'], d['roofline']['frac'], d.get('north_star_1e9'))"
# kNN reference lines of this box (before the round's kernel work)
for w in normals_knn16 normals_knn16_sheet voxelgrid_xyz filter_big_columnar filter_big_interleaved; do
  python bench.py --workload $w --steps 6 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 > $out/base_$w.json
  python -c "import json; d=json.load(open('$out/base_$w.json')); print('$w', d['ms_per_step'], d['roofline']['frac'])"
done
