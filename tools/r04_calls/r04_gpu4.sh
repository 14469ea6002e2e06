cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04
(timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -8) > gpurun_out/r04/gputest.log 2>&1
cat gpurun_out/r04/gputest.log
bash tools/r04_lines.sh box1 > gpurun_out/r04/lines_box1.txt 2>&1
tail -70 gpurun_out/r04/lines_box1.txt
timeout 1500 python tools/exp_jit_layouts.py --seeds 24 --points 100000000 --steps 8 --out gpurun_out/r04/r04_random_layouts.jsonl 2>&1 | tail -3
bash tools/run_profiles_r04.sh 2>&1 | tail -30
python bench.py --steps 20 --warmup 5 2>/dev/null | tail -1 > gpurun_out/r04/r04a_bench_line.json
cat gpurun_out/r04/r04a_bench_line.json
