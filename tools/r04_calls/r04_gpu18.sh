cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04
SPECS_FILE=tools/r04_specs_final2.txt bash tools/run_profiles_r04.sh 2>&1 | grep -v "simple_timer" | tail -8
bash tools/r04_lines.sh box3 > gpurun_out/r04/lines_box3.txt 2>&1
tail -62 gpurun_out/r04/lines_box3.txt
python bench.py --steps 20 --warmup 5 2>/dev/null | tail -1 > gpurun_out/r04/r04_bench_line.json
cat gpurun_out/r04/r04_bench_line.json | cut -c1-300
