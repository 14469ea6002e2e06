cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06
# E16: the compaction's ragged last tile on a second stream next to the streaming kernel (PST_FILTER_TAIL_SIDE, default on)
timeout 1200 python -m pytest tests/test_filter_append.py tests/test_expressions.py tests/test_gpu_parity.py tests/test_deep_fuzz.py -m gpu -q -p no:cacheprovider -x -k "filter or compaction or append" 2>&1 | tail -3
for w in filter_las0_columnar filter_las0_interleaved filter_big_columnar filter_big_interleaved; do
  timeout 900 python tools/abab.py --workload $w --a "PST_FILTER_TAIL_SIDE=0" --b "PST_FILTER_TAIL_SIDE=1" --pairs 6 --steps 20 --out gpurun_out/r06/abab_tail_side_$w.txt 2>&1 | tail -4
done
