cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05
timeout 900 tools/bin/ts6 shapes > gpurun_out/r05/ts6_shapes.txt 2>&1
echo "ts6 rc=$?"
grep -E "^check|fault" gpurun_out/r05/ts6_shapes.txt
for m in 7 4 3; do for n in 100000000 1000000000; do echo "== mode $m n $n best"; grep "mode=$m n= *$n " gpurun_out/r05/ts6_shapes.txt | awk '{print $(NF-1), $0}' | sort -rn | head -8 | cut -d' ' -f2-;  grep "mode=$m n= *$n r4" gpurun_out/r05/ts6_shapes.txt; done; done
