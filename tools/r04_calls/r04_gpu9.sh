cd $GRAFT_REPO_ROOT
for B in 0 6000 8192 12207 16384 20000 24414 32768 40000 50000 61035 65536 80000 100000; do
  echo "B=$B: $(for N in 1000000000; do echo -n "N=$N "; PST_STREAM_XCD_BLOCK=$B N=$N timeout 600 python tools/exp_placement.py 2>&1 | grep 'library pool' | awk '{print $(NF-1)}' | tr '\n' ' '; done)"
done
