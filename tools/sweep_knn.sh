#!/bin/bash
# Grid-parameter sweep of the kNN search (run on the GPU box): M = expected points in the a-priori ball, rx = fine x cells per h.
cd "$(dirname "$0")/.."
for m in ${MS:-24 28 32}; do for rx in ${RXS:-4 6 8}; do
  r=$(PST_KNN_TAU_M=$m PST_KNN_RX=$rx python bench.py --workload normals_knn16 --steps 4 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; print(json.loads(sys.stdin.read())['ms_per_step'])")
  echo "M=$m rx=$rx ms=$r"
done; done
