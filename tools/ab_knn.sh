#!/bin/bash
# Same-box A/B of library builds on the kNN workload (run on the GPU box): tools/ab_knn.sh <lib.so> ...   ("-" = the in-tree build)
cd "$(dirname "$0")/.."
for lib in "$@"; do
  if [ "$lib" = "-" ]; then unset PASTURE_AMD_LIB; else export PASTURE_AMD_LIB=$lib; fi
  echo "== $lib: $(tools/kt.sh normals_knn16 2>&1 | head -1 | awk '{print $(NF-3), $(NF-2)}')"
done
