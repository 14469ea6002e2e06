#!/bin/bash
# A second build of the library with the kNN box kernel's counters compiled in (-DPST_KNN_STATS): pasture_amd/libpasture_amd_stats.so.
# Select it with PASTURE_AMD_LIB=$PWD/pasture_amd/libpasture_amd_stats.so (same-box runs beside the product build).
set -e
cd "$(dirname "$0")/../pasture_amd/csrc"
make -j8 > /dev/null
mkdir -p build_stats
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -ffp-contract=off --offload-arch=gfx950 -Wall -Wno-unused-function -DPST_KNN_STATS -c normals_tile.hip -o build_stats/normals_tile.o
objs=$(ls build/*.o | grep -v normals_tile.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libpasture_amd_stats.so $objs build_stats/normals_tile.o -ldl
echo built ../libpasture_amd_stats.so
