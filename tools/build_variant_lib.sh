#!/bin/bash
# A second build of the library with extra compile-time switches on some translation units, for same-box A/Bs beside the product build:
#   tools/build_variant_lib.sh <name> "<flags>" <file.hip> [more files]   ->  pasture_amd/libpasture_amd_<name>.so   (select with PASTURE_AMD_LIB)
set -e
name=$1; flags=$2; shift 2
cd "$(dirname "$0")/../pasture_amd/csrc"
make -j8 > /dev/null
mkdir -p build_$name
skip=""
for f in "$@"; do
  o=build_$name/${f%.*}.o
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -ffp-contract=off --offload-arch=gfx950 -Wall -Wno-unused-function -Ibuild $flags -x hip -c $f -o $o
  skip="$skip|${f%.*}.o"
done
objs=$(ls build/*.o | grep -vE "/(${skip#|})$")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libpasture_amd_$name.so $objs build_$name/*.o -ldl
echo built pasture_amd/libpasture_amd_$name.so
