#!/bin/bash
cd /root/repo
hipcc -O2 --offload-arch=gfx950 -Ipasture_amd/csrc tools/test_radix_sort.hip pasture_amd/csrc/radix_sort.hip -o /tmp/test_radix && /tmp/test_radix | tail -4
python -m pytest tests -x -q -m gpu -k "voxel" 2>&1 | tail -2
python bench.py --no-cpu-baseline --workload voxelgrid_xyz --steps 10 --warmup 3 2>&1 | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d['ms_per_step'])"
python bench.py --no-cpu-baseline --workload normals_knn16 --steps 6 --warmup 2 2>&1 | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d['ms_per_step'])"
