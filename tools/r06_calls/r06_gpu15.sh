cd $GRAFT_REPO_ROOT
/opt/rocm/bin/hipcc -O2 --offload-arch=gfx950 -Ipasture_amd/csrc tools/test_radix_sort.hip pasture_amd/csrc/radix_sort.hip -o /tmp/test_radix 2>&1 | grep -v warning | head -5
timeout 1200 /tmp/test_radix | grep -v "^ok n=" | tail -20
timeout 900 python -m pytest tests/test_voxel_grid.py -x -q -m gpu 2>&1 | tail -3
