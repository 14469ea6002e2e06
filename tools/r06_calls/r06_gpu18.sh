cd $GRAFT_REPO_ROOT
/opt/rocm/bin/hipcc -O2 --offload-arch=gfx950 -Ipasture_amd/csrc tools/exp_scan_time.hip pasture_amd/csrc/radix_sort.hip -o /tmp/st 2>&1 | grep -v warning | head -3; /tmp/st
