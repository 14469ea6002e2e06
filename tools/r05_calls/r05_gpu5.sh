cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05
timeout 600 tools/bin/ts6 sizes 3 2 512 > gpurun_out/r05/ts6_sizes_k3_b512x2.txt 2>&1
timeout 600 tools/bin/ts6 spacing 3 2 512 > gpurun_out/r05/ts6_spacing_k3_b512x2.txt 2>&1
grep -E "^sizes" gpurun_out/r05/ts6_sizes_k3_b512x2.txt
grep -E "^spacing" gpurun_out/r05/ts6_spacing_k3_b512x2.txt | awk 'NR%4==1'
