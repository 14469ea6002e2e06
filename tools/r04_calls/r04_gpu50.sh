cd $GRAFT_REPO_ROOT
bash tools/ab.sh gpurun_ab/lib_spans6.so gpurun_ab/lib_list.so filter_big_columnar 2>&1 | grep -v amdgpu
bash tools/ab.sh gpurun_ab/lib_spans6.so gpurun_ab/lib_list.so filter_big_columnar 2>&1 | grep -v amdgpu
