cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06
for w in filter_las1_columnar filter_las2_columnar filter_las0_columnar; do
timeout 900 python tools/abab.py --workload $w --a "X=1" --b "PASTURE_AMD_LIB=$PWD/pasture_amd/libpasture_amd_b1.so" --pairs 5 --steps 20 --out gpurun_out/r06/abab_filter_b1_$w.txt 2>&1 | tail -3
done
