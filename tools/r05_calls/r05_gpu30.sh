cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r05
python tools/abab.py --workload columns_to_las0 --steps 20 --pairs 6 --a "" --b "PST_LAS_DECODE=0" --flags-a "--plan specialised" --flags-b "--plan specialised" --out gpurun_out/r05/abab_c2r_family.txt | tail -7
