cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06
T=$(date +%s)
timeout 900 python tools/abab.py --workload columns_to_las0 --a "PST_COLUMN_STAGGER=0" --b "PST_COLUMN_STAGGER=1024" --pairs 8 --steps 20 --out gpurun_out/r06/abab_stagger_c2l_$T.txt 2>&1 | tail -7
timeout 900 python tools/abab.py --workload las0_to_columns --a "PST_COLUMN_STAGGER=0" --b "PST_COLUMN_STAGGER=1024" --pairs 8 --steps 20 --out gpurun_out/r06/abab_stagger_l2c_$T.txt 2>&1 | tail -7
