cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05
O=gpurun_out/r05/abab_1.txt
python tools/abab.py --workload convert_affine_bounds --a "PST_STREAM_RESIDENT=8" --b "" --out $O
python tools/abab.py --workload convert_affine_bounds --points 1000000000 --steps 8 --a "PST_STREAM_RESIDENT=8" --b "" --out $O
python tools/abab.py --workload narrow_f64_f32 --a "PST_RESIDENT=0" --b "PST_RESIDENT=4" --out $O
python tools/abab.py --workload rawlas_to_records --a "PST_RESIDENT=0" --b "PST_RESIDENT=6" --out $O
python tools/abab.py --workload benchlayout_records_to_columns --a "PST_RESIDENT=0" --b "PST_RESIDENT=8" --out $O
