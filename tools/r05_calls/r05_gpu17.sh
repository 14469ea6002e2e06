cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r05
echo "== expressions"; timeout 600 python -m pytest tests/test_expressions.py -m gpu -q --tb=short -k "gxx_twin and 11" 2>&1 | grep -vE "amdgpu.ids" | tail -60 | cut -c1-2000
echo "== new tests"; timeout 900 python -m pytest tests/test_las_encode.py tests/test_gpu_parity.py -m gpu -q --tb=short -k "reciprocal or cross_lane" -s 2>&1 | grep -vE "amdgpu.ids" | tail -30 | cut -c1-1500
echo "== abab"
python tools/abab.py --workload las0_encode --steps 20 --a "PST_LAS_EXACT_DIV=1" --b "" --out gpurun_out/r05/abab_div.txt | tail -4
python tools/abab.py --workload normals_knn16 --pairs 6 --steps 5 --a "PASTURE_AMD_LIB=$PWD/pasture_amd/libpasture_amd_nofastfit.so" --b "" --out gpurun_out/r05/abab_fastfit.txt | tail -4
python tools/abab.py --workload normals_knn16 --pairs 4 --steps 5 --a "" --b "PST_KNN_FIT=rows" --out gpurun_out/r05/abab_rows.txt | tail -4
echo "== lds pmc"
i=0
for set in "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_INSTS_VALU" "SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_INSTS_SALU SQ_INSTS_VMEM"; do
  out=/tmp/pmc_knn_$i; rm -rf $out; mkdir -p $out
  timeout 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $out -o p -- python bench.py --no-cpu-baseline --no-north-star --no-extra-legs --workload normals_knn16 --steps 1 --warmup 0 > $out/log.txt 2>&1
  python - $out <<'PY' | tee -a gpurun_out/r05/knn_lds_pmc.txt
import csv, glob, collections, sys
f = glob.glob(sys.argv[1] + "/**/p_counter_collection.csv", recursive=True)
acc = collections.defaultdict(float); n = collections.defaultdict(int)
for path in f:
    for r in csv.DictReader(open(path)):
        if "knn_tile2_kernel" in r["Kernel_Name"]:
            acc[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]] += 1
for k in acc: print(f"{k:32s} {acc[k]/max(n[k],1):18.0f}  (launches {n[k]})")
if not acc: print("no counters", open(sys.argv[1] + "/log.txt").read()[-600:])
PY
  i=$((i+1))
done
echo "== phases"
bash tools/r05_knn_phases.sh normals_knn16 2>&1 | tail -12
echo "== full suite"
timeout 1500 python -m pytest tests -m gpu -q --tb=line 2>&1 | grep -vE "amdgpu.ids" | tail -15 | cut -c1-600
cp gpurun_out/curvature_floor_use.json gpurun_out/r05/r05_curvature_floor_use.json 2>/dev/null
