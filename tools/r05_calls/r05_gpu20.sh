cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r05
B="--no-cpu-baseline --no-north-star --no-extra-legs"
echo "== product lib, 8e6, steps 1 warmup 0"; timeout 300 python bench.py --workload normals_knn16 --points 8000000 --steps 1 --warmup 0 $B 2>&1 | grep -v amdgpu.ids | tail -2 | cut -c1-300
echo "== stats lib, 8e6, steps 3 warmup 1"; PASTURE_AMD_LIB=$PWD/pasture_amd/libpasture_amd_stats.so timeout 300 python bench.py --workload normals_knn16 --points 8000000 --steps 3 --warmup 1 $B 2>&1 | grep -v amdgpu.ids | grep -a "pst knn\|fault\|Error\|error" | tail -8 | cut -c1-700
echo "== stats lib, 8e6, PST_KNN_DEBUG"; PST_KNN_DEBUG=1 PASTURE_AMD_LIB=$PWD/pasture_amd/libpasture_amd_stats.so timeout 300 python bench.py --workload normals_knn16 --points 8000000 --steps 1 --warmup 0 $B 2>&1 | grep -v amdgpu.ids | tail -12 | cut -c1-400
echo "== lds pmc of the record-side kernels"
for w in las0_encode filter_las0_columnar columns_to_las0; do
 i=0
 for set in "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_LDS" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM" "SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_ACTIVE_INST_VMEM SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAVES"; do
  out=/tmp/pmc_${w}_$i; rm -rf $out; mkdir -p $out
  timeout 200 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $out -o p -- python bench.py $B --workload $w --steps 2 --warmup 1 > $out/log.txt 2>&1
  python - $out $w <<'PY' | tee -a gpurun_out/r05/record_side_pmc.txt
import csv, glob, collections, sys
f = glob.glob(sys.argv[1] + "/**/p_counter_collection.csv", recursive=True)
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(lambda: collections.defaultdict(int))
for path in f:
    for r in csv.DictReader(open(path)):
        k = r["Kernel_Name"]
        if any(s in k for s in ("las_encode_kernel", "filter_stream", "filter_las", "las_columns_to_records", "filter_big")) and "fold" not in k:
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[k][r["Counter_Name"]] += 1
for k in acc:
    print(sys.argv[2], k[:90])
    for c in acc[k]: print(f"    {c:28s} {acc[k][c]/max(n[k][c],1):16.0f}  (launches {n[k][c]})")
if not acc: print("no counters", open(sys.argv[1] + "/log.txt").read()[-400:])
PY
  i=$((i+1))
 done
done
