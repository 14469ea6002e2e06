cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
B="--workload normals_knn16_sheet --steps 3 --warmup 1 --no-cpu-baseline --no-north-star --no-extra-legs"
rm -rf /tmp/kt; timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/kt -o b -- python bench.py $B > /dev/null 2>&1
python - <<P
import sqlite3,glob
db=glob.glob('/tmp/kt/**/*.db',recursive=True)[0]
cur=sqlite3.connect(db).cursor()
for r in cur.execute("select name,total_calls,average from top_kernels order by total_duration desc limit 4"): print("   ",r[0][:80],r[1],round(r[2],1),"us")
P
rm -rf /tmp/kt; timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d /tmp/kt -o b -- python bench.py $B > /dev/null 2>&1
python - <<P
import sqlite3,glob
db=glob.glob('/tmp/kt/**/*.db',recursive=True)[0]
cur=sqlite3.connect(db).cursor()
for r in cur.execute("select kernel_name, avg(value) from counters_collection where counter_name='WRITE_SIZE' and kernel_name like '%tile2%' group by kernel_name"): print("   WRITE KiB", r[0][:70], round(r[1]))
P
