#!/bin/bash
# Runs on the GPU box: builds tools/pmc_calibrate.hip, collects FETCH_SIZE and WRITE_SIZE in separate --pmc passes (plus kernel times) and
# writes gpurun_out/r02/pmc_calibration.json: counter bytes per access for every access pattern.
set -u
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
out=gpurun_out/r02
mkdir -p $out /tmp/pmccal
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 tools/pmc_calibrate.hip -o /tmp/pmccal/cal || exit 1
for pass in fetch write; do
  c=FETCH_SIZE; [ $pass = write ] && c=WRITE_SIZE
  rm -rf /tmp/pmccal/$pass
  rocprofv3 --kernel-trace --pmc $c -d /tmp/pmccal/$pass -o cal -- /tmp/pmccal/cal > /tmp/pmccal/$pass.log 2>&1
done
python3 - <<'PY'
import glob, json, sqlite3
N = 50_000_000
requested = {"stream_read16": 16, "stream_read8": 8, "stream_read4": 4, "gather_read<8>": 8, "gather_read<24>": 24, "gather_read<32>": 32, "gather_line<4>": 64, "gather_line<8>": 128,
             "stream_write16": 16, "scatter_write<8>": 8, "scatter_write<12>": 12, "scatter_write<32>": 32}
res = {}
for pass_, counter in (("fetch", "FETCH_SIZE"), ("write", "WRITE_SIZE")):
    db = glob.glob(f"/tmp/pmccal/{pass_}/**/*_results.db", recursive=True)[0]
    cur = sqlite3.connect(db).cursor()
    for name, val, cnt in cur.execute("select kernel_name, avg(value), count(*) from counters_collection where counter_name=? group by kernel_name", (counter,)):
        key = next((k for k in requested if k.replace("<", "").replace(">", "") in name.replace("<", "").replace(">", "").replace("void ", "")), None)
        if key is None:
            continue
        res.setdefault(key, {"requested_bytes_per_access": requested[key]})[f"{counter}_bytes_per_access"] = round(val * 1024 / N, 3)
    for name, avg in cur.execute("select name, average from top_kernels"):
        key = next((k for k in requested if k.replace("<", "").replace(">", "") in name.replace("<", "").replace(">", "").replace("void ", "")), None)
        if key is not None and pass_ == "fetch":
            res[key]["avg_ms"] = round(avg / 1e3, 4)  # top_kernels.average is in microseconds
            res[key]["requested_GBps"] = round(requested[key] * N / (avg * 1e-6) / 1e9, 1)
            res[key]["accesses_per_us"] = round(N / avg, 1)
json.dump({"accesses_per_launch": N, "region_bytes": 4 << 30, "unit": "counter value (KiB) x 1024 / accesses", "patterns": res},
          open("gpurun_out/r02/pmc_calibration.json", "w"), indent=1)
print(json.dumps(res, indent=1))
PY
