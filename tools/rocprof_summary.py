#!/usr/bin/env python3
"""Summarise rocprofv3 (ROCm 7.2, rocpd sqlite output) runs of bench.py into the files kept under profiles/.

  python tools/rocprof_summary.py --round r01 --workload convert_affine_bounds --points 100000000 \
      --kt gpurun_out/prof/kt/bench_results.db --fetch gpurun_out/prof/fetch/bench_results.db --write gpurun_out/prof/write/bench_results.db

Counter handling follows /opt/skills/guides/MI355X_MICROARCH.md (HBM / rocprofv3 section): FETCH_SIZE and WRITE_SIZE are
collected in SEPARATE --pmc passes (they do not fit one pass), values are KiB, and on gfx950 FETCH_SIZE reports exactly
half of the bytes of a wide coalesced streaming read => doubled here.  WRITE_SIZE is taken as is (calibrated in the same
run: rocclr's fillBufferAligned of exactly 2.4e9 B reads 2,343,750 KiB).
"""
import argparse
import json
import os
import sqlite3


def top_kernels(db):
    cur = sqlite3.connect(db).cursor()
    return list(cur.execute("select name, total_calls, total_duration, average, percentage from top_kernels"))


def counter_avg(db, counter):
    cur = sqlite3.connect(db).cursor()
    q = "select kernel_name, count(*), avg(value), min(value), max(value) from counters_collection where counter_name=? group by kernel_name"
    return {r[0]: r[1:] for r in cur.execute(q, (counter,))}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--round", required=True)
    ap.add_argument("--workload", default="convert_affine_bounds")
    ap.add_argument("--points", type=int, default=100_000_000)
    ap.add_argument("--kernel", default="vec3f64_stream_kernel")
    ap.add_argument("--kt")
    ap.add_argument("--fetch")
    ap.add_argument("--write")
    ap.add_argument("--cmd", default="python bench.py --no-cpu-baseline")
    ap.add_argument("--out", default="profiles")
    a = ap.parse_args()
    os.makedirs(a.out, exist_ok=True)
    lines = []
    if a.kt:
        lines.append(f"# rocprofv3 --kernel-trace --stats -- {a.cmd}   (round {a.round}, workload {a.workload}, {a.points} points)")
        lines.append(f"{'kernel':110s} {'calls':>6s} {'total_us':>12s} {'avg_us':>10s} {'pct':>7s}")
        for name, calls, total, avg, pct in top_kernels(a.kt):
            lines.append(f"{name[:110]:110s} {calls:6d} {total:12.1f} {avg:10.2f} {pct:7.2f}")
    traffic = {}
    fetch = counter_avg(a.fetch, "FETCH_SIZE") if a.fetch else {}
    write = counter_avg(a.write, "WRITE_SIZE") if a.write else {}
    if fetch or write:
        lines.append("")
        lines.append("# PMC passes (separate runs): values in KiB per dispatch; FETCH_SIZE x2 correction on gfx950")
        lines.append(f"{'kernel':110s} {'n':>4s} {'FETCH_KiB':>14s} {'WRITE_KiB':>14s} {'HBM_bytes/launch':>18s}")
        for k in sorted(set(fetch) | set(write)):
            f = fetch.get(k, (0, 0.0, 0, 0))
            w = write.get(k, (0, 0.0, 0, 0))
            total = 2.0 * f[1] * 1024 + w[1] * 1024
            lines.append(f"{k[:110]:110s} {max(f[0], w[0]):4d} {f[1]:14.1f} {w[1]:14.1f} {total:18.0f}")
            if a.kernel in k:
                traffic = {"points": a.points, "kernel": k, "fetch_kib_raw": f[1], "write_kib": w[1],
                           "bytes_per_launch": round(total), "correction": "FETCH_SIZE x2 (gfx950 wide coalesced reads), KiB x1024",
                           "round": a.round}
    txt = os.path.join(a.out, f"{a.round}_{a.workload}_rocprof.txt")
    with open(txt, "w") as fh:
        fh.write("\n".join(lines) + "\n")
    print("\n".join(lines))
    if traffic:
        tpath = os.path.join(a.out, "hbm_traffic.json")
        allt = json.load(open(tpath)) if os.path.exists(tpath) else {}
        allt[a.workload] = traffic
        json.dump(allt, open(tpath, "w"), indent=1)


if __name__ == "__main__":
    main()
