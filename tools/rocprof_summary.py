#!/usr/bin/env python3
"""Summarise rocprofv3 (ROCm 7.2, rocpd sqlite output) runs of bench.py into the files kept under profiles/.

  python tools/rocprof_summary.py --round r01 --workload convert_affine_bounds --points 100000000 \
      --kt gpurun_out/prof/kt/bench_results.db --fetch gpurun_out/prof/fetch/bench_results.db --write gpurun_out/prof/write/bench_results.db

Counter handling follows /opt/skills/guides/MI355X_MICROARCH.md (HBM / rocprofv3 section): FETCH_SIZE and WRITE_SIZE are
collected in SEPARATE --pmc passes (they do not fit one pass), values are KiB.  The read-side correction is PER ACCESS PATTERN,
calibrated on this hardware with tools/pmc_calibrate.{hip,sh} (profiles/pmc_calibration.json):
  * lane-contiguous streaming reads of 4, 8 or 16 bytes per lane: FETCH_SIZE reports exactly HALF the bytes  => factor 2;
  * random accesses: FETCH_SIZE counts 64 B per REQUEST whatever its size (an 8-byte read, a 32-byte sector, a 64-byte line and a
    full 128-byte line all count 63.9 B; a Vec3f64 at an 8-aligned address 72.4 B = 1.13 requests) and random 64-byte and 128-byte
    lines are served at the same request rate (50.7 vs 47.1 G/s) => the bytes moved lie between 1x and 2x the counter; gather-bound
    kernels are listed with factor 1 (lower bound) and the x2 figure as upper bound.
WRITE_SIZE is exact: 16.06 B per 16-byte streaming store, 32.0 B per random 8-byte or aligned 32-byte store, 40.0 B per random
12-byte store (32-byte sectors, one in four straddles).
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


# kernels whose reads are dominated by random accesses (everything else streams lane-contiguous vectors)
GATHER_KERNELS = ("reorder_kernel", "voxel_reduce_kernel", "knn_grid_kernel", "knn_bruteforce_kernel", "voxel_mode_big_kernel")


def fetch_factor(kernel_name):
    return 1.0 if any(g in kernel_name for g in GATHER_KERNELS) else 2.0


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
    ap.add_argument("--key", default=None, help="key in hbm_traffic.json and file-name stem (default: the workload)")
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
        lines.append("# PMC passes (separate runs): values in KiB per dispatch; fetch factor per access pattern (profiles/pmc_calibration.json):")
        lines.append("#   x2 = lane-contiguous streaming reads (counter reports half); x1 = random-access kernels (64 B counted per request: lower bound, x2 = upper bound)")
        lines.append(f"{'kernel':110s} {'n':>4s} {'FETCH_KiB':>14s} {'factor':>6s} {'WRITE_KiB':>14s} {'HBM_bytes/launch':>18s}")
        for k in sorted(set(fetch) | set(write)):
            f = fetch.get(k, (0, 0.0, 0, 0))
            w = write.get(k, (0, 0.0, 0, 0))
            ff = fetch_factor(k)
            total = ff * f[1] * 1024 + w[1] * 1024
            lines.append(f"{k[:110]:110s} {max(f[0], w[0]):4d} {f[1]:14.1f} {ff:6.0f} {w[1]:14.1f} {total:18.0f}")
            if a.kernel in k:
                gather = ff == 1.0
                traffic = {"points": a.points, "kernel": k, "fetch_kib_raw": f[1], "write_kib": w[1], "fetch_factor": ff,
                           "bytes_per_launch": round(total),
                           "calibrated_on": ("random accesses: 64 B counted per request of any size up to a 128-byte line (tools/pmc_calibrate: gather_read<8> 63.9, "
                                             "gather_line<8> 63.9 B per access) -- lower bound, upper bound = x2" if gather else
                                             "lane-contiguous streaming reads: counter reports half the bytes (tools/pmc_calibrate: stream_read16 8.0, "
                                             "stream_read8 4.0, stream_read4 2.0 B per 16 / 8 / 4 requested)"),
                           "correction": f"FETCH_SIZE x{ff:g}, WRITE_SIZE x1, KiB x1024", "round": a.round}
                if gather:
                    traffic["bytes_per_launch_upper"] = round(2.0 * f[1] * 1024 + w[1] * 1024)
    key = a.key or a.workload
    txt = os.path.join(a.out, f"{a.round}_{key}_rocprof.txt")
    with open(txt, "w") as fh:
        fh.write("\n".join(lines) + "\n")
    print("\n".join(lines))
    if traffic:
        tpath = os.path.join(a.out, "hbm_traffic.json")
        allt = json.load(open(tpath)) if os.path.exists(tpath) else {}
        allt[key] = traffic
        json.dump(allt, open(tpath, "w"), indent=1)


if __name__ == "__main__":
    main()
