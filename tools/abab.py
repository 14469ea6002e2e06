#!/usr/bin/env python3
"""Same-box A/B protocol for every tuning claim in DESIGN.md (round 5): >= 8 INTERLEAVED pairs of bench.py processes (A B A B ...), the
kernel time of each (HIP events, roofline.kernel_ms_avg), median and inter-quartile range per side, the ratio of the medians and how many of
the pairs agree with its sign.  A and B differ by environment switches (PST_*), by the library build (PASTURE_AMD_LIB) or by bench.py flags.

  tools/abab.py --workload las0_to_columns --a "PST_RESIDENT=0" --b "PST_RESIDENT=6" [--pairs 8] [--steps 20] [--flags-a "..."] [--out profiles/x.txt]
"""
import argparse
import json
import os
import shlex
import statistics
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def quartiles(v):
    q = statistics.quantiles(v, n=4, method="inclusive") if len(v) > 1 else [v[0]] * 3
    return q[0], q[1], q[2]


def run(workload, env_s, flags, steps, points):
    env = dict(os.environ)
    for kv in shlex.split(env_s or ""):
        k, _, v = kv.partition("=")
        env[k] = v
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--workload", workload, "--steps", str(steps), "--warmup", "3", "--no-cpu-baseline", "--no-north-star",
           "--no-extra-legs"] + (["--points", str(points)] if points else []) + shlex.split(flags or "")
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    line = json.loads(out.stdout.strip().splitlines()[-1])
    return line["roofline"]["kernel_ms_avg"], line["roofline"]["frac"], line["config"].get("plan")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", required=True)
    ap.add_argument("--a", default="")
    ap.add_argument("--b", default="")
    ap.add_argument("--flags-a", default="")
    ap.add_argument("--flags-b", default="")
    ap.add_argument("--pairs", type=int, default=8)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--points", type=int, default=0)
    ap.add_argument("--out", default="")
    a = ap.parse_args()
    ms = {"A": [], "B": []}
    frac = {"A": [], "B": []}
    plans = {}
    for i in range(a.pairs):
        order = ("A", "B") if i % 2 == 0 else ("B", "A")  # alternate who goes first: drift inside a pair cancels over the pairs
        for side in order:
            m, f, plan = run(a.workload, a.a if side == "A" else a.b, a.flags_a if side == "A" else a.flags_b, a.steps, a.points)
            ms[side].append(m)
            frac[side].append(f)
            plans[side] = plan
    qa, qb = quartiles(ms["A"]), quartiles(ms["B"])
    ratio = qb[1] / qa[1]
    agree = sum(1 for x, y in zip(ms["A"], ms["B"]) if (y < x) == (ratio < 1))
    lines = [
        f"ABAB {a.workload}" + (f" points={a.points}" if a.points else "") + f": {a.pairs} interleaved pairs, {a.steps} timed steps each (HIP events, kernel ms per step)",
        f"  A: {a.a or '(default)'} {a.flags_a}  plan={plans.get('A')}",
        f"  B: {a.b or '(default)'} {a.flags_b}  plan={plans.get('B')}",
        "  A ms: " + " ".join(f"{x:.4f}" for x in ms["A"]),
        "  B ms: " + " ".join(f"{x:.4f}" for x in ms["B"]),
        f"  A median {qa[1]:.4f} ms (IQR {qa[0]:.4f}-{qa[2]:.4f}), frac of peak median {statistics.median(frac['A']):.4f}",
        f"  B median {qb[1]:.4f} ms (IQR {qb[0]:.4f}-{qb[2]:.4f}), frac of peak median {statistics.median(frac['B']):.4f}",
        f"  B / A = {ratio:.4f} ({(1 / ratio - 1) * 100:+.1f} % throughput for B); {agree} of {a.pairs} pairs agree in sign"
        + ("; the IQRs overlap: NOT a decision" if not (qb[2] < qa[0] or qa[2] < qb[0]) else "; the IQRs are disjoint"),
    ]
    text = "\n".join(lines)
    print(text, flush=True)
    if a.out:
        with open(a.out, "a") as f:
            f.write(text + "\n\n")


if __name__ == "__main__":
    main()
