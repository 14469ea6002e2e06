#!/usr/bin/env python3
"""Per-phase vector-instruction budget of the kNN box kernel (round-4 review, item 3): STATIC instruction counts of knn_tile2_kernel's loops,
read off the compiler's assembly (CPU only: hipcc cross-compiles), times the DYNAMIC trip counts a -DPST_KNN_STATS run of the same workload
prints (scan steps, insertion iterations and insertion rounds per query wave, query waves per box), against the measured total
(SQ_INSTS_VALU of a rocprofv3 --pmc pass).  Ablation runs under rocprofv3 (tools/r05_knn_phases.sh) take minutes per pass on the GPU box; this
needs one stats run.

  tools/knn_phase_budget.py --stats-line "<the two '[pst knn tile2 D]' lines>" --valu-total 9.12e9 --points 100000000 > profiles/r05_knn_phases.txt

How the assembly is cut: -DPST_KNN_MARKS leaves comment lines (PST_KNN_MARK) that NAME the loops -- the chunk loop is the loop around the marker
chunk_begin, the scan loop the one around step_head, the insertion loop the one around flush_iter -- and LLVM annotates every basic block with its
innermost loop ("in Loop: Header=BBn_m Depth=d").  A block is charged to its innermost loop; blocks of the chunk loop itself are split at the
markers scan_loop_begin / proof_begin / fit_begin / fit_end in text order (the markers may float by a few instructions: volatile asm is ordered
against other volatile asm only)."""
import argparse
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KERNEL = "_ZN12_GLOBAL__N_116knn_tile2_kernelILi16ELi512ELi3000ELb0ELi4ELi4ELb0ELb0ELi1EEEvNS_9Tile2ArgsE"


def kernel_asm():
    out = "/tmp/pst_knn_marks.s"
    src = os.path.join(ROOT, "pasture_amd", "csrc", "normals_tile.hip")
    cmd = ["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "--offload-arch=gfx950", "-Wno-unused-function", "-I" + os.path.join(ROOT, "pasture_amd", "csrc", "build"),
           "-DPST_KNN_MARKS", "--cuda-device-only", "-S", src, "-o", out]
    subprocess.run(cmd, check=True, stderr=subprocess.DEVNULL)
    lines, on = [], False
    for l in open(out):
        if l.startswith(KERNEL + ":"):
            on = True
        if on:
            lines.append(l.rstrip("\n"))
            if "s_endpgm" in l:
                break
    return lines


def classify(op):
    if op.startswith("v_"):
        return "valu"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith(("global_", "scratch_", "buffer_", "flat_")):
        return "vmem"
    if op.startswith("s_"):
        return "salu"
    return None


def parse(lines):
    """-> blocks: list of dict(label, loop (innermost header label or None), depth, counts, marks, first_line)"""
    blocks = []
    cur = {"label": "entry", "loop": None, "depth": 0, "counts": {"valu": 0, "lds": 0, "vmem": 0, "salu": 0}, "marks": [], "line": 0}
    blocks.append(cur)
    pending_header = None
    for i, l in enumerate(lines):
        s = l.strip()
        m = re.match(r"^(\.LBB\d+_\d+):", s) or re.match(r"^; %bb\.(\d+):", s)
        if m:
            label = m.group(1) if s.startswith(".LBB") else "bb." + m.group(1)
            cur = {"label": label, "loop": None, "depth": 0, "counts": {"valu": 0, "lds": 0, "vmem": 0, "salu": 0}, "marks": [], "line": i}
            blocks.append(cur)
            pending_header = label
        mm = re.search(r"in Loop: Header=(BB\d+_\d+) Depth=(\d+)", s)
        if mm:
            cur["loop"], cur["depth"] = "." + "L" + mm.group(1), int(mm.group(2))
        mm = re.search(r"This (?:Inner )?Loop Header: Depth=(\d+)", s)
        if mm and pending_header:
            cur["loop"], cur["depth"] = pending_header, int(mm.group(1))
        if "PSTMARK" in s:
            cur["marks"].append(s.split("PSTMARK")[1].strip())
            cur.setdefault("mark_lines", {})[s.split("PSTMARK")[1].strip()] = i
            continue
        if not s or s.startswith((";", ".")):
            continue
        k = classify(s.split()[0])
        if k:
            cur["counts"][k] += 1
    return blocks


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scan-steps", type=float, required=True, help="scan steps per query wave (stats build)")
    ap.add_argument("--flush-iters", type=float, required=True, help="insertion iterations per query wave")
    ap.add_argument("--flushes", type=float, required=True, help="insertion rounds per query wave")
    ap.add_argument("--waves-per-box", type=float, required=True, help="query waves per box")
    ap.add_argument("--wg-waves", type=int, default=8)
    ap.add_argument("--valu-total", type=float, required=True, help="SQ_INSTS_VALU of the kernel, one launch")
    ap.add_argument("--points", type=float, default=1e8)
    ap.add_argument("--kernel-ms", type=float, default=0.0)
    ap.add_argument("--stats-text", default="", help="the stats build's lines, quoted verbatim in the output")
    a = ap.parse_args()
    blocks = parse(kernel_asm())
    loop_of_mark = {}
    for b in blocks:
        for m in b["marks"]:
            loop_of_mark.setdefault(m, (b["loop"], b["depth"]))
    chunk_loop = loop_of_mark.get("chunk_begin", (None, 0))[0]
    scan_loop = loop_of_mark.get("step_head", (None, 0))[0]
    flush_loop = loop_of_mark.get("flush_iter", (None, 0))[0]
    line_of = {}
    for b in blocks:
        for m in b["marks"]:
            line_of.setdefault(m, b["mark_lines"][m])  # (the marker's own line: it may sit at the END of a long block)

    def add(dst, c):
        for k in dst:
            dst[k] += c[k]
    zero = lambda: {"valu": 0, "lds": 0, "vmem": 0, "salu": 0}  # noqa: E731
    reg = {n: zero() for n in ("per workgroup: directory, staging, loop-invariant set-up", "chunk set-up (hand-out, query cell, nine trimmed segments)", "scan step (segment table shift, ballots, 4 candidates)",
                               "insertion round: set-up + limit update", "insertion iteration (17 v_med3_u32 + key)", "proof (bin gaps; rare f64 pair settle)", "plane fit (16 gathers, 12 sums, cubic, cross products)",
                               "results (index, stores) + chunk loop latch", "other inner loops of a chunk (row search of the hand-out)")}
    names = list(reg)
    for b in blocks:
        c, lp = b["counts"], b["loop"]
        if lp == flush_loop and flush_loop:
            add(reg[names[4]], c)
        elif lp == scan_loop and scan_loop:
            # blocks of the scan loop: the step path lies before the insertion loop in text order, the insertion round's set-up / limit update around it
            if b["line"] < line_of.get("flush_loop_begin", 1 << 30) - 40:
                add(reg[names[2]], c)
            else:
                add(reg[names[3]], c)
        elif lp == chunk_loop and chunk_loop:
            ln = b["line"]
            if ln < line_of.get("scan_loop_begin", 0):
                add(reg[names[1]], c) if ln >= line_of.get("chunk_begin", 0) - 60 else add(reg[names[7]], c)
            elif ln < line_of.get("proof_end", 0):
                add(reg[names[5]], c)
            elif ln < line_of.get("fit_end", 0):
                add(reg[names[6]], c)
            else:
                add(reg[names[7]], c)
        elif b["depth"] >= 2:
            add(reg[names[8]], c)
        else:
            add(reg[names[0]], c)
    qw = a.points / 64.0
    per_qw_total = a.valu_total / qw
    trips = {names[0]: a.wg_waves / a.waves_per_box, names[1]: 1.0, names[2]: a.scan_steps, names[3]: a.flushes, names[4]: a.flush_iters, names[5]: 1.0, names[6]: 1.0, names[7]: 1.0, names[8]: 4.0}
    print("# knn_tile2_kernel<16, 512, 3000, ...> (10^8 uniform points, k = 16): vector (wave64) instructions per QUERY WAVE (64 queries), by phase")
    print("# static = instructions of the phase's basic blocks in the compiler's assembly (tools/knn_phase_budget.py, -DPST_KNN_MARKS); trips = how often a query wave")
    print("# runs them (stats build, below); static x trips is an UPPER estimate for phases with rarely taken blocks (the proof's f64 pair settle: 15 blocks,")
    print("# one wave in five enters any) and exact for the loops.  The per-workgroup part is paid by all %d waves of a workgroup, %.2f of them per query wave." % (a.wg_waves, a.wg_waves / a.waves_per_box))
    if a.stats_text:
        for l in a.stats_text.split("\\n"):
            print("#   " + l)
    print()
    print(f"{'phase':92s} {'VALU':>6s} {'LDS':>5s} {'VMEM':>5s} {'SALU':>5s} {'trips':>7s} {'VALU x trips':>13s} {'share':>6s}")
    tot = 0.0
    rows = []
    for n in names:
        c = reg[n]
        v = c["valu"] * trips[n]
        if n == names[5]:
            v = min(v, 260.0 + 0.2 * (c["valu"] - 260.0))  # (the pair-settle blocks: entered by one wave in five)
        rows.append((n, c, trips[n], v))
        tot += v
    for n, c, t, v in rows:
        print(f"{n:92s} {c['valu']:6d} {c['lds']:5d} {c['vmem']:5d} {c['salu']:5d} {t:7.2f} {v:13.0f} {100.0 * v / per_qw_total:5.1f}%")
    print(f"{'sum of the estimates':92s} {'':6s} {'':5s} {'':5s} {'':5s} {'':7s} {tot:13.0f} {100.0 * tot / per_qw_total:5.1f}%")
    print(f"{'measured: SQ_INSTS_VALU / query waves':92s} {'':6s} {'':5s} {'':5s} {'':5s} {'':7s} {per_qw_total:13.0f} 100.0%")
    if a.kernel_ms:
        bound = a.valu_total / 6.144e11 * 1e3
        print(f"\nissue bound: {a.valu_total:.3e} wave instructions / 6.144e11 per second (256 CUs x 4 SIMDs x 2.4 GHz / 4 cycles) = {bound:.2f} ms; kernel {a.kernel_ms:.2f} ms = {bound / a.kernel_ms:.2f} of it")


if __name__ == "__main__":
    main()
