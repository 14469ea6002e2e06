#!/usr/bin/env python3
"""DESIGN.md section 4's table, GENERATED from the committed bench lines (round-4 review: no hand-picked best-box figures):
  tools/gen_workload_table.py profiles/r05_workloads_box*.jsonl > profiles/r05_workloads_table.md
One row per (workload, requested plan): algorithmic bytes per point, the kernel family the library reported (pst_last_plan_kinds), and for
every box the fraction of the 8 TB/s peak with the kernel milliseconds per 10^8-point step; last column = the range over the boxes."""
import json
import sys
from collections import OrderedDict

rows = OrderedDict()
boxes = []
for path in sys.argv[1:]:
    box = path.split("_")[-1].split(".")[0]
    boxes.append(box)
    for line in open(path):
        line = line.strip()
        if not line.startswith("{"):
            continue
        d = json.loads(line)
        c, r = d["config"], d["roofline"]
        name = c["workload"].split(":")[0]
        if name.startswith("randomlayout"):
            name += f" (seed {c['workload'].split('seed ')[1].split(':')[0]})" if "seed " in c["workload"] else ""
        key = (name, c.get("plan_requested") or "auto")
        e = rows.setdefault(key, {"bpp": r.get("algorithmic_bytes_per_point"), "plans": set(), "per_box": {}})
        e["plans"].add("+".join(c["plan"]) if c.get("plan") else "-")
        e["per_box"][box] = (r["frac"], r["kernel_ms_avg"])
print("| workload (`bench.py --workload`) | plan requested | kernel family reported | B/pt | " + " | ".join(f"{b}: frac of 8 TB/s (ms)" for b in boxes) + " | range |")
print("|---|---|---|---|" + "---|" * len(boxes) + "---|")
for (name, plan), e in rows.items():
    cells = []
    fr = []
    for b in boxes:
        if b in e["per_box"]:
            f, ms = e["per_box"][b]
            fr.append(f)
            cells.append(f"{f:.3f} ({ms:.3f})")
        else:
            cells.append("")
    rng = f"{min(fr):.3f}–{max(fr):.3f}" if len(fr) > 1 else (f"{fr[0]:.3f}" if fr else "")
    print(f"| {name} | {plan} | {', '.join(sorted(e['plans']))} | {e['bpp']} | " + " | ".join(cells) + f" | {rng} |")
