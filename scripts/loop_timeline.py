#!/usr/bin/env python3
"""Reads a rocprofv3 kernel trace (csv) of scripts/r03_probe.py loop and prints, for the LAST whitened loop in it, what ran when:
per kernel family the number of dispatches, the mean duration, and — for the SpMM of the intermediate iterations — how long the
statistics kernels ran inside its window.  Usage: loop_timeline.py <dir with *_kernel_trace.csv> [label]"""
import csv
import glob
import json
import sys

path = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = []
with open(path) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
fam = lambda k: ("spmm" if "spmm_rows_kernel" in k else "gram32" if ("gram32_kernel" in k or "gram16_kernel" in k) else "gram64" if "gram_kernel" in k else
                 "project" if "project_" in k and "pack" not in k else "other")
# the last whitened loop = everything after the last-but-(iters) projection ... simply: the last 7 projections and what lies between
proj = [i for i, r in enumerate(rows) if fam(r[2]) == "project"]
last = proj[-7:]
lo, hi = rows[last[0]][0], rows[last[-1]][1]
win = [r for r in rows if lo <= r[0] <= hi]
out = {"label": sys.argv[2] if len(sys.argv) > 2 else "", "window_ms": (hi - lo) / 1e6, "per_iteration_ms": (hi - lo) / 1e6 / 6}
for name in ("spmm", "gram32", "gram64", "project", "other"):
    d = [(e - s) / 1e6 for s, e, k in win if fam(k) == name]
    if d:
        out[name] = {"dispatches": len(d), "mean_ms": sum(d) / len(d), "sum_ms": sum(d)}
# overlap of gram32 with the SpMM
sp = [(s, e) for s, e, k in win if fam(k) == "spmm"]
gr = [(s, e) for s, e, k in win if fam(k) == "gram32"]
ov = sum(max(0, min(e1, e2) - max(s1, s2)) for s1, e1 in sp for s2, e2 in gr) / 1e6
out["gram32_inside_spmm_ms_total"] = ov
print(json.dumps(out))
