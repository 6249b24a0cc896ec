#!/usr/bin/env python3
"""Join the MARK lines of tools/coherence_probe.py with the k_extend durations of the rocprofv3 kernel trace taken over
the same run: usage  coherence_probe_report.py <probe stdout> <rocprof output dir>  -> JSON lines, one per traced order."""
import csv
import glob
import json
import re
import sys

marks = [l.split(" ", 1)[1].strip() for l in open(sys.argv[1]) if l.startswith("MARK ")]
rows = []
for path in glob.glob(sys.argv[2] + "/**/*kernel_trace.csv", recursive=True):
    with open(path) as fh:
        for r in csv.DictReader(fh):
            if r["Kernel_Name"].startswith("void k_extend") or r["Kernel_Name"].startswith("k_extend"):
                rows.append((int(r["Start_Timestamp"]), (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6))
rows.sort()
big = [ms for _, ms in rows if ms > 0.2]  # the one-ray launches of scene set-up are microseconds long
if len(big) != len(marks):
    print(json.dumps({"error": f"{len(marks)} marks but {len(big)} large k_extend launches", "ms": big}))
    sys.exit(1)
base = {}
for m, ms in zip(marks, big):
    n = int(re.search(r"n=(\d+)", m).group(1))
    name = m.split(" ")[0]
    label = m[len(name) + 1:m.rfind(" n=")]
    if label == "natural":
        base[name] = ms
    print(json.dumps({"set": name, "order": label, "rays": n, "k_extend_ms": round(ms, 3), "ps_per_ray": round(ms * 1e9 / n, 1),
                      "vs_natural": round(ms / base[name], 3) if name in base else None}))
