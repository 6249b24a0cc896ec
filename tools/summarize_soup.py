#!/usr/bin/env python3
"""Condense the counter passes of tools/profile_soup.sh (gpurun_out/soup/) into profiles/<tag>_soup.json:
per kernel the counter sums, and for k_extend the fabric bytes per ray (FETCH_SIZE is in KiB and counts
64 B per 128-B request on gfx950, so it is doubled; WRITE_SIZE in KiB is exact).  tools/summarize_soup.py <tag>"""
import collections
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
O = os.path.join(ROOT, "gpurun_out", "soup")
tag = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(float))
launches = collections.defaultdict(set)
for d in ("fetch", "write", "sq", "grbm"):
    for f in sorted(glob.glob(os.path.join(O, d, "*", "*_counter_collection.csv")), key=os.path.getmtime)[-1:]:
        with open(f) as fh:
            for row in csv.DictReader(fh):
                k = row["Kernel_Name"].split("(")[0]
                agg[k][row["Counter_Name"]] += float(row["Counter_Value"])
                launches[(k, d)].add(row["Dispatch_Id"])
out = {k: dict(v, launches=max(len(s) for (kk, _), s in launches.items() if kk == k)) for k, v in agg.items()}
run = json.loads(open(os.path.join(O, "fetch.log")).read().strip().splitlines()[-1])
plain = json.loads(open(os.path.join(O, "plain.log")).read().strip().splitlines()[-1])
# the two timed renders and the warm-up run the uninstrumented kernel: three renders' rays
timed = {k: v for k, v in out.items() if "k_extend<" in k and k.split("k_extend<")[1].split(",")[1].strip().startswith("false")}
name, ext = max(timed.items(), key=lambda kv: kv[1]["launches"])
rays = 2.0 * run["rays_traced_per_render"]
fetch, write = ext["FETCH_SIZE"] * 1024.0 * 2.0, ext.get("WRITE_SIZE", 0.0) * 1024.0
summary = {
    "run": plain, "kernel": name, "renders_counted": 2, "rays": rays,
    "fetch_bytes_per_ray": fetch / rays, "write_bytes_per_ray": write / rays, "fabric_bytes_per_ray": (fetch + write) / rays,
    "fabric_gb_s_at_plain_speed": (fetch + write) / 2.0 / (plain["extend_ms"] * 1e-3) / 1e9,
    "l2_hit_rate": ext.get("TCC_HIT_sum", 0.0) / max(ext.get("TCC_HIT_sum", 0.0) + ext.get("TCC_MISS_sum", 0.0), 1.0),
    "counters": out,
}
os.makedirs(os.path.join(ROOT, "profiles"), exist_ok=True)
json.dump(summary, open(os.path.join(ROOT, "profiles", f"{tag}_soup.json"), "w"), indent=1, sort_keys=True)
ks = sorted(glob.glob(os.path.join(O, "stats", "*", "*_kernel_stats.csv")), key=os.path.getmtime)
if ks:
    shutil.copy(ks[-1], os.path.join(ROOT, "profiles", f"{tag}_soup_kernel_stats.csv"))
print(json.dumps({k: v for k, v in summary.items() if k != "counters"}, indent=1))
