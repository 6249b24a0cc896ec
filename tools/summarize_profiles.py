#!/usr/bin/env python3
"""Condense rocprofv3 outputs under gpurun_out/ into the small files kept in profiles/.

  tools/summarize_profiles.py <tag> <kernel_stats_dir> <bench_json_of_a_pmc_run> <pmc_dir> [<pmc_dir> ...]

Writes profiles/<tag>_kernel_stats.csv (copy of rocprofv3's --stats summary),
profiles/<tag>_pmc.json (counter sums per kernel over one render) and
profiles/<tag>_traffic.json (fabric bytes per ray of the dominant kernel, with the
gfx950 correction: FETCH_SIZE counts 64 B per 128-B request, so it is doubled;
WRITE_SIZE is exact; both are in KiB)."""
import collections
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    tag, stats_dir, bench_json = sys.argv[1:4]
    pmc_dirs = sys.argv[4:]
    os.makedirs(os.path.join(ROOT, "profiles"), exist_ok=True)
    # a directory may hold the files of earlier runs too: the newest one counts
    ks = sorted(glob.glob(os.path.join(stats_dir, "*", "*_kernel_stats.csv")), key=os.path.getmtime)
    if ks:
        shutil.copy(ks[-1], os.path.join(ROOT, "profiles", f"{tag}_kernel_stats.csv"))
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    launches = collections.Counter()
    for d in pmc_dirs:
        for f in sorted(glob.glob(os.path.join(d, "*", "*_counter_collection.csv")), key=os.path.getmtime)[-1:]:
            seen = set()
            with open(f) as fh:
                for row in csv.DictReader(fh):
                    k = row["Kernel_Name"].split("(")[0]
                    agg[k][row["Counter_Name"]] += float(row["Counter_Value"])
                    key = (k, row["Dispatch_Id"])
                    if key not in seen:
                        seen.add(key)
            for k, _ in seen:
                launches[k] = max(launches[k], sum(1 for kk, _ in seen if kk == k))
    out = {k: dict(v, launches=launches[k]) for k, v in agg.items()}
    json.dump(out, open(os.path.join(ROOT, "profiles", f"{tag}_pmc.json"), "w"), indent=1, sort_keys=True)
    bench = json.load(open(bench_json))
    rays = bench["rays_per_step"] * bench["steps"]
    # one timed render + one instrumented (counted) render run per bench invocation: use the timed kernel only
    # the timed kernel is the uninstrumented instantiation (second template argument false)
    timed = [v for k, v in out.items() if "k_extend<" in k and k.split("k_extend<")[1].split(",")[1].strip().startswith("false")]
    ext = max(timed, key=lambda v: v["launches"]) if timed else None
    traffic = {}
    if ext and "FETCH_SIZE" in ext:
        fetch = ext["FETCH_SIZE"] * 1024.0 * 2.0
        write = ext.get("WRITE_SIZE", 0.0) * 1024.0
        traffic = {
            "kernel": "k_extend", "rays": rays, "launches": ext["launches"],
            "fetch_bytes_per_ray": fetch / rays, "write_bytes_per_ray": write / rays,
            "hbm_bytes_per_ray": (fetch + write) / rays,
            "note": "FETCH_SIZE (KiB) x 2 (gfx950 counts 64 B per 128-B request) + WRITE_SIZE (KiB); Infinity-Cache hits are included",
        }
        if "TCC_HIT_sum" in ext:
            traffic["l2_hit_rate"] = ext["TCC_HIT_sum"] / (ext["TCC_HIT_sum"] + ext["TCC_MISS_sum"])
    json.dump(traffic, open(os.path.join(ROOT, "profiles", f"{tag}_traffic.json"), "w"), indent=1, sort_keys=True)
    print(json.dumps(traffic, indent=1))


if __name__ == "__main__":
    main()
