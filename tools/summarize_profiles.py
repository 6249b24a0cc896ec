#!/usr/bin/env python3
"""Condense rocprofv3 outputs under gpurun_out/ into the small files kept in profiles/.

  tools/summarize_profiles.py <tag> <kernel_stats_dir> <bench_json_of_a_pmc_pass> <pmc_dir> [<pmc_dir> ...] [--fetch-factor F]

Every pmc directory holds one `rocprofv3 --pmc ...` pass of `bench.py --profile-pass` (exactly one render).
Writes
  profiles/<tag>_kernel_stats.csv   copy of rocprofv3's --stats summary (of a `bench.py --steps 2 --warmup 1` run)
  profiles/<tag>_pmc.json           counter sums per kernel over that one render
  profiles/<tag>_traffic.json       fabric bytes per ray of the dominant kernel
  profiles/roofline_inputs.json     what bench.py derives its counter-based roofline figures from (and a tagged copy)
FETCH_SIZE and WRITE_SIZE are in KiB.  On gfx950 FETCH_SIZE counts 64 B per 128-B request of a wide streaming read,
hence the factor 2 (--fetch-factor overrides it with what tools/fetch_calibration measured for this access pattern)."""
import collections
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from raytracer_amd import abi  # noqa: E402


def short_name(kernel):
    k = kernel.split("(")[0].strip()
    if k.startswith("void "):
        k = k[5:]
    return k


def main():
    argv = sys.argv[1:]
    fetch_factor = 2.0
    if "--fetch-factor" in argv:
        i = argv.index("--fetch-factor")
        fetch_factor = float(argv[i + 1])
        del argv[i:i + 2]
    tag, stats_dir, bench_json = argv[:3]
    pmc_dirs = argv[3:]
    os.makedirs(os.path.join(ROOT, "profiles"), exist_ok=True)
    # a directory may hold the files of earlier runs too: the newest one counts
    ks = sorted(glob.glob(os.path.join(stats_dir, "**", "*_kernel_stats.csv"), recursive=True), key=os.path.getmtime)
    kernel_ms = {}
    if ks:
        shutil.copy(ks[-1], os.path.join(ROOT, "profiles", f"{tag}_kernel_stats.csv"))
        with open(ks[-1]) as fh:
            for row in csv.DictReader(fh):
                kernel_ms[short_name(row["Name"])] = {"calls": int(row["Calls"]), "avg_ms": float(row["AverageNs"]) / 1e6,
                                                      "total_ms": float(row["TotalDurationNs"]) / 1e6}
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    launches = collections.Counter()
    for d in pmc_dirs:
        for f in sorted(glob.glob(os.path.join(d, "**", "*_counter_collection.csv"), recursive=True), key=os.path.getmtime)[-1:]:
            seen = set()
            with open(f) as fh:
                for row in csv.DictReader(fh):
                    k = short_name(row["Kernel_Name"])
                    agg[k][row["Counter_Name"]] += float(row["Counter_Value"])
                    seen.add((k, row["Dispatch_Id"]))
            for k in {k for k, _ in seen}:
                launches[k] = max(launches[k], sum(1 for kk, _ in seen if kk == k))
    out = {k: dict(v, launches=launches[k]) for k, v in agg.items()}
    json.dump(out, open(os.path.join(ROOT, "profiles", f"{tag}_pmc.json"), "w"), indent=1, sort_keys=True)
    bench = json.loads([l for l in open(bench_json) if l.startswith("{")][-1])
    rays_traced = bench["rays_traced_per_step_rank0"] * bench["steps"]
    rays = bench["rays_per_step"] * bench["steps"]

    def member(k, prefix):  # k_extend stands for both walks of the closest-hit step: the general one and the packet walk of bounce 0
        return k == prefix or k.startswith(prefix + "<") or (prefix == "k_extend" and k.startswith("k_extend_packet<"))

    def family(prefix):  # all instantiations of a kernel template that ran in the profiled render
        tot = collections.defaultdict(float)
        for k, v in out.items():
            if member(k, prefix):
                for c, x in v.items():
                    tot[c] += x
        return dict(tot)

    kernels = {}
    for name in ("k_extend", "k_shade", "k_generate", "k_accumulate"):
        fam = family(name)
        if fam:
            # the instantiations the timed renders launch: k_extend's second template argument marks the counting build,
            # which only the one instrumented render of a bench run uses
            ms = [v for k, v in kernel_ms.items() if member(k, name) and not k.startswith(name + "<false, true")
                  and not k.startswith(name + "<true, true") and not k.startswith("k_extend_packet<true")]
            if ms:
                fam["kernel_avg_launch_ms"] = sum(m["total_ms"] for m in ms) / max(sum(m["calls"] for m in ms), 1)
            kernels[name] = fam
    # the counters belong to the kernels as they were when the passes ran: run this tool on the tree the passes were taken on
    inputs = {"tag": tag, "source_hash": abi.kernel_source_hash(), "source_hash_of": list(abi.KERNEL_SOURCES), "rays": rays_traced, "rays_accepted": rays, "fetch_factor": fetch_factor, "kernels": kernels,
              "workload": bench["config"]["workload"],
              "note": "counter sums of ONE render (bench.py --profile-pass), one rocprofv3 --pmc pass per counter group; "
                      "FETCH_SIZE / WRITE_SIZE in KiB; SQ_* cycle counters in quad-cycles; GRBM_GUI_ACTIVE summed over 8 XCDs"}
    for path in (os.path.join(ROOT, "profiles", "roofline_inputs.json"), os.path.join(ROOT, "profiles", f"{tag}_roofline_inputs.json")):
        json.dump(inputs, open(path, "w"), indent=1, sort_keys=True)
    traffic = {}
    ext = kernels.get("k_extend")
    if ext and "FETCH_SIZE" in ext:
        fetch = ext["FETCH_SIZE"] * 1024.0 * fetch_factor
        write = ext.get("WRITE_SIZE", 0.0) * 1024.0
        traffic = {
            "kernel": "k_extend", "rays_traced": rays_traced, "launches": ext["launches"],
            "fetch_bytes_per_ray": fetch / rays_traced, "write_bytes_per_ray": write / rays_traced,
            "hbm_bytes_per_ray": (fetch + write) / rays_traced,
            "note": f"FETCH_SIZE (KiB) x {fetch_factor} + WRITE_SIZE (KiB); Infinity-Cache hits are included",
        }
        if "TCC_HIT_sum" in ext:
            traffic["l2_hit_rate"] = ext["TCC_HIT_sum"] / (ext["TCC_HIT_sum"] + ext["TCC_MISS_sum"])
    json.dump(traffic, open(os.path.join(ROOT, "profiles", f"{tag}_traffic.json"), "w"), indent=1, sort_keys=True)
    print(json.dumps(traffic, indent=1))


if __name__ == "__main__":
    main()
