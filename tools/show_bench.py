"""Print the interesting fields of bench.py JSON lines."""
import json
import sys

for f in sys.argv[1:]:
    try:
        d = json.load(open(f))
        r = d["roofline"]
        print(f, "Mrays/s %.0f" % d["value"], "step_ms %.0f" % d["ms_per_step"], "extend %.0f" % r["kernel_ms_per_step"],
              "shade %.0f" % r.get("shade_kernel_ms_per_step", 0), "other %.0f" % r["other_kernels_ms_per_step"],
              "frac %.3f" % (r.get("frac") or 0), "Vint %.2f Vleaf %.2f" % (r["node_visits_per_ray_device"], r["leaf_tests_per_ray_device"]))
        lu, wc = r.get("simd_lane_utilisation", {}), r.get("wave_cycle_shares", {})
        print("   lanes node %.3f leaf %.3f | wave cycles refill %.3f node %.3f leaf %.3f" % (lu.get("node_steps", 0), lu.get("leaf_tests", 0),
              wc.get("refill", 0), wc.get("node", 0), wc.get("leaf", 0)))
        print("   extend ms by depth", ["%.0f" % x for x in r.get("kernel_ms_by_depth_per_step", [])],
              "Mrays", [int(x / 1e6) for x in d["rays_by_depth_rank0_last_step"]])
    except Exception as e:  # noqa: BLE001
        print(f, "ERR", e)
