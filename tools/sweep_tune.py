#!/usr/bin/env python3
"""Render the headline image with one tuning parameter of the handle swept (rt_hip_scene_tune), all on one scene copy in one
process: tools/sweep_tune.py name v1 v2 ... [--reuse]   -> kernel times per value (the default is measured first and last)."""
import os
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from raytracer_amd import RenderIntegrator, abi, scenes  # noqa: E402

argv = [a for a in sys.argv[1:] if a != "--reuse"]
name, values = argv[0], [int(v) for v in argv[1:]]
workdir = os.path.join(tempfile.gettempdir(), f"rt_bench_{os.getuid()}_0")
sp, cp = scenes.materialize_c3(workdir, 100_000, 1200, 1200)
ri = RenderIntegrator.new_from_json(sp, cp).prepare()
res = ri.resident(0)
cfg = ri.config.copy()
cfg.flags = abi.RT_FLAG_TIME_KERNELS | (abi.RT_FLAG_REUSE_PRIMARY_HITS if "--reuse" in sys.argv else 0)
res.render(cfg, want_linear=False, want_samples=False)
for v in values + values[:1]:
    res.tune(**{name: v})
    best = None
    for _ in range(2):
        _, st = res.render(cfg, want_linear=False, want_samples=False)
        if best is None or st.trace_kernel_ms < best.trace_kernel_ms:
            best = st
    print(f"{name}={v}: extend {best.trace_kernel_ms:7.1f} shade {best.shade_kernel_ms:7.1f} other {best.other_kernel_ms:6.1f} ms   extend by bounce "
          + " ".join(f"{x:6.1f}" for x in list(best.trace_ms_by_depth)[:8]), flush=True)
