#!/usr/bin/env python3
"""One rank's 1/8 share of the headline image rendered many times back to back: does the time settle (clocks, caches, allocator)?"""
import os, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from raytracer_amd import RenderIntegrator, abi, parallel, scenes
workdir = os.path.join(tempfile.gettempdir(), f"rt_bench_{os.getuid()}_0")
sp, cp = scenes.materialize_c3(workdir, 100_000, 1200, 1200)
ri = RenderIntegrator.new_from_json(sp, cp).prepare()
res = ri.resident(0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8
cfg = parallel.shard_config(ri.config, 0, n, 1)
cfg.flags = abi.RT_FLAG_TIME_KERNELS
for k in range(12):
    t0 = time.perf_counter()
    _, st = res.render(cfg, want_linear=False, want_samples=False)
    print(f"{k}: wall {1e3*(time.perf_counter()-t0):7.1f} ms  extend {st.trace_kernel_ms:6.1f} shade {st.shade_kernel_ms:6.1f} other {st.other_kernel_ms:5.1f}  bounce0 {st.trace_ms_by_depth[0]:5.1f}", flush=True)
