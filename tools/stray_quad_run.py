#!/usr/bin/env python3
"""The benchmark scene plus one quad that is not axis-aligned.  The reference boxes such a quad
by two opposite corners (quad.rs:86), so it sticks out of its box and the tree no longer bounds
its contents.  Checks that the two-pass walk (order-free walk for rays clear of the quad, the
reference's visit rule for the rest) returns what the reference's rule alone returns, and times
both.  tools/stray_quad_run.py [n_triangles] [width] [height]"""
import json
import os
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from raytracer_amd import RenderIntegrator, abi, scenes

n = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000
w = int(sys.argv[2]) if len(sys.argv) > 2 else 1200
h = int(sys.argv[3]) if len(sys.argv) > 3 else 1200
d = os.path.join(tempfile.gettempdir(), f"rt_stray_{os.getuid()}")
sp, cp = scenes.materialize_c3(d, n, w, h, batch=20, max_batches=4, min_batches=2)
with open(sp) as f:
    sc = json.load(f)
sc["elements"].append({"JSONQuad": {"q": scenes._v(60, 380, 300), "u": scenes._v(90, 40, 20), "v": scenes._v(-20, 70, 50),
                                    "material": scenes.metal(0.8, 0.8, 0.8, 0.0), "attractor": False}})
with open(sp, "w") as f:
    json.dump(sc, f)
ri = RenderIntegrator.new_from_json(sp, cp).prepare()
res = ri.resident(0)
cfg = ri.config.copy()
p2, t2 = res.primary_hits(cfg)
cfg.flags = abi.RT_FLAG_REFERENCE_VISIT_RULE
p1, t1 = res.primary_hits(cfg)
quad_id = ri.scene.contents.n_elements - 1
print(f"primary hits: ids equal {np.mean(p1 == p2):.6f}, t equal {np.mean(t1 == t2):.6f}, on the quad {np.mean(p1 == quad_id):.4f}", flush=True)
assert np.array_equal(p1, p2) and np.array_equal(t1, t2)
for name, flags in (("two-pass", 0), ("reference rule", abi.RT_FLAG_REFERENCE_VISIT_RULE)):
    cfg.flags = flags | abi.RT_FLAG_TIME_KERNELS
    res.render(cfg)
    _, st = res.render(cfg)
    print(f"{name}: {st.wall_ms:.1f} ms, {st.rays/st.wall_ms/1e3:.0f} Mrays/s, extend {st.trace_kernel_ms:.1f} shade {st.shade_kernel_ms:.1f} "
          f"other {st.other_kernel_ms:.1f} ms", flush=True)
