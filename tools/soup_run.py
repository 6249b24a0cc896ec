#!/usr/bin/env python3
"""Triangle soup with glass and metal triangles and a thin lens (the shape of BASELINE's largest
configuration) at a size given on the command line: primary hits and random rays against the
CPU oracle on a sample, then a timed render.  tools/soup_run.py [n_triangles] [width] [height] [--no-oracle]
With --no-oracle (for counter runs) the comparison is skipped and a JSON line with the ray counts is printed last."""
import os
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from raytracer_amd import RenderIntegrator, abi, scenes
from oracle import oracle

argv = [a for a in sys.argv[1:] if not a.startswith("--")]
no_oracle = "--no-oracle" in sys.argv
n = int(argv[0]) if len(argv) > 0 else 1_000_000
w = int(argv[1]) if len(argv) > 1 else 960
h = int(argv[2]) if len(argv) > 2 else 540
d = os.path.join(tempfile.gettempdir(), f"rt_soup_{os.getuid()}")
t0 = time.time()
sp, cp = scenes.materialize_soup(d, n, w, h, batch=16, max_batches=4, min_batches=2)
t1 = time.time()
ri = RenderIntegrator.new_from_json(sp, cp).prepare()
t2 = time.time()
res = ri.resident(0)
t3 = time.time()
print(f"files {t1-t0:.1f} s, host {t2-t1:.2f} s, upload {t3-t2:.2f} s, {ri.scene.contents.n_elements} elements, depth {ri.bvh_depth()}", flush=True)
if not no_oracle:
    rng = np.random.RandomState(1)
    m = 200_000
    o = rng.uniform(5.0, 550.0, size=(m, 3))
    dd = rng.normal(size=(m, 3))
    pg, tg = res.trace_rays(o, dd)
    po, to = oracle.trace_rays(ri.scene, o, dd)
    print(f"random rays: ids equal {np.mean(pg == po):.6f}, t equal {np.mean(tg == to):.6f}, hits {np.mean(po >= 0):.3f}", flush=True)
    assert np.array_equal(pg, po) and np.array_equal(tg, to)
cfg = ri.config.copy()
cfg.flags = abi.RT_FLAG_TIME_KERNELS
res.render(cfg)
_, st = res.render(cfg)
print(f"render {w}x{h}: {st.wall_ms:.1f} ms, {st.rays/st.wall_ms/1e3:.0f} Mrays/s, extend {st.trace_kernel_ms:.1f} shade {st.shade_kernel_ms:.1f} "
      f"other {st.other_kernel_ms:.1f} ms, rays/sample {st.rays/st.samples:.2f}", flush=True)
cfg.flags = abi.RT_FLAG_COUNT_VISITS
_, sc_ = res.render(cfg)
print(f"node visits/ray {sc_.node_visits/sc_.rays_traced:.1f}, leaf tests/ray {sc_.leaf_tests/sc_.rays_traced:.1f}, "
      f"lanes busy: node steps {sc_.node_visits/(64*sc_.wave_node_steps):.2f}, leaf tests {sc_.leaf_tests/(64*sc_.wave_leaf_steps):.2f}", flush=True)
if no_oracle:
    import json
    print(json.dumps({"triangles": n, "width": w, "height": h, "timed_renders": 2, "rays_per_render": int(st.rays), "rays_traced_per_render": int(sc_.rays_traced),
                      "extend_ms": st.trace_kernel_ms, "shade_ms": st.shade_kernel_ms, "wall_ms": st.wall_ms}), flush=True)
