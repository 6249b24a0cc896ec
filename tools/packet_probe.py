#!/usr/bin/env python3
"""What the packet walk is worth on rays other than primary ones: the same ray sets traced by the general walk and by the
packet walk (tune packet=2), under rocprofv3 --kernel-trace (join with tools/coherence_probe_report.py).  Sets: primary rays;
rays from the primary hit points aimed at random points of the light (what the mixture density's attractor half sends);
cosine-distributed rays from the same points."""
import os
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from raytracer_amd import RenderIntegrator, abi, scenes  # noqa: E402

rng = np.random.default_rng(3)
workdir = os.path.join(tempfile.gettempdir(), f"rt_bench_{os.getuid()}_0")
sp, cp = scenes.materialize_c3(workdir, 100_000, 1200, 1200)
ri = RenderIntegrator.new_from_json(sp, cp).prepare()
res = ri.resident(0)
sc = ri.scene.contents
cam = sc.camera
c = np.array(cam.camera_center.tuple())
p00, du, dv = (np.array(getattr(cam, k).tuple()) for k in ("pixel00_loc", "pixel_delta_u", "pixel_delta_v"))
W = H = 1200
SPP = int(os.environ.get("PROBE_SPP", "32"))
ys, xs = np.meshgrid(np.arange(H), np.arange(W), indexing="ij")
order = np.lexsort((xs.ravel() % 8, ys.ravel() % 8, xs.ravel() // 8, ys.ravel() // 8))
px = np.repeat(xs.ravel()[order], SPP) + rng.random(W * H * SPP)
py = np.repeat(ys.ravel()[order], SPP) + rng.random(W * H * SPP)
d0 = p00 + du * px[:, None] + dv * py[:, None] - c
o0 = np.broadcast_to(c, d0.shape).copy()
res.tune(packet=0)
print(f"MARK primary general n={len(o0)}", flush=True)
prim0, t0 = res.trace_rays(o0, d0)
res.tune(packet=2)
print(f"MARK primary packet n={len(o0)}", flush=True)
prim1, t1 = res.trace_rays(o0, d0)
assert np.array_equal(prim0, prim1) and np.array_equal(t0, t1)
hit = prim0 >= 0
dn = d0[hit] / np.linalg.norm(d0[hit], axis=1, keepdims=True)
p = o0[hit] + dn * t0[hit][:, None]
q = sc.quads[0]
q0, u, v = np.array(q.q.tuple()), np.array(q.u.tuple()), np.array(q.v.tuple())
to_light = q0 + rng.random((len(p), 1)) * u + rng.random((len(p), 1)) * v - p
cosine = rng.normal(size=p.shape)
for name, d in (("to_light", to_light), ("random_directions", cosine)):
    out = []
    for label, mode in (("general", 0), ("packet", 2)):
        res.tune(packet=mode)
        print(f"MARK {name} {label} n={len(p)}", flush=True)
        out.append(res.trace_rays(p, d))
    assert np.array_equal(out[0][0], out[1][0]) and np.array_equal(out[0][1], out[1][1])
print("DONE", flush=True)
