#!/usr/bin/env python3
"""Soak run with rays chosen to sit on the decisions' edges, over the random mixed-primitive scenes of tests/test_gpu_fuzz.py:
directions along the axes and in the coordinate planes, components of 1e-30 and of exactly -0.0, origins on quad planes and
sphere surfaces, rays aimed at triangle and quad corners and at sphere silhouettes.  The device must equal the oracle bit for
bit, except that it may find a hit the REFERENCE loses: a ray through a box corner or edge can fail the reference's f64 slab
test (aabb.rs:187-204) by a last bit, and the primitives below are then never tested, while the device's boxes are conservative.
So: wherever the two differ the device's hit must be nearer (or equally near, at a tie the reference resolved among fewer
primitives), never farther, and never a miss where the oracle hits -- with one exception that is counted apart: a hit the
oracle reports on a triangle of zero area.  Its determinant is rounding noise; a ray aimed exactly at the triangle's line can
come out of triangle.rs:76-117 as a "hit" at any t, far from the triangle's box, and the order-free walk, which drops boxes
that begin beyond the closest hit so far, may never test that triangle.  tools/adversarial_soak.py FIRST LAST"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from oracle import oracle
from raytracer_amd import abi
from test_gpu_fuzz import _scene

first, last = int(sys.argv[1]), int(sys.argv[2])
t0 = time.time()
bad = []
n_rays = n_differ = n_noise = 0
for seed in range(first, last):
    ri = _scene(seed)
    sc = ri.scene.contents
    rng = np.random.RandomState(5000 + seed)
    corners = []
    for i in range(sc.n_triangles):
        t = sc.triangles[i]
        v0 = np.array(t.v0.tuple())
        corners += [v0, v0 + np.array(t.v0v1.tuple()), v0 + np.array(t.v0v2.tuple()), v0 + 0.5 * np.array(t.v0v1.tuple())]
    planes = []
    for i in range(sc.n_quads):
        q = sc.quads[i]
        q0, u, v = np.array(q.q.tuple()), np.array(q.u.tuple()), np.array(q.v.tuple())
        corners += [q0, q0 + u, q0 + v, q0 + u + v, q0 + 0.5 * u]
        planes += [q0 + rng.uniform() * u + rng.uniform() * v for _ in range(4)]
    for i in range(sc.n_spheres):
        s = sc.spheres[i]
        c, r = np.array(s.center.tuple()), abs(s.radius)
        dirs = rng.normal(size=(4, 3))
        dirs /= np.linalg.norm(dirs, axis=1, keepdims=True)
        planes += list(c + r * dirs)          # points on the surface (origins there)
        corners += list(c + r * dirs)         # and as targets: rays that graze or hit the surface point
    corners = np.array(corners) if corners else np.zeros((1, 3))
    planes = np.array(planes) if planes else np.full((1, 3), 278.0)
    m = 6000
    o = rng.uniform(-50.0, 600.0, size=(m, 3))
    kind = rng.randint(0, 6, size=m)
    d = rng.normal(size=(m, 3))
    axis = np.eye(3)[rng.randint(0, 3, size=m)] * rng.choice([-1.0, 1.0], size=(m, 1))
    d = np.where((kind == 0)[:, None], axis, d)                                                   # along an axis
    d = np.where((kind == 1)[:, None], d * np.array([1.0, 1.0, 0.0])[rng.permutation(3)], d)      # in a coordinate plane
    tiny = d.copy()
    tiny[np.arange(m), rng.randint(0, 3, size=m)] = rng.choice([1e-30, -1e-30, -0.0, 1e-300], size=m)
    d = np.where((kind == 2)[:, None], tiny, d)                                                   # a vanishing component
    tgt = corners[rng.randint(0, len(corners), size=m)]
    d = np.where((kind == 3)[:, None], tgt - o, d)                                                # aimed at a corner / edge midpoint / surface point
    on = planes[rng.randint(0, len(planes), size=m)]
    o = np.where((kind == 4)[:, None], on, o)                                                     # starting on a surface
    o = np.where((kind == 5)[:, None], np.round(o), o)                                            # integer origins and ...
    d = np.where((kind == 5)[:, None], np.round(3 * d) + (np.abs(np.round(3 * d)).sum(axis=1, keepdims=True) == 0), d)  # ... small-integer directions
    res = ri.resident(0)
    pg, tg = res.trace_rays(o, d)
    po, to = oracle.trace_rays(ri.scene, o, d)
    differ = (pg != po) | (tg != to)
    n_rays += m
    n_differ += int(differ.sum())
    # zero-area triangles among the elements: |e1 x e2| vanishes against |e1| |e2|
    flat = np.zeros(sc.n_elements + 1, dtype=bool)
    for e in range(sc.n_elements):
        er = sc.elements[e]
        if er.kind == abi.RT_PRIM_TRIANGLE:
            t = sc.triangles[er.index]
            e1, e2 = np.array(t.v0v1.tuple()), np.array(t.v0v2.tuple())
            flat[e] = np.linalg.norm(np.cross(e1, e2)) <= 1e-10 * np.linalg.norm(e1) * np.linalg.norm(e2)
    noise_hit = differ & (po >= 0) & flat[np.maximum(po, 0)]
    n_noise += int(noise_hit.sum())
    wrong = differ & ~noise_hit & ((pg < 0) | ((po >= 0) & (tg > to)))
    if wrong.any():
        bad.append(seed)
        i = np.flatnonzero(wrong)[:5]
        print("WRONG seed", seed, int(wrong.sum()), "kinds", kind[i], "device", pg[i], tg[i], "oracle", po[i], to[i], flush=True)
    ri.close()
print(f"seeds {first}..{last - 1}: {len(bad)} with a farther or lost hit {bad}; {n_differ} of {n_rays} rays differ: {n_noise} are hits the reference reports on zero-area triangles, the others end nearer than the reference's (hits it loses at box corners); {time.time() - t0:.0f} s", flush=True)
sys.exit(1 if bad else 0)
