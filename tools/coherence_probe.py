#!/usr/bin/env python3
"""What ray ORDER is worth to the closest-hit kernel.

Builds bounce-1-like and bounce-2-like ray sets on the headline scene (origins on the surfaces the camera sees, directions
cosine-distributed about the normal there; the second set starts where the first one lands), then traces every set several
times through rt_hip_trace_rays in different orders of the SAME rays: as the paths come (pixel by pixel), shuffled, sorted by
origin cell, by direction bin, by both.  Run it under `rocprofv3 --kernel-trace`: the durations of the large k_extend launches,
in order, belong to the lines this script prints (tools/coherence_probe_report.py joins the two).

Results cannot depend on the order (closest hit, ties by leaf rank): the script checks that too.
"""
import os
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from raytracer_amd import RenderIntegrator, abi, scenes  # noqa: E402

rng = np.random.default_rng(11)
workdir = os.path.join(tempfile.gettempdir(), f"rt_bench_{os.getuid()}_0")
sp, cp = scenes.materialize_c3(workdir, 100_000, 1200, 1200)
ri = RenderIntegrator.new_from_json(sp, cp).prepare()
res = ri.resident(0)
sc = ri.scene.contents


def struct_array(ptr, n, words):
    return np.ctypeslib.as_array((abi.C.c_double * (n * words)).from_address(abi.C.addressof(ptr.contents))).reshape(n, words) if n else np.zeros((0, words))


tri = struct_array(sc.triangles, sc.n_triangles, 13)
quad = struct_array(sc.quads, sc.n_quads, 18)
sph = struct_array(sc.spheres, sc.n_spheres, 5)
elem = np.ctypeslib.as_array((abi.C.c_int32 * (2 * sc.n_elements)).from_address(abi.C.addressof(sc.elements.contents))).reshape(-1, 2)


def normals_at(prim, point):
    kind, idx = elem[prim, 0], elem[prim, 1]
    n = np.zeros((len(prim), 3))
    m = kind == abi.RT_PRIM_TRIANGLE
    n[m] = tri[idx[m], 9:12]
    m = kind == abi.RT_PRIM_QUAD
    n[m] = quad[idx[m], 9:12]
    m = kind == abi.RT_PRIM_SPHERE
    n[m] = (point[m] - sph[idx[m], 0:3]) / sph[idx[m], 3:4]
    return n / np.linalg.norm(n, axis=1, keepdims=True)


def cosine_about(n):
    r1, r2 = rng.random(len(n)), rng.random(len(n))
    phi = 2 * np.pi * r1
    local = np.stack([np.cos(phi) * np.sqrt(r2), np.sin(phi) * np.sqrt(r2), np.sqrt(1 - r2)], axis=1)
    a = np.where((np.abs(n[:, 0]) > 0.9)[:, None], np.array([0.0, 1.0, 0.0]), np.array([1.0, 0.0, 0.0]))
    v = np.cross(n, a)
    v /= np.linalg.norm(v, axis=1, keepdims=True)
    u = np.cross(n, v)
    return u * local[:, 0:1] + v * local[:, 1:2] + n * local[:, 2:3]


def next_rays(o, d, prim, t):
    hit = prim >= 0
    dn = d[hit] / np.linalg.norm(d[hit], axis=1, keepdims=True)
    p = o[hit] + dn * t[hit][:, None]
    n = normals_at(prim[hit], p)
    n = np.where((np.sum(n * dn, axis=1) > 0)[:, None], -n, n)  # leave on the side the ray came from
    return p, cosine_about(n)


# primary rays: W x H pixels, SPP per pixel, pixel by pixel (the order of the device's own queues)
W = H = int(os.environ.get("PROBE_SIDE", "600"))
SPP = int(os.environ.get("PROBE_SPP", "16"))
cam = sc.camera
c = np.array(cam.camera_center.tuple())
p00, du, dv = (np.array(getattr(cam, k).tuple()) for k in ("pixel00_loc", "pixel_delta_u", "pixel_delta_v"))
ys, xs = np.meshgrid(np.arange(H), np.arange(W), indexing="ij")
# 8 x 8 tiles, as the active list orders the pixels
order = np.lexsort((xs.ravel() % 8, ys.ravel() % 8, xs.ravel() // 8, ys.ravel() // 8))
px = np.repeat(xs.ravel()[order] * (1200 / W), SPP) + rng.random(W * H * SPP) * (1200 / W)
py = np.repeat(ys.ravel()[order] * (1200 / H), SPP) + rng.random(W * H * SPP) * (1200 / H)
d0 = p00 + du * px[:, None] + dv * py[:, None] - c
o0 = np.broadcast_to(c, d0.shape).copy()
print(f"MARK primary natural n={len(o0)}", flush=True)
prim0, t0 = res.trace_rays(o0, d0)
o1, d1 = next_rays(o0, d0, prim0, t0)
print(f"MARK bounce1 natural n={len(o1)}", flush=True)
prim1, t1 = res.trace_rays(o1, d1)
o2, d2 = next_rays(o1, d1, prim1, t1)

lo = np.minimum(o1.min(axis=0), o2.min(axis=0))
hi = np.maximum(o1.max(axis=0), o2.max(axis=0))


def morton(o, bits):
    q = np.clip(((o - lo) / (hi - lo) * (1 << bits)).astype(np.int64), 0, (1 << bits) - 1)
    code = np.zeros(len(o), dtype=np.int64)
    for b in range(bits):
        for a in range(3):
            code |= ((q[:, a] >> b) & 1) << (3 * b + a)
    return code


def octant(d):
    return (d[:, 0] < 0).astype(np.int64) | ((d[:, 1] < 0).astype(np.int64) << 1) | ((d[:, 2] < 0).astype(np.int64) << 2)


def cube_bin(d, k):
    """6 cube faces x k x k cells."""
    a = np.abs(d)
    face = np.argmax(a, axis=1)
    m = d[np.arange(len(d)), face]
    u = d[np.arange(len(d)), (face + 1) % 3] / np.abs(m)
    v = d[np.arange(len(d)), (face + 2) % 3] / np.abs(m)
    cu = np.clip(((u + 1) / 2 * k).astype(np.int64), 0, k - 1)
    cv = np.clip(((v + 1) / 2 * k).astype(np.int64), 0, k - 1)
    return ((face * 2 + (m < 0)) * k + cu) * k + cv


def orders(o, d):
    n = len(o)
    nat = np.arange(n)
    yield "natural", nat
    yield "shuffled", rng.permutation(n)
    yield "origin cell 4x2x4 (stable)", np.argsort((morton(o, 2) & ~np.int64(2)), kind="stable")
    yield "origin morton 2^15 cells", np.argsort(morton(o, 5), kind="stable")
    yield "octant (stable)", np.argsort(octant(d), kind="stable")
    yield "cube 6x2x2 (stable)", np.argsort(cube_bin(d, 2), kind="stable")
    yield "cube 6x4x4 (stable)", np.argsort(cube_bin(d, 4), kind="stable")
    yield "origin 64 cells, then octant", np.lexsort((octant(d), morton(o, 2)))
    yield "origin 512 cells, then octant", np.lexsort((octant(d), morton(o, 3)))
    yield "origin 512 cells, then cube 6x4x4", np.lexsort((cube_bin(d, 4), morton(o, 3)))
    yield "octant, then origin morton 2^15", np.lexsort((morton(o, 5), octant(d)))
    yield "cube 6x4x4, then origin morton 2^15", np.lexsort((morton(o, 5), cube_bin(d, 4)))
    # what a kernel could do per block of 4096 consecutive paths: sort inside the block only
    blk = nat // 4096
    yield "octant inside blocks of 4096", np.lexsort((octant(d), blk))
    yield "cube 6x2x2 inside blocks of 4096", np.lexsort((cube_bin(d, 2), blk))
    blk = nat // 32768
    yield "cube 6x4x4 inside blocks of 32768", np.lexsort((cube_bin(d, 4), blk))


for name, (o, d) in (("bounce1", (o1, d1)), ("bounce2", (o2, d2))):
    ref = None
    for label, perm in orders(o, d):
        print(f"MARK {name} {label} n={len(o)}", flush=True)
        prim, t = res.trace_rays(o[perm], d[perm])
        back = np.empty_like(perm)
        back[perm] = np.arange(len(perm))
        prim, t = prim[back], t[back]
        if ref is None:
            ref = (prim, t)
        else:
            assert np.array_equal(prim, ref[0]) and np.array_equal(t, ref[1]), "results depend on the order of the rays"
print("DONE", flush=True)
