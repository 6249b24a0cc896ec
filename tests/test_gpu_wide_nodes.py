"""The order-free walk on wide nodes (up to four children per 64-byte record, made at upload for very large trees,
RT_HIP_WIDE=1 forces them): same hits, same image as on the binary records and as the oracle's."""
import numpy as np
import pytest

from helpers import knot_scene, random_rays
from oracle import oracle
from raytracer_amd import scenes

pytestmark = pytest.mark.gpu


def _both(monkeypatch, ri, o, d, cfg):
    from raytracer_amd import abi
    from raytracer_amd.integrator import ResidentScene
    results = []
    for wide in ("0", "1"):
        monkeypatch.setenv("RT_HIP_WIDE", wide)
        res = ResidentScene(ri.scene, 0)
        p, t = res.trace_rays(o, d)
        out, st = res.render(cfg)
        counted = cfg.copy()
        counted.flags = abi.RT_FLAG_COUNT_VISITS
        _, sc = res.render(counted)
        results.append((p, t, out["linear"], int(st.rays), sc.node_visits / sc.rays_traced))
        res.close()
    return results


@pytest.mark.parametrize("triangles", [300, 20000])
def test_wide_nodes_give_the_binary_walks_hits_and_image(monkeypatch, tmp_scene_dir, triangles):
    """300 triangles: the whole stack of the wide walk fits the LDS; 20 000: its deepest levels spill."""
    ri = knot_scene(tmp_scene_dir, triangles, 40, 40, 8, 2, 2)
    o, d = random_rays(150_000, seed=37)
    (p0, t0, l0, r0, v0), (p1, t1, l1, r1, v1) = _both(monkeypatch, ri, o, d, ri.config.copy())
    assert np.array_equal(p0, p1) and np.array_equal(t0, t1) and np.array_equal(l0, l1) and r0 == r1
    assert v1 < 0.7 * v0  # the wide records really were walked: about half as many node steps
    po, to = oracle.trace_rays(ri.scene, o, d)
    assert np.array_equal(p1, po) and np.array_equal(t1, to)


def test_wide_nodes_on_a_soup_with_every_primitive_kind(monkeypatch, tmp_scene_dir):
    """Overlapping boxes, glass and metal triangles, the quads of the room among the leaves, a thin lens; some rays along the axes."""
    from raytracer_amd import RenderIntegrator
    sp, cp = scenes.materialize_soup(tmp_scene_dir, 20000, 40, 40, batch=8, max_batches=2, min_batches=2)
    ri = RenderIntegrator.new_from_json(sp, cp).prepare()
    rng = np.random.RandomState(5)
    o = rng.uniform(5.0, 550.0, size=(100_000, 3))
    d = rng.normal(size=(100_000, 3))
    d[:2000] = np.eye(3)[rng.randint(0, 3, size=2000)] * rng.choice([-1.0, 1.0], size=(2000, 1))
    (p0, t0, l0, r0, v0), (p1, t1, l1, r1, v1) = _both(monkeypatch, ri, o, d, ri.config.copy())
    assert np.array_equal(p0, p1) and np.array_equal(t0, t1) and np.array_equal(l0, l1) and r0 == r1
    assert v1 < 0.7 * v0
    po, to = oracle.trace_rays(ri.scene, o, d)
    assert np.array_equal(p1, po) and np.array_equal(t1, to)


def _triangles_of(scene):
    """v0, v0v1, v0v2 of every triangle of an RtScene as (n, 3) arrays."""
    import ctypes as C
    s = scene.contents
    n = s.n_triangles
    raw = np.ctypeslib.as_array(C.cast(s.triangles, C.POINTER(C.c_double)), shape=(n, C.sizeof(s.triangles[0]) // 8))
    return raw[:, 0:3].copy(), raw[:, 3:6].copy(), raw[:, 6:9].copy()


@pytest.mark.parametrize("which", ["soup", "mesh"])
def test_rays_aimed_at_vertices_and_edges(monkeypatch, tmp_scene_dir, which):
    """Rays aimed at the places where the triangle test decides by its last bits: vertices, points on the edges, points a
    few ulps inside and outside them, from near and far, head-on and grazing.  The walk on wide nodes and the walk on
    binary ones must agree bit for bit, and with the oracle wherever the reference itself reaches the triangle."""
    from raytracer_amd import RenderIntegrator
    from raytracer_amd.integrator import ResidentScene
    if which == "soup":
        sp, cp = scenes.materialize_soup(tmp_scene_dir, 20000, 8, 8, batch=4, max_batches=1, min_batches=1)
        ri = RenderIntegrator.new_from_json(sp, cp).prepare()
    else:
        ri = knot_scene(tmp_scene_dir, 20000, 8, 8, 4, 1, 1)
    v0, e1, e2 = _triangles_of(ri.scene)
    rng = np.random.RandomState(11)
    m = 60_000
    pick = rng.randint(0, len(v0), size=m)
    # barycentric targets: vertices, edge points, points just inside / outside an edge, interior points
    kind = rng.randint(0, 6, size=m)
    a = rng.uniform(size=m)
    b = rng.uniform(size=m) * (1.0 - a)
    eps = rng.choice([0.0, 1e-16, -1e-16, 1e-12, -1e-12, 1e-7, -1e-7, 3e-5, -3e-5], size=m)
    u = np.where(kind == 0, np.round(a), np.where(kind == 1, a, np.where(kind == 2, eps, np.where(kind == 3, a, a))))
    v = np.where(kind == 0, np.round(b) * (1 - np.round(a)), np.where(kind == 1, eps, np.where(kind == 2, b, np.where(kind == 3, 1.0 - a + eps, b))))
    target = v0[pick] + e1[pick] * u[:, None] + e2[pick] * v[:, None]
    normal = np.cross(e1[pick], e2[pick])
    normal /= np.maximum(np.linalg.norm(normal, axis=1, keepdims=True), 1e-300)
    # origins: in front of the triangle (it is culled from behind), at distances from 1e-3 to 1e3, head-on to grazing
    dist = 10.0 ** rng.uniform(-3.0, 3.0, size=m)
    tilt = rng.normal(size=(m, 3)) * rng.choice([0.0, 0.3, 3.0, 300.0], size=(m, 1))
    side = normal + np.cross(normal, tilt)
    side /= np.maximum(np.linalg.norm(side, axis=1, keepdims=True), 1e-300)
    o = target + side * dist[:, None] * np.where(kind == 5, -1.0, 1.0)[:, None]
    d = target - o
    results = []
    for wide in ("0", "1"):
        monkeypatch.setenv("RT_HIP_WIDE", wide)
        res = ResidentScene(ri.scene, 0)
        results.append(res.trace_rays(o, d))
        res.close()
    (p0, t0), (p1, t1) = results
    assert np.array_equal(p0, p1) and np.array_equal(t0, t1)
    assert np.mean(p1 >= 0) > 0.5  # the rays do hit things
    # Against the oracle the same holds except where the REFERENCE loses the hit: a ray through a vertex passes through a
    # corner of the leaf's box, where the reference's f64 slab test (aabb.rs:187-204) fails by a last bit and the triangle is
    # never tested; the device's boxes are conservative and reach it.  Such rays end nearer on the device, never farther.
    po, to = oracle.trace_rays(ri.scene, o, d)
    differ = (p1 != po) | (t1 != to)
    assert differ.mean() < 0.02, differ.mean()
    # (at a vertex shared by several triangles of a mesh all of them are hit at the same t and the last one in leaf order
    # wins: the reference, having lost some of them, may name another one at that same t)
    assert np.all(p1[differ] >= 0) and np.all((po[differ] < 0) | (t1[differ] <= to[differ]))
    assert not np.any(differ & (kind != 0))  # only the rays aimed at vertices
