"""Traversal stacks of trees too high for on-chip memory: the binary order-free walks keep a lane's newest 16 stack entries in
LDS and move blocks of 8 older ones to device memory and back (RT_HIP_SPILL_ABOVE sets the tree height above which upload
arranges that; the default is 22).  Which memory holds an entry cannot change a hit, an image or a count."""
import os

import numpy as np
import pytest

from helpers import knot_scene, random_rays
from oracle import oracle
from raytracer_amd import RenderIntegrator, abi, scenes

pytestmark = pytest.mark.gpu


def _upload_both_ways(monkeypatch, ri, o, d, cfg):
    """(hits, t, linear image, rays, block moves) with every stack in LDS and with the overflow stack forced."""
    from raytracer_amd.integrator import ResidentScene
    results = []
    for above in ("64", "16"):
        monkeypatch.setenv("RT_HIP_SPILL_ABOVE", above)
        res = ResidentScene(ri.scene, 0)
        p, t = res.trace_rays(o, d)
        out, st = res.render(cfg)
        counted = cfg.copy()
        counted.flags = abi.RT_FLAG_COUNT_VISITS
        out_c, sc = res.render(counted)
        assert np.array_equal(out["linear"], out_c["linear"])
        results.append((p, t, out["linear"], int(st.rays), int(sc.stack_block_moves)))
        res.close()
    return results


def test_overflow_stack_on_a_mesh_gives_the_same_hits_and_image(monkeypatch, tmp_scene_dir):
    ri = knot_scene(tmp_scene_dir, 20000, 40, 40, 8, 2, 2)
    assert ri.bvh_depth() > 16
    o, d = random_rays(150_000, seed=41)
    (p0, t0, l0, r0, m0), (p1, t1, l1, r1, m1) = _upload_both_ways(monkeypatch, ri, o, d, ri.config.copy())
    assert m0 == 0
    assert np.array_equal(p0, p1) and np.array_equal(t0, t1) and np.array_equal(l0, l1) and r0 == r1
    po, to = oracle.trace_rays(ri.scene, o, d)
    assert np.array_equal(p1, po) and np.array_equal(t1, to)


def _stacked_sheets(path, n):
    """n large parallel triangles 0.003 apart, all facing the camera: a ray along z passes through every box of their subtree,
    so on its way down to the first sheet it pushes one entry per level."""
    z = np.linspace(100.0, 500.0, n)
    v = np.empty((n, 3, 3))
    v[:, 0] = np.stack([np.full(n, 60.0), np.full(n, 60.0), z], axis=1)
    v[:, 1] = np.stack([np.full(n, 60.0), np.full(n, 500.0), z], axis=1)
    v[:, 2] = np.stack([np.full(n, 500.0), np.full(n, 60.0), z], axis=1)
    e1, e2 = v[0, 1] - v[0, 0], v[0, 2] - v[0, 0]
    assert np.dot(e1, np.cross(np.array([0.0, 0.0, 1.0]), e2)) > 0.0  # triangle.rs:76-117 culls det < EPSILON: front face towards -z
    faces = np.arange(3 * n).reshape(n, 3)
    scenes.write_obj(path, v.reshape(-1, 3), faces)


def test_blocks_of_stack_entries_really_move_and_change_nothing(monkeypatch, tmp_scene_dir):
    n = 1 << 17
    os.makedirs(tmp_scene_dir, exist_ok=True)
    obj = os.path.join(tmp_scene_dir, "sheets.obj")
    _stacked_sheets(obj, n)
    els = [scenes.obj_element(obj, scenes.lambertian(0.6, 0.6, 0.6))] + scenes.cornell_elements()
    cfg = scenes.config(16, 16, 4, 2, 2, max_depth=3)
    ri = RenderIntegrator.new({"camera": scenes.camera(), "elements": els}, cfg).prepare()
    assert 16 < ri.bvh_depth() <= 64
    rng = np.random.RandomState(3)
    m = 20_000
    o = np.stack([rng.uniform(80.0, 300.0, m), rng.uniform(80.0, 300.0, m), rng.uniform(-200.0, 90.0, m)], axis=1)
    d = np.stack([rng.normal(0.0, 0.02, m), rng.normal(0.0, 0.02, m), np.ones(m)], axis=1)
    # a few from behind, through every sheet's back: no hit among them, every box entered
    o[:50, 2] = 520.0
    d[:50, 2] = -1.0
    (p0, t0, l0, r0, m0), (p1, t1, l1, r1, m1) = _upload_both_ways(monkeypatch, ri, o, d, ri.config.copy())
    assert m0 == 0 and m1 > 0, (m0, m1)
    assert np.array_equal(p0, p1) and np.array_equal(t0, t1) and np.array_equal(l0, l1) and r0 == r1
    # The oracle enters every box along a ray's whole line (aabb.rs:187-204 ignores the interval): 260 000 of them per ray here, so
    # it gets the first 1 500 rays.  The sheets are flat boxes met head-on, where the reference's f64 slab test can lose one by a
    # last bit and report a sheet further on; the device's boxes are conservative: it may end nearer, never farther.
    k = 1500
    po, to = oracle.trace_rays(ri.scene, o[:k], d[:k])
    differ = (p1[:k] != po) | (t1[:k] != to)
    assert differ.mean() < 0.02, differ.mean()
    assert np.all(p1[:k][differ] >= 0) and np.all((po[differ] < 0) | (t1[:k][differ] <= to[differ]))


@pytest.mark.parametrize("spill_above", ["64", "16"])
def test_walk_on_the_f32_node_records(monkeypatch, tmp_scene_dir, spill_above):
    """Scenes whose boxes the 16-bit grid would inflate too much are walked on the 64-byte f32 records (RT_HIP_NO_GRID forces it):
    same hits and image as on the grid records and as the oracle's, with the whole stack in LDS and with the overflow stack."""
    from raytracer_amd.integrator import ResidentScene
    ri = knot_scene(tmp_scene_dir, 20000, 40, 40, 8, 2, 2)
    o, d = random_rays(150_000, seed=43)
    grid = ri.resident(0)
    p0, t0 = grid.trace_rays(o, d)
    out0, st0 = grid.render(ri.config.copy())
    monkeypatch.setenv("RT_HIP_NO_GRID", "1")
    monkeypatch.setenv("RT_HIP_SPILL_ABOVE", spill_above)
    res = ResidentScene(ri.scene, 0)
    p1, t1 = res.trace_rays(o, d)
    out1, st1 = res.render(ri.config.copy())
    res.close()
    assert np.array_equal(p0, p1) and np.array_equal(t0, t1)
    assert np.array_equal(out0["linear"], out1["linear"]) and st0.rays == st1.rays
    po, to = oracle.trace_rays(ri.scene, o, d)
    assert np.array_equal(p1, po) and np.array_equal(t1, to)
