"""rt_hip_scene_build: the BVH built on the device from the primitives alone -- by default the reference's own top-down
build run level by level on the GPU (same tree as the host's), for comparison a linear BVH over Morton codes or a tree by
locally-ordered clustering (RT_HIP_BUILD=0 / 1) -- opt-in beside the default path that takes the host's reference-order tree.

Checked against the CPU oracle walking the DOWNLOADED device tree (rt_hip_scene_download_tree): hit distances and
element ids exact, ties included, since ties follow the depth-first leaf order of the tree that is walked.  Against
the host-built tree the hit distances are still exact; ids may differ only among primitives hit at exactly the same t."""
import ctypes as C

import numpy as np
import pytest

from raytracer_amd import RenderIntegrator, abi, scenes
from raytracer_amd.integrator import ResidentScene
from oracle import oracle

from helpers import c1, knot_scene, noise_gate, random_rays, teapot_scene

pytestmark = pytest.mark.gpu


def _scene_with_nodes(ri, nodes):
    """A copy of the prepared scene whose tree is `nodes` (for the oracle)."""
    s = abi.RtScene()
    C.memmove(C.byref(s), ri.scene, C.sizeof(abi.RtScene))
    s.nodes = C.cast(nodes, C.POINTER(abi.RtNode))
    s.n_nodes = len(nodes)
    return s


def _check_tree(ri, nodes):
    sc = ri.scene.contents
    n = sc.n_elements
    assert len(nodes) == 2 * n - 1
    leaves = [nd.element_index for nd in nodes if nd.is_leaf]
    assert sorted(leaves) == list(range(n))                      # every element under exactly one leaf
    internal = [i for i, nd in enumerate(nodes) if not nd.is_leaf]
    assert len(internal) == n - 1
    seen = np.zeros(len(nodes), dtype=np.int32)
    for i in internal:
        seen[nodes[i].child_l_index] += 1
        seen[nodes[i].child_r_index] += 1
    assert seen[0] == 0 and (seen[1:] == 1).all()                # a tree rooted at node 0
    # child boxes of leaf children contain the element's true bounds
    for i in internal[:: max(1, len(internal) // 500)]:
        nd = nodes[i]
        for child, mn, mx in ((nd.child_l_index, nd.child_l_min, nd.child_l_max), (nd.child_r_index, nd.child_r_min, nd.child_r_max)):
            if nodes[child].is_leaf:
                er = sc.elements[nodes[child].element_index]
                if er.kind == abi.RT_PRIM_TRIANGLE:
                    t = sc.triangles[er.index]
                    v0 = np.array(t.v0.tuple())
                    pts = [v0, v0 + np.array(t.v0v1.tuple()), v0 + np.array(t.v0v2.tuple())]
                elif er.kind == abi.RT_PRIM_QUAD:
                    q = sc.quads[er.index]
                    q0, u, v = np.array(q.q.tuple()), np.array(q.u.tuple()), np.array(q.v.tuple())
                    pts = [q0, q0 + u, q0 + v, q0 + u + v]
                else:
                    s = sc.spheres[er.index]
                    c, r = np.array(s.center.tuple()), abs(s.radius)
                    pts = [c - r, c + r]
                pts = np.array(pts)
                assert (pts.min(0) >= np.array(mn[:])).all() and (pts.max(0) <= np.array(mx[:])).all()


def _preorder(nodes, n_nodes):
    """The tree as its preorder sequence (-1 for an internal node, else the leaf's element): equal sequences, equal trees."""
    seq, stack = [], [0]
    while stack:
        nd = nodes[stack.pop()]
        if nd.is_leaf:
            seq.append(nd.element_index)
        else:
            seq.append(-1)
            stack.append(nd.child_r_index)
            stack.append(nd.child_l_index)
    assert len(seq) == n_nodes
    return seq


@pytest.mark.parametrize("split", ["SAH", "Mid"])
@pytest.mark.parametrize("which", ["c1", "knot", "teapot", "soup", "tilted", "pair"])
def test_top_down_device_build_makes_the_hosts_tree(which, split, tmp_scene_dir, monkeypatch):
    """The default method of rt_hip_scene_build restates BVHNode_SAH::build (bvh_node_sah.rs:256-461) level by level on the
    device: same topology and the same depth-first leaf order as the host's restatement of the reference's recursion, for its
    bucketed surface-area split and for its midpoint split; on trees whose boxes bound their contents therefore the same
    hits, ids included, as the uploaded tree gives."""
    from helpers import soup_scene
    if which == "c1":
        ri = c1(width=32, height=32, batch=2, split=split)
    elif which == "knot":
        ri = knot_scene(tmp_scene_dir, 20000, 32, 32, 2, 1, 1, split=split)
    elif which == "teapot":
        ri = teapot_scene(32, 32, 2, 1, 1, split=split)
    elif which == "soup":
        ri = soup_scene(3000, 32, 32, 2, 1, 1, split=split)
    elif which == "tilted":
        from test_gpu_parity import _tilted_quad_scene
        ri = _tilted_quad_scene(split=split)
    else:
        ri = RenderIntegrator.new({"camera": scenes.camera(), "elements": [scenes.sphere((278, 278, 200), 120, scenes.lambertian(.5, .5, .5)),
                                                                         scenes.sphere((100, 278, 200), 50, scenes.lambertian(.5, .5, .5))]},
                                  scenes.config(16, 16, 2, 1, 1, split=split)).prepare()
    monkeypatch.delenv("RT_HIP_BUILD", raising=False)             # the scene names the rule (RtScene.split_method)
    assert ri.scene.contents.split_method == (abi.RT_SPLIT_SAH if split == "SAH" else abi.RT_SPLIT_MID)
    dev = ResidentScene(ri.scene, 0, build_on_device=True)
    nodes = dev.download_tree()
    sc = ri.scene.contents
    assert _preorder(nodes, len(nodes)) == _preorder(sc.nodes, sc.n_nodes)
    if which != "tilted":  # (there the uploaded tree follows the reference's visit rule, the device-built one holds true bounds)
        host = ri.resident(0)
        o, d = random_rays(100_000, seed=75)
        ph, th = host.trace_rays(o, d)
        pd, td = dev.trace_rays(o, d)
        assert np.array_equal(th, td) and np.array_equal(ph, pd)
    dev.close()


@pytest.mark.parametrize("split", ["SAH", "Mid"])
def test_top_down_device_build_on_coinciding_centroids_and_tiny_scenes(split, monkeypatch):
    """Elements whose centroids coincide are split into halves by position (bvh_node_sah.rs:351-362), which makes the ORDER the
    earlier bucket sorts left behind visible in the tree: 37 copies of one triangle, 5 of another one with the same centroid
    and a few spheres must come out as the host's tree, and a ray through the copies must name the same copy as on the
    host's tree (the last one in leaf order, interval.rs:31-33).  Scenes of 3, 4 and 5 elements: the smallest recursions."""
    monkeypatch.delenv("RT_HIP_BUILD", raising=False)
    tri = lambda a, b, c: {"JSONTriangle": {"v0": scenes._v(*a), "v1": scenes._v(*b), "v2": scenes._v(*c), "material": scenes.lambertian(.5, .5, .5)}}
    copies = [tri((100, 100, 300), (300, 100, 300), (200, 300, 300)) for _ in range(37)]
    # same bounding box, hence the same centroid, other triangle
    copies += [tri((100, 100, 300), (300, 300, 300), (200, 100, 300)) for _ in range(5)]
    spheres = [scenes.sphere((60 + 90 * k, 420, 250), 30, scenes.lambertian(.5, .5, .5)) for k in range(5)]
    for elements in (copies + spheres, spheres[:3], spheres[:4], spheres[:5], copies[:3]):
        ri = RenderIntegrator.new({"camera": scenes.camera(), "elements": elements}, scenes.config(16, 16, 2, 1, 1, split=split)).prepare()
        dev = ResidentScene(ri.scene, 0, build_on_device=True)
        nodes = dev.download_tree()
        sc = ri.scene.contents
        assert _preorder(nodes, len(nodes)) == _preorder(sc.nodes, sc.n_nodes)
        rng = np.random.RandomState(5)
        o = np.tile(np.array([200.0, 170.0, -400.0]), (2000, 1)) + rng.normal(scale=40.0, size=(2000, 3)) * np.array([1, 1, 0])
        d = np.tile(np.array([0.0, 0.0, 1.0]), (2000, 1)) + rng.normal(scale=0.05, size=(2000, 3))
        ph, th = ri.resident(0).trace_rays(o, d)
        pd, td = dev.trace_rays(o, d)
        assert np.array_equal(th, td) and np.array_equal(ph, pd)
        if len(elements) > 40:
            assert (ph >= 0).mean() > 0.3                                 # the stack of copies is hit (and one copy of it wins every tie)
        dev.close()
        ri.close()


@pytest.mark.parametrize("method", [0, 1, 2, 3])
@pytest.mark.parametrize("which", ["c1", "knot", "tilted", "single"])
def test_device_built_tree_gives_the_oracles_hits(which, method, tmp_scene_dir, monkeypatch):
    monkeypatch.setenv("RT_HIP_BUILD", str(method))
    if which == "c1":
        ri = c1(width=48, height=48, batch=4)
    elif which == "knot":
        ri = knot_scene(tmp_scene_dir, 20000, 64, 64, 4, 1, 1)
    elif which == "tilted":
        from test_gpu_parity import _tilted_quad_scene
        ri = _tilted_quad_scene()
    else:
        ri = RenderIntegrator.new({"camera": scenes.camera(), "elements": [scenes.sphere((278, 278, 200), 120, scenes.lambertian(.5, .5, .5))]},
                                  scenes.config(16, 16, 2, 1, 1)).prepare()
    dev = ResidentScene(ri.scene, 0, build_on_device=True)
    nodes = dev.download_tree()
    if which != "single":
        _check_tree(ri, nodes)
    twin = _scene_with_nodes(ri, nodes)
    o, d = random_rays(150_000, seed=71)
    pg, tg = dev.trace_rays(o, d)
    po, to = oracle.trace_rays(C.byref(twin), o, d)
    assert (po >= 0).mean() > (0.05 if which == "single" else 0.3)
    assert np.array_equal(tg, to)
    assert np.array_equal(pg, po)
    cfg = ri.config.copy()
    pg, tg = dev.primary_hits(cfg)
    po, to = oracle.primary_hits(C.byref(twin), cfg)
    assert np.array_equal(tg, to) and np.array_equal(pg, po)
    dev.close()


def test_device_built_tree_with_wide_nodes_and_padded_triangle_records(monkeypatch, tmp_scene_dir):
    """What scenes beyond the caches get (forced here by RT_HIP_WIDE / RT_HIP_TRI_PAD): the device-built binary records collapsed
    to four-wide ones, triangle records one per 128-byte line.  Same hits, same image as on the plain device-built tree."""
    ri = knot_scene(tmp_scene_dir, 20000, 48, 48, 8, 2, 2)
    o, d = random_rays(150_000, seed=73)
    results = []
    for forced in ("0", "1"):
        monkeypatch.setenv("RT_HIP_WIDE", forced)
        monkeypatch.setenv("RT_HIP_TRI_PAD", forced)
        dev = ResidentScene(ri.scene, 0, build_on_device=True)
        p, t = dev.trace_rays(o, d)
        out, st = dev.render(ri.config.copy())
        counted = ri.config.copy()
        counted.flags = abi.RT_FLAG_COUNT_VISITS
        _, sc = dev.render(counted)
        nodes = dev.download_tree()
        results.append((p, t, out["linear"], int(st.rays), sc.node_visits / sc.rays_traced, nodes))
        dev.close()
    (p0, t0, l0, r0, v0, n0), (p1, t1, l1, r1, v1, n1) = results
    assert np.array_equal(p0, p1) and np.array_equal(t0, t1) and np.array_equal(l0, l1) and r0 == r1
    assert v1 < 0.7 * v0                                          # the wide records really were walked
    assert bytes(n0) == bytes(n1)                                 # the padded records download as the same tree
    twin = _scene_with_nodes(ri, n1)
    po, to = oracle.trace_rays(C.byref(twin), o, d)
    assert np.array_equal(p1, po) and np.array_equal(t1, to)


def test_device_built_tree_against_the_host_built_one(tmp_scene_dir):
    """Same closest hits as with the reference-order tree: t exact; ids differ only where two primitives are hit at
    exactly the same t (the duplicated wall triangles of the reference's scene).  Images agree within noise."""
    ri = knot_scene(tmp_scene_dir, 20000, 96, 96, 16, 4, 1)
    host = ri.resident(0)
    dev = ResidentScene(ri.scene, 0, build_on_device=True)
    o, d = random_rays(200_000, seed=72)
    ph, th = host.trace_rays(o, d)
    pd, td = dev.trace_rays(o, d)
    assert np.array_equal(th, td)
    differ = ph != pd
    assert differ.mean() < 0.2                                    # only rays that end on a doubled wall triangle
    sc = ri.scene.contents
    for i in np.flatnonzero(differ)[:200]:                       # each of them is a genuine hit of the other element at the same t
        e = sc.elements[int(pd[i])]
        assert e.kind == abi.RT_PRIM_TRIANGLE
        assert oracle.triangle_hit(sc.triangles[e.index], o[i], d[i], 0.0001, 1.7976931348623157e308) == td[i]
    cfg = ri.config.copy()
    a, sa = host.render(cfg)
    b, sb = dev.render(cfg)
    other = cfg.copy()
    other.seed = 99
    c, _ = host.render(other)
    # the device tree's image differs from the host tree's by less than a second seed does (duplicates share a material)
    noise_gate(a["linear"], c["linear"], b["linear"], 0.5)
    assert abs(int(sa.rays) - int(sb.rays)) <= 1e-3 * sa.rays
    dev.close()


def test_device_build_on_the_reference_teapot_scene():
    ri = teapot_scene(128, 128, 8, 2, 2)
    dev = ResidentScene(ri.scene, 0, build_on_device=True)
    nodes = dev.download_tree()
    _check_tree(ri, nodes)
    twin = _scene_with_nodes(ri, nodes)
    cfg = ri.config.copy()
    pg, tg = dev.primary_hits(cfg)
    po, to = oracle.primary_hits_threads(C.byref(twin), cfg, 8)
    assert np.array_equal(pg, po) and np.array_equal(tg, to)
    out_g, st = dev.render(cfg)
    out_o, _ = oracle.render(C.byref(twin), cfg, 16)
    from helpers import assert_image_parity
    assert_image_parity(out_g, out_o, "teapot, device-built tree", dev, cfg)
    dev.close()


def test_device_build_rejects_what_it_cannot_build():
    ri = c1(width=8, height=8, batch=1)
    sc = ri.scene.contents
    lib = abi.hip_lib()
    h = C.c_void_p()
    s = abi.RtScene()
    C.memmove(C.byref(s), C.byref(sc), C.sizeof(abi.RtScene))
    s.n_elements = 0
    assert lib.rt_hip_scene_build(C.byref(s), 0, C.byref(h)) == abi.RT_ERR_BAD_ARG
    C.memmove(C.byref(s), C.byref(sc), C.sizeof(abi.RtScene))
    sph = (abi.RtSphere * sc.n_spheres)()
    C.memmove(sph, sc.spheres, C.sizeof(sph))
    sph[0].radius = float("inf")
    s.spheres = sph
    assert lib.rt_hip_scene_build(C.byref(s), 0, C.byref(h)) == abi.RT_ERR_UNSUPPORTED and b"finite" in lib.rt_hip_last_error()
    assert lib.rt_hip_scene_build(C.byref(sc), 99, C.byref(h)) == abi.RT_ERR_BAD_ARG


def test_host_integrator_can_leave_the_build_to_the_device(tmp_scene_dir):
    """rt_host_set_device_build: render() of the host library with the tree built on the GPU gives the image of the default
    path (the same tree; the Cornell scene's axis-aligned quads are bounded by the reference's boxes, so both walks agree)."""
    import os
    from helpers import knot_scene
    a = knot_scene(tmp_scene_dir, 20000, 48, 48, 8, 3, 1)
    a.render(0)
    obj = os.path.join(tmp_scene_dir, "knot_20000.obj")
    b = RenderIntegrator.new(scenes.scene_c3(obj), scenes.config(48, 48, 8, 3, 1)).set_device_build()
    b.prepare()
    assert b.bvh_depth() == -1 and b.scene.contents.n_nodes == 0
    st = b.render(0)
    assert st.rays > 0
    assert np.array_equal(a.pixels, b.pixels) and np.array_equal(a.sample_pixels, b.sample_pixels)
    with pytest.raises(abi.RtError):
        b.set_device_build(False)                                  # too late: the integrator is prepared
