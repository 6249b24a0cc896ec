"""Host side above the C ABI (C++: scene loading, element expansion, camera,
BVH build, PNG output) checked against an independent pure-Python/numpy
restatement (oracle/host_ref.py) and against the reference's own unit-test
known answers for the pieces it pins (aabb.rs:337-391)."""
import ctypes as C
import json
import os
import struct
import zlib

import numpy as np
import pytest

from raytracer_amd import RenderIntegrator, abi, scenes
from oracle import host_ref

from helpers import c1, knot_scene


def read_png(path):
    """Minimal 8-bit RGB PNG reader (filters 0-4)."""
    data = open(path, "rb").read()
    assert data[:8] == b"\x89PNG\r\n\x1a\n"
    pos, chunks = 8, []
    while pos < len(data):
        n, tag = struct.unpack(">I4s", data[pos:pos + 8])
        body = data[pos + 8:pos + 8 + n]
        crc = struct.unpack(">I", data[pos + 8 + n:pos + 12 + n])[0]
        assert zlib.crc32(tag + body) & 0xffffffff == crc
        chunks.append((tag, body))
        pos += 12 + n
    w, h, depth, ctype = struct.unpack(">IIBB", chunks[0][1][:10])
    assert (depth, ctype) == (8, 2)
    raw = zlib.decompress(b"".join(b for t, b in chunks if t == b"IDAT"))
    stride = w * 3
    out = np.zeros((h, stride), dtype=np.uint8)
    prev = np.zeros(stride, dtype=np.int32)
    for y in range(h):
        f = raw[y * (stride + 1)]
        line = np.frombuffer(raw[y * (stride + 1) + 1:(y + 1) * (stride + 1)], dtype=np.uint8).astype(np.int32)
        assert f == 0, "writer uses filter 0"
        out[y] = line
        prev = line
    return w, h, out.reshape(h, w, 3), dict((t, b) for t, b in chunks)


def _scene_arrays(ri):
    sc = ri.scene.contents
    nodes = [(n.is_leaf, n.element_index, n.child_l_index, n.child_r_index,
              tuple(n.child_l_min), tuple(n.child_l_max), tuple(n.child_r_min), tuple(n.child_r_max))
             for n in (sc.nodes[i] for i in range(sc.n_nodes))]
    return sc, nodes


@pytest.mark.parametrize("split", ["SAH", "Mid", None])
def test_bvh_matches_python_restatement_c1(split):
    cfg = scenes.config(32, 32, 4, 1, 1, split=split)
    scene = scenes.scene_c1()
    ri = RenderIntegrator.new(scene, cfg).prepare()
    sc, nodes = _scene_arrays(ri)
    ref = host_ref.build_scene(scene, cfg)
    assert sc.n_elements == len(ref["elements"]) == 28
    assert sc.n_nodes == 2 * sc.n_elements - 1 == len(ref["nodes"])
    assert nodes == ref["nodes"]
    for i, e in enumerate(ref["elements"]):
        mn, mx = ri.element_bbox(i)
        assert mn == tuple(e["bbox"][0]) and mx == tuple(e["bbox"][1]), i
        assert (sc.elements[i].kind, sc.elements[i].index) == (e["kind"], e["index"])
    assert [sc.attractors[i] for i in range(sc.n_attractors)] == ref["attractors"] == [27]


def test_bvh_matches_python_restatement_mesh(tmp_scene_dir):
    ri = knot_scene(tmp_scene_dir, 600, 16, 16, 1, 1, 1)
    sc, nodes = _scene_arrays(ri)
    obj = os.path.join(tmp_scene_dir, "knot_600.obj")
    ref = host_ref.build_scene(scenes.scene_c3(obj), scenes.config(16, 16, 1, 1, 1))
    assert sc.n_triangles == 600 + 24
    assert nodes == ref["nodes"]
    # triangles: v0, edges, unit normal
    for k in (0, 17, 599, 600, 623):
        t = sc.triangles[k]
        r = ref["triangles"][k]
        assert t.v0.tuple() == r["v0"] and t.v0v1.tuple() == r["v0v1"] and t.v0v2.tuple() == r["v0v2"] and t.normal.tuple() == r["normal"]


def test_camera_matches_python_restatement():
    for defocus in (0.0, 0.6):
        scene = scenes.scene_c1(defocus_angle=defocus)
        cfg = scenes.config(400, 300, 4, 1, 1)
        ri = RenderIntegrator.new(scene, cfg).prepare()
        cam = ri.scene.contents.camera
        ref = host_ref.make_camera(cfg, scene["camera"])
        for name in ("camera_center", "pixel00_loc", "pixel_delta_u", "pixel_delta_v", "defocus_disk_u", "defocus_disk_v"):
            assert getattr(cam, name).tuple() == ref[name], name
        assert cam.defocus_angle == defocus
    # with the shipped camera u = -X: image x runs against world X
    assert cam.pixel_delta_u.x < 0 and cam.pixel_delta_v.y < 0


def test_quad_precompute_and_bbox():
    ri = c1()
    sc = ri.scene.contents
    q = sc.quads[0]
    assert q.normal.tuple() == (0.0, 1.0, 0.0) or q.normal.tuple() == (0.0, -1.0, 0.0) or abs(abs(q.normal.y) - 1) < 1e-15
    assert q.area == 130.0 * 105.0
    assert q.attractor == 1
    mn, mx = ri.element_bbox(27)
    # flat in y: padded by 5e-5 each side (aabb.rs:122-152)
    assert mn == (213.0, 554.0 - 0.00005, 227.0) and mx == (343.0, 554.0 + 0.00005, 332.0)


def test_box_expands_to_twelve_front_facing_triangles():
    """cube.rs:67-80 winding: every triangle's normal points out of the box."""
    ri = c1()
    sc = ri.scene.contents
    tris = [sc.triangles[i] for i in range(12)]  # the box comes first among the triangles
    centre = np.mean([np.array(t.v0.tuple()) + (np.array(t.v0v1.tuple()) + np.array(t.v0v2.tuple())) / 3 for t in tris], axis=0)
    for t in tris:
        c = np.array(t.v0.tuple()) + (np.array(t.v0v1.tuple()) + np.array(t.v0v2.tuple())) / 3
        assert np.dot(np.array(t.normal.tuple()), c - centre) > 0


def test_mat4_times_point_known_answer_from_reference_tests():
    # mat4.rs:484-511 (test_multiply_vec3): Mat4 * Vec4(v, 1), the product the element transforms apply to every vertex
    m = [[1., 2., 3., 4.], [2., 3., 4., 5.], [3., 4., 5., 6.]]
    assert host_ref._apply(m, (1., 2., 3.)) == (1. * 1. + 2. * 2. + 3. * 3. + 4. * 1., 2. * 1. + 3. * 2. + 4. * 3. + 5. * 1.,
                                               3. * 1. + 4. * 2. + 5. * 3. + 6. * 1.)
    identity = [[1., 0., 0., 0.], [0., 1., 0., 0.], [0., 0., 1., 0.]]
    assert host_ref._apply(identity, (1., 2., 3.)) == (1., 2., 3.)
    # mat4.rs:28-99: the constructors the loader chains (scale, the three rotations, translation): their matrices as written there
    assert host_ref._scale((2., 3., 4.)) == [[2., 0., 0., 0.], [0., 3., 0., 0.], [0., 0., 4., 0.]]
    t = 0.3
    c, sn = np.cos(t), np.sin(t)
    assert host_ref._rot_x(t) == [[1., 0., 0., 0.], [0., c, sn, 0.], [0., -sn, c, 0.]]
    assert host_ref._rot_y(t) == [[c, 0., -sn, 0.], [0., 1., 0., 0.], [sn, 0., c, 0.]]
    assert host_ref._rot_z(t) == [[c, -sn, 0., 0.], [sn, c, 0., 0.], [0., 0., 1., 0.]]
    assert host_ref._apply(host_ref._translate((5., 6., 7.)), (1., 2., 3.)) == (6., 8., 10.)


def test_aabb_known_answers_from_reference_tests():
    # aabb.rs:337-391: centroid, largest_axis, area (half the surface area)
    b = host_ref.Aabb.from_points((0.0, 0.0, 0.0), (4.0, 2.0, 1.0))
    assert b.centroid == (2.0, 1.0, 0.5)
    assert b.largest_axis() == 0
    assert b.area() == 4 * 2 + 2 * 1 + 1 * 4
    assert host_ref.Aabb.from_points((0.0, 0.0, 0.0), (1.0, 5.0, 2.0)).largest_axis() == 1
    assert host_ref.Aabb.from_points((0.0, 0.0, 0.0), (1.0, 1.0, 2.0)).largest_axis() == 2
    # ties: Y over X, Z last (aabb.rs:165-175)
    assert host_ref.Aabb.from_points((0.0, 0.0, 0.0), (2.0, 2.0, 1.0)).largest_axis() == 1
    assert host_ref.Aabb.from_points((0.0, 0.0, 0.0), (2.0, 2.0, 2.0)).largest_axis() == 2


def test_sphere_boxes_known_answers_from_reference_tests():
    """elements/tests.rs:75-99: Sphere::new((0,0,0), 1) is boxed by [-1, 1]^3 and
    Sphere::new((2,1,-1), 1) by x [1,3], y [0,2], z [-2,0]; tests.rs:67-72: centre and radius kept."""
    sc = {"camera": scenes.camera(), "elements": [scenes.sphere((0, 0, 0), 1.0, scenes.lambertian(1, 1, 1), attractor=False),
                                                   scenes.sphere((2, 1, -1), 1.0, scenes.lambertian(1, 1, 1), attractor=False),
                                                   scenes.sphere((1, 2, -1), 2.0, scenes.lambertian(1, 1, 1), attractor=False)]}
    ri = RenderIntegrator.new(sc, scenes.config(8, 8, 1, 1, 1)).prepare()
    assert ri.element_bbox(0) == ((-1.0, -1.0, -1.0), (1.0, 1.0, 1.0))
    assert ri.element_bbox(1) == ((1.0, 0.0, -2.0), (3.0, 2.0, 0.0))
    s2 = ri.scene.contents.spheres[2]
    assert s2.center.tuple() == (1.0, 2.0, -1.0) and s2.radius == 2.0


def test_triangle_record_known_answers_from_reference_tests():
    """elements/tests.rs:28-39: Triangle::new keeps v0 and the edges to v1 = (0,2,0), v2 = (2,2,0)."""
    sc = {"camera": scenes.camera(), "elements": [{"JSONTriangle": {"v0": scenes._v(0, 0, 0), "v1": scenes._v(0, 2, 0), "v2": scenes._v(2, 2, 0),
                                                                     "material": scenes.lambertian(1, 1, 1)}},
                                                   {"JSONTriangle": {"v0": scenes._v(5, 5, 5), "v1": scenes._v(6, 5, 5), "v2": scenes._v(5, 6, 5),
                                                                     "material": scenes.lambertian(1, 1, 1)}}]}
    ri = RenderIntegrator.new(sc, scenes.config(8, 8, 1, 1, 1)).prepare()
    t = ri.scene.contents.triangles[0]
    assert t.v0.tuple() == (0.0, 0.0, 0.0) and t.v0v1.tuple() == (0.0, 2.0, 0.0) and t.v0v2.tuple() == (2.0, 2.0, 0.0)
    assert t.normal.tuple() == (0.0, 0.0, -1.0)  # (v1 - v0) x (v2 - v0), normalised (triangle.rs:58)


def test_obj_loader_fan_and_slashes(tmp_path):
    p = tmp_path / "m.obj"
    p.write_text("# c\nv 0 0 0\nv 1 0 0\nv 1 1 0\nv 0 1 0\nvn 0 0 1\nf 1/1/1 2/1/1 3/1/1 4/1/1\nf -4 -3 -2\n")
    scene = {"camera": scenes.camera(), "elements": [scenes.obj_element(str(p), scenes.lambertian(.5, .5, .5), scale=(2, 2, 2))]}
    ri = RenderIntegrator.new(scene, scenes.config(8, 8, 1, 1, 1)).prepare()
    sc = ri.scene.contents
    assert sc.n_triangles == 3  # quad fan (2) + one triangle
    assert sc.triangles[0].v0v1.tuple() == (2.0, 0.0, 0.0) and sc.triangles[1].v0v2.tuple() == (0.0, 2.0, 0.0)
    assert sc.triangles[2].v0.tuple() == (0.0, 0.0, 0.0)


def test_obj_loader_parallel_parse_is_order_exact(tmp_path, monkeypatch):
    """The loader splits big files over threads; objects, relative indices, polygons and the
    order of the triangles must come out as from a single pass."""
    rng = np.random.RandomState(4)
    lines, n_v = ["# generated"], 0
    for obj in range(5):
        lines.append(f"o part{obj}")
        first = n_v
        for _ in range(300):
            x, y, z = rng.uniform(-3, 3, size=3)
            lines.append(f"v {x:.6f} {y:.6f} {z:.6f}")
            n_v += 1
            if n_v - first >= 4 and rng.uniform() < 0.7:
                k = rng.randint(3, 6)
                if rng.uniform() < 0.5:  # absolute (file-wide) indices into this object, with /vt/vn decorations
                    idx = rng.randint(first + 1, n_v + 1, size=k)
                    lines.append("f " + " ".join(f"{i}/1/1" for i in idx))
                else:                    # relative ones
                    idx = -rng.randint(1, n_v - first + 1, size=k)
                    lines.append("f " + " ".join(str(i) for i in idx))
        lines.append("vn 0 0 1")
    p = tmp_path / "parts.obj"
    p.write_text("\n".join(lines) + "\n")
    scene = {"camera": scenes.camera(), "elements": [scenes.obj_element(str(p), scenes.lambertian(.5, .5, .5), scale=(2, 3, 4),
                                                                          rotate=(0.1, -0.7, 0.3), transpose=(5, 6, 7))]}

    def load():
        ri = RenderIntegrator.new(scene, scenes.config(8, 8, 1, 1, 1)).prepare()
        sc = ri.scene.contents
        tri = np.array([sc.triangles[i].v0.tuple() + sc.triangles[i].v0v1.tuple() + sc.triangles[i].v0v2.tuple() + sc.triangles[i].normal.tuple()
                        for i in range(sc.n_triangles)])
        return tri, _scene_arrays(ri)[1]

    monkeypatch.setenv("RT_HOST_THREADS", "1")
    tri1, nodes1 = load()
    assert len(tri1) > 1500
    monkeypatch.setenv("RT_HOST_THREADS", "7")
    monkeypatch.setenv("RT_HOST_PARSE_CHUNK", "997")
    tri7, nodes7 = load()
    assert np.array_equal(tri1, tri7, equal_nan=True)  # repeated indices give degenerate triangles with NaN normals
    assert nodes1 == nodes7


def test_obj_loader_rejects_indices_outside_the_object(tmp_path):
    p = tmp_path / "bad.obj"
    p.write_text("o a\nv 0 0 0\nv 1 0 0\nv 0 1 0\no b\nv 0 0 1\nv 1 0 1\nv 0 1 1\nf 1 2 3\n")  # face of b names a's vertices
    scene = {"camera": scenes.camera(), "elements": [scenes.obj_element(str(p), scenes.lambertian(.5, .5, .5))]}
    with pytest.raises(abi.RtError) as e:
        RenderIntegrator.new(scene, scenes.config(8, 8, 1, 1, 1)).prepare()
    assert "out of range" in str(e.value)


def test_teapot_like_counts(tmp_scene_dir):
    ri = knot_scene(tmp_scene_dir, 6320, 16, 16, 1, 1, 1)
    sc = ri.scene.contents
    # the teapot scene of the reference: 6 320 mesh tris + 12 box + 12 wall tris + quad light (+ our glass sphere)
    assert sc.n_triangles == 6320 + 24 and sc.n_quads == 1 and sc.n_spheres == 1
    assert sc.n_nodes == 2 * sc.n_elements - 1
    assert 10 <= ri.bvh_depth() <= 64


def test_errors_are_reported_not_thrown(tmp_path):
    with pytest.raises(abi.RtError) as e:
        RenderIntegrator.new_from_json(str(tmp_path / "nope.json"), str(tmp_path / "nope2.json"))
    assert "cannot open" in str(e.value)
    with pytest.raises(abi.RtError):
        RenderIntegrator.new("{", json.dumps(scenes.config(8, 8, 1, 1, 1)))
    bad = scenes.config(8, 8, 1, 1, 1)
    del bad["pixel_radius"]
    with pytest.raises(abi.RtError) as e:
        RenderIntegrator.new(scenes.scene_c1(), bad)
    assert "pixel_radius" in str(e.value)
    ri = RenderIntegrator.new({"camera": scenes.camera(), "elements": [{"JSONBlob": {}}]}, scenes.config(8, 8, 1, 1, 1))
    with pytest.raises(abi.RtError) as e:
        ri.prepare()
    assert "JSONBlob" in str(e.value)


def test_png_writer(tmp_path):
    ri = c1(width=16, height=8)
    px = np.zeros((8, 16, 3))
    px[..., 0] = 1.2          # clamps to 255
    px[..., 2] = 0.5          # truncates to 127 (color.rs:279-286)
    px[0, 0] = [np.nan, -1.0, 0.999]
    sp = np.full((8, 16, 3), 0.25)
    ri.set_pixels(px, sp)
    a, b = str(tmp_path / "a.png"), str(tmp_path / "b.png")
    ri.save_to_png(a, b)
    w, h, rgb, chunks = read_png(a)
    assert (w, h) == (16, 8)
    assert tuple(rgb[3, 5]) == (255, 0, 127)
    assert tuple(rgb[0, 0]) == (255, 0, 254)  # NaN.min(1.0) is 1.0 in Rust
    assert struct.unpack(">I", chunks[b"gAMA"])[0] == 45455
    assert struct.unpack(">8I", chunks[b"cHRM"]) == (31270, 32900, 64000, 33000, 30000, 60000, 15000, 6000)
    _, _, rgb2, _ = read_png(b)
    assert (rgb2 == 63).all()
