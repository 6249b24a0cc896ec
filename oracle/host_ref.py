"""Pure-Python restatement of the reference's load-time host code, used only
to check the C++ host (raytracer_amd/csrc/host) on small scenes.

TEST INFRASTRUCTURE ONLY.  Follows, with Python floats (IEEE f64, no fused
multiply-add):
  render/camera.rs:35-91        Camera::new
  elements/triangle.rs:55-70    Triangle::new
  elements/quad.rs:69-89        Quad::new
  elements/sphere.rs:29-47,72-83
  elements/cube.rs:26-144       JSONCube::add_as_element
  elements/obj.rs:27-103        JSONObj::add_as_element
  linalg/mat4.rs:28-99,281-304  transforms
  bvh/aabb.rs:45-175            Aabb
  bvh/bvh_node_sah.rs:28-79,147-159,256-461   BVH build (Mid / 16-bucket SAH)
"""
import math

EPSILON = 2.220446049250313e-16


def _fmin(a, b):  # f64::min: the non-NaN operand
    if a != a:
        return b
    if b != b:
        return a
    return a if a < b else b


def _fmax(a, b):
    if a != a:
        return b
    if b != b:
        return a
    return a if a > b else b


def add(a, b):
    return (a[0] + b[0], a[1] + b[1], a[2] + b[2])


def sub(a, b):
    return (a[0] - b[0], a[1] - b[1], a[2] - b[2])


def mulf(a, q):
    return (a[0] * q, a[1] * q, a[2] * q)


def divf(a, q):
    return (a[0] / q, a[1] / q, a[2] / q)


def dot(a, b):
    return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]


def cross(a, b):
    return (a[1] * b[2] - a[2] * b[1], a[2] * b[0] - a[0] * b[2], a[0] * b[1] - a[1] * b[0])


def length(a):
    return math.sqrt(a[0] * a[0] + a[1] * a[1] + a[2] * a[2])


def normalized(a):
    l = length(a)
    return (a[0] / l, a[1] / l, a[2] / l)


def vmin(a, b):
    return (_fmin(a[0], b[0]), _fmin(a[1], b[1]), _fmin(a[2], b[2]))


def vmax(a, b):
    return (_fmax(a[0], b[0]), _fmax(a[1], b[1]), _fmax(a[2], b[2]))


class Aabb:
    def __init__(self, mn, mx):
        self.min, self.max = mn, mx
        self.centroid = tuple(mn[a] + (mx[a] - mn[a]) / 2. for a in range(3))  # aabb.rs:214-220

    @staticmethod
    def from_points(p, q):  # aabb.rs:59-75
        return Aabb(tuple(_fmin(q[a], p[a]) for a in range(3)), tuple(_fmax(q[a], p[a]) for a in range(3)))

    @staticmethod
    def merge(a, b):  # aabb.rs:81-97
        return Aabb(vmin(a.min, b.min), vmax(a.max, b.max))

    def pad(self):  # aabb.rs:122-152
        delta = 0.0001
        mn, mx = list(self.min), list(self.max)
        for a in range(3):
            if not (mx[a] - mn[a] >= delta):
                mn[a], mx[a] = mn[a] - delta / 2.0, mx[a] + delta / 2.0
        return Aabb(tuple(mn), tuple(mx))

    def area(self):  # aabb.rs:115-119
        d = sub(self.max, self.min)
        return d[0] * d[1] + d[1] * d[2] + d[2] * d[0]

    def largest_axis(self):  # aabb.rs:165-175
        s = sub(self.max, self.min)
        if s[0] > s[1] and s[0] > s[2]:
            return 0
        return 1 if s[1] > s[2] else 2


def _v(j):
    return (float(j["x"]), float(j["y"]), float(j["z"]))


def make_camera(cfg, jc):
    relative_up = (0.0, 1.0, 0.0)
    center, look_at = _v(jc["camera_center"]), _v(jc["camera_look_at"])
    focus_dist = float(jc["camera_focus_dist"])
    w_img, h_img = float(cfg["img_width"]), float(cfg["img_height"])
    ratio = w_img / h_img
    theta = float(jc["camera_fov_vertical"]) * (math.pi / 180.0)
    h = math.tan(theta / 2.0)
    vv = 2.0 * h * focus_dist
    vu = vv * ratio
    w = normalized(sub(center, look_at))
    u = normalized(cross(relative_up, w))
    v = cross(w, u)
    viewport_u = mulf(u, vu)
    viewport_v = mulf(mulf(v, -1.0), vv)
    pdu = divf(viewport_u, w_img)
    pdv = divf(viewport_v, w_img / ratio)
    upper_left = sub(sub(sub(center, mulf(w, focus_dist)), divf(viewport_u, 2.0)), divf(viewport_v, 2.0))
    pixel00 = add(upper_left, mulf(add(pdu, pdv), 0.5))
    radius = focus_dist * math.tan((float(jc["camera_defocus_angle"]) / 2.0) * (math.pi / 180.0))
    return {
        "camera_center": center, "pixel00_loc": pixel00, "pixel_delta_u": pdu, "pixel_delta_v": pdv,
        "defocus_disk_u": mulf(u, radius), "defocus_disk_v": mulf(v, radius),
    }


def _apply(m, v):
    return tuple(m[r][0] * v[0] + m[r][1] * v[1] + m[r][2] * v[2] + m[r][3] * 1.0 for r in range(3))


def _translate(d):
    return [[1., 0., 0., d[0]], [0., 1., 0., d[1]], [0., 0., 1., d[2]]]


def _scale(s):
    return [[s[0], 0., 0., 0.], [0., s[1], 0., 0.], [0., 0., s[2], 0.]]


def _rot_x(t):
    return [[1., 0., 0., 0.], [0., math.cos(t), math.sin(t), 0.], [0., -math.sin(t), math.cos(t), 0.]]


def _rot_y(t):
    return [[math.cos(t), 0., -math.sin(t), 0.], [0., 1., 0., 0.], [math.sin(t), 0., math.cos(t), 0.]]


def _rot_z(t):
    return [[math.cos(t), -math.sin(t), 0., 0.], [math.sin(t), math.cos(t), 0., 0.], [0., 0., 1., 0.]]


class _Scene:
    def __init__(self):
        self.elements, self.triangles, self.quads, self.spheres = [], [], [], []

    def add_triangle(self, v0, v1, v2):
        e1, e2 = sub(v1, v0), sub(v2, v0)
        self.triangles.append({"v0": v0, "v0v1": e1, "v0v2": e2, "normal": normalized(cross(e1, e2))})
        box = Aabb.from_points(vmin(v0, vmin(v1, v2)), vmax(v0, vmax(v1, v2))).pad()
        self.elements.append({"kind": 2, "index": len(self.triangles) - 1, "box": box})

    def add_quad(self, q, u, v, attractor):
        self.quads.append({"q": q, "u": u, "v": v, "attractor": attractor})
        box = Aabb.from_points(q, add(add(q, u), v)).pad()
        self.elements.append({"kind": 1, "index": len(self.quads) - 1, "box": box})

    def add_sphere(self, c, r, attractor):
        self.spheres.append({"center": c, "radius": r, "attractor": attractor})
        rv = (r, r, r)
        self.elements.append({"kind": 0, "index": len(self.spheres) - 1, "box": Aabb.from_points(sub(c, rv), add(c, rv))})


def _transform_chain(body):
    ms = []
    if body.get("scale") is not None:
        ms.append(_scale(_v(body["scale"])))
    if body.get("rotate") is not None:
        r = body["rotate"]
        ms += [_rot_x(float(r["theta_x"])), _rot_y(float(r["theta_y"])), _rot_z(float(r["theta_z"]))]
    return ms


def _expand(scene):
    s = _Scene()
    for je in scene["elements"]:
        (tag, body), = je.items()
        if tag == "JSONQuad":
            s.add_quad(_v(body["q"]), _v(body["u"]), _v(body["v"]), body.get("attractor") is True)
        elif tag == "JSONSphere":
            c = _v(body["center"])
            if body.get("transpose") is not None:
                c = _apply(_translate(_v(body["transpose"])), c)
            s.add_sphere(c, float(body["radius"]), body.get("attractor") is True)
        elif tag == "JSONTriangle":
            s.add_triangle(_v(body["v0"]), _v(body["v1"]), _v(body["v2"]))
        elif tag == "JSONBox":
            a, b = _v(body["a"]), _v(body["b"])
            mn, mx = vmin(a, b), vmax(a, b)
            d = sub(mx, mn)
            vs = [(1., -1., 1.), (1., -1., -1.), (1., 1., -1.), (1., 1., 1.), (-1., -1., 1.), (-1., -1., -1.), (-1., 1., -1.), (-1., 1., 1.)]
            tris = [(1, 5, 6), (1, 6, 2), (5, 7, 6), (5, 4, 7), (0, 2, 3), (0, 1, 2), (2, 7, 3), (2, 6, 7), (0, 7, 4), (0, 3, 7), (4, 5, 1), (1, 0, 4)]
            chain = [_scale((d[0] / 2., d[1] / 2., d[2] / 2.))] + _transform_chain(body)
            chain.append(_translate((mn[0] + d[0] / 2.0, mn[1] + d[1] / 2.0, mn[2] + d[2] / 2.0)))
            if body.get("transpose") is not None:
                chain.append(_translate(_v(body["transpose"])))
            for m in chain:
                vs = [_apply(m, v) for v in vs]
            for t in tris:
                s.add_triangle(vs[t[0]], vs[t[1]], vs[t[2]])
        elif tag == "JSONObj":
            vs, faces = [], []
            for line in open(body["filepath"]):
                p = line.split()
                if not p:
                    continue
                if p[0] == "v":
                    vs.append((float(p[1]), float(p[2]), float(p[3])))
                elif p[0] == "f":
                    idx = [int(tok.split("/")[0]) for tok in p[1:]]
                    idx = [i - 1 if i > 0 else len(vs) + i for i in idx]
                    for k in range(1, len(idx) - 1):
                        faces.append((idx[0], idx[k], idx[k + 1]))
            chain = _transform_chain(body)
            if body.get("transpose") is not None:
                chain.append(_translate(_v(body["transpose"])))
            for m in chain:
                vs = [_apply(m, v) for v in vs]
            for a, b, c in faces:
                s.add_triangle(vs[a], vs[b], vs[c])
        else:
            raise ValueError(tag)
    return s


def _bounds(objects, indices):  # aabb.rs:99-112
    box = objects[indices[0]]["box"]
    if len(indices) > 1:
        for i in indices:
            box = Aabb.merge(box, objects[i]["box"])
    return box


def _leaf(e):
    z = (0.0, 0.0, 0.0)
    return (1, e, 0, 0, z, z, z, z)


def _build(objects, indices, nodes, method):
    if len(indices) == 1:
        nodes.append(_leaf(indices[0]))
        return len(nodes) - 1
    if len(indices) == 2:
        me = len(nodes)
        l, r = objects[indices[0]]["box"], objects[indices[1]]["box"]
        nodes.append((0, 0, me + 1, me + 2, l.min, l.max, r.min, r.max))
        nodes.append(_leaf(indices[0]))
        nodes.append(_leaf(indices[1]))
        return me
    me = len(nodes)
    nodes.append(None)
    cb = Aabb.from_points(objects[indices[0]]["box"].centroid, objects[indices[1]]["box"].centroid)
    for i in indices:
        c = objects[i]["box"].centroid
        cb = Aabb.merge(cb, Aabb.from_points(c, c))
    axis = cb.largest_axis()
    extent = cb.max[axis] - cb.min[axis]
    left, right = [], []
    if extent < EPSILON:
        mid = len(indices) // 2
        left, right = list(indices[:mid]), list(indices[mid:])
    elif method == "Mid":
        split = cb.min[axis] + extent / 2.0
        for i in indices:
            (left if objects[i]["box"].centroid[axis] < split else right).append(i)
    else:
        buckets = [[] for _ in range(16)]
        k0 = cb.min[axis]
        k1 = (16.0 * (1.0 - 0.0001)) / (cb.max[axis] - cb.min[axis])
        for i in indices:
            buckets[int(k1 * (objects[i]["box"].centroid[axis] - k0))].append(i)
        best_split, best_cost = 0, math.inf
        for split in range(1, 16):
            la = lc = ra = rc = 0.0
            for b in range(16):
                if buckets[b]:
                    area = _bounds(objects, buckets[b]).area()
                    if b < split:
                        la += area
                        lc += float(len(buckets[b]))
                    else:
                        ra += area
                        rc += float(len(buckets[b]))
            cost = la * lc + ra * rc
            if cost < best_cost:
                best_split, best_cost = split, cost
        for b in range(16):
            (left if b < best_split else right).extend(buckets[b])
    l = _build(objects, left, nodes, method)
    r = _build(objects, right, nodes, method)
    lb, rb = _bounds(objects, left), _bounds(objects, right)
    nodes[me] = (0, 0, l, r, lb.min, lb.max, rb.min, rb.max)
    return me


def build_scene(scene, cfg):
    """-> dict(elements, triangles, nodes, attractors) in the layout the tests compare."""
    import sys
    s = _expand(scene)
    nodes = []
    sys.setrecursionlimit(10000)
    _build(s.elements, list(range(len(s.elements))), nodes, cfg.get("bvh_split_method") or "Mid")
    attractors = [i for i, e in enumerate(s.elements)
                  if (e["kind"] == 0 and s.spheres[e["index"]]["attractor"]) or (e["kind"] == 1 and s.quads[e["index"]]["attractor"])]
    return {
        "elements": [{"kind": e["kind"], "index": e["index"], "bbox": (e["box"].min, e["box"].max)} for e in s.elements],
        "triangles": s.triangles,
        "nodes": nodes,
        "attractors": attractors,
    }
