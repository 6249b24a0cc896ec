"""Pins the CPU oracle against every known answer the reference's own unit tests
hold for the hot path (the reference cannot be built or run here: it is Rust and
no toolchain exists), then checks the pieces no reference test pins (Sobol, RNG,
filter, stats, PDFs) for the properties their published algorithms guarantee."""
import math

import numpy as np
import pytest

from raytracer_amd import abi
from oracle import host_ref, oracle

EPS = 1e-6  # assert_approx_eq! default


# ------------------------------------------------------------- bvh/aabb.rs:321-335
def test_aabb_hit_known_answers():
    assert oracle.aabb_hit((1.0, 1.0, 1.0), (2.0, 2.0, 2.0), (0.0, 0.0, 0.0), (1.0, 1.0, 1.0)) is True
    assert oracle.aabb_hit((1.0, 1.0, 1.0), (2.0, 2.0, 2.0), (5.0, 5.0, 5.0), (6.0, 6.0, 6.0)) is False


def test_aabb_constructors_known_answers():
    # aabb.rs:266-302 (new_from_points, new_from_aabbs), 337-391 (centroid, largest_axis, area)
    a = host_ref.Aabb.from_points((0.1, 0.1, 0.1), (0.2, 0.0, 0.2))
    assert a.min == (0.1, 0.0, 0.1) and a.max == (0.2, 0.1, 0.2)
    b = host_ref.Aabb.merge(host_ref.Aabb.from_points((1.0, 1.0, 1.0), (2.0, 2.0, 2.0)),
                            host_ref.Aabb.from_points((3.0, 3.0, 3.0), (4.0, 4.0, 4.0)))
    assert b.min == (1.0, 1.0, 1.0) and b.max == (4.0, 4.0, 4.0)
    assert host_ref.Aabb.from_points((1.0, 1.0, 1.0), (2.0, 2.0, 2.0)).centroid == (1.5, 1.5, 1.5)
    assert host_ref.Aabb.from_points((1.0, 1.0, 1.0), (2.0, 3.0, 2.0)).largest_axis() == 1
    assert host_ref.Aabb.from_points((1.0, 1.0, 1.0), (2.0, 3.0, 5.0)).largest_axis() == 2
    assert host_ref.Aabb.from_points((1.0, 1.0, 1.0), (2.0, 3.0, 2.0)).area() == (1. * 2.) + (2. * 1.) + (1. * 1.)
    assert host_ref.Aabb.from_points((-2.0, 1.0, -2.0), (2.0, 5.0, 2.0)).area() == 48.0


# ---------------------------------------------------------------- render/ray.rs:73-92
def test_ray_new_and_at():
    r, signs = oracle.ray_new((0.1, 0.2, 0.3), (0.2, 0.3, 0.4))
    n = math.sqrt(0.2 * 0.2 + 0.3 * 0.3 + 0.4 * 0.4)
    assert np.allclose(r[0], (0.1, 0.2, 0.3), atol=EPS)
    assert np.allclose(r[1], (0.2 / n, 0.3 / n, 0.4 / n), atol=EPS)
    assert np.array_equal(r[2], 1.0 / r[1]) and signs == [0, 0, 0]
    assert np.allclose(oracle.ray_at((0, 0, 0), (0, 0, 1), 5.0), (0.0, 0.0, 5.0), atol=EPS)
    # sign convention: `< 0.0`, so -0.0 counts as positive while its inverse is -inf (ray.rs:38-47)
    r, signs = oracle.ray_new((0, 0, 0), (-0.0, -1.0, 0.0))
    assert signs == [0, 1, 0] and r[2][0] == -math.inf and r[2][2] == math.inf


# ------------------------------------------------------------ render/color.rs:256-286
def test_color_known_answers():
    c = oracle.color_ops((0.5 + 0.1, 0.4 + 1.0, -2.3 + 0.1), 2.2)
    assert np.allclose(c["clamp"], (0.6, 1.0, 0.0), atol=EPS)
    g = oracle.color_ops((0.5, 0.4, 0.3), 2.2)["gamma"]
    assert np.allclose(g, [x ** (1.0 / 2.2) for x in (0.5, 0.4, 0.3)], atol=EPS)
    assert oracle.color_ops((1.2, 0.0, 0.5), 2.2)["rgb"] == [255, 0, 127]
    # NaN clamps to 1.0 (f64::min returns the other operand), color.rs:29-35
    assert oracle.color_ops((float("nan"), 2.0, -1.0), 2.2)["clamp"] == [1.0, 1.0, 0.0]
    lum = oracle.color_ops((0.2, 0.5, 0.9), 2.2)
    assert lum["illuminance"] == 0.2126 * 0.2 + 0.7152 * 0.5 + 0.0722 * 0.9
    assert lum["illuminance_approx"] == (0.2 + 0.2 + 0.9 + 0.5 + 0.5 + 0.5) / 6.0


# ----------------------------------------------------------- linalg/vec3.rs:465-526
def test_vec3_known_answers():
    assert abs(oracle.vec3_ops((1.0, 0.0, 5.0), (0.0, 2.0, 4.0))["distance"] - 2.44948974278) < EPS
    assert abs(oracle.vec3_ops((2.0, 3.0, 5.0), (2.0, 0.0, 9.0))["distance"] - 5.0) < EPS
    assert abs(oracle.vec3_ops((3.0, 1.0, 2.0), (0, 0, 0))["length"] - 3.741657386773941) < EPS
    assert oracle.vec3_ops((2.0, 3.0, 6.0), (0, 0, 0))["length"] == 7.0
    assert oracle.vec3_ops((0.0, 0.0, 0.0), (0, 0, 0))["length"] == 0.0
    assert oracle.vec3_ops((3.0, 1.0, 2.0), (0, 0, 0))["length_squared"] == 14.0
    assert abs(oracle.vec3_ops((0.1, 0.2, 0.3), (0.2, 0.3, 0.4))["dot"] - 0.2) < EPS
    assert oracle.vec3_ops((1.0, 0.0, 0.0), (3.0, 2.0, 4.0))["cross"] == [0.0, -4.0, 2.0]
    assert np.allclose(oracle.vec3_ops((0.0, 2.0, 0.0), (0, 0, 0))["normalized"], (0.0, 1.0, 0.0), atol=EPS)
    assert np.allclose(oracle.vec3_ops((3.0, 1.0, 2.0), (0, 0, 0))["normalized"],
                       (0.801783725737273, 0.267261241912424, 0.534522483824848), atol=EPS)


# -------------------------------------------------------- elements/tests.rs:101-115
def test_sphere_hit_known_answer():
    h = oracle.sphere_hit((0.0, 0.0, -3.0), 1.0, (0.0, 0.0, 0.0), (0.0, 0.0, -1.0), 0.0, 1.7976931348623157e308)
    assert h["t"] == 2.0 and h["point"] == [0.0, 0.0, -2.0]
    assert h["front_face"] and h["normal"] == [0.0, 0.0, 1.0]


def test_sphere_hit_quirks():
    # only the numerically smaller root is considered: a ray starting inside never hits (sphere.rs:120-126)
    assert oracle.sphere_hit((0, 0, 0), 1.0, (0, 0, 0), (0, 0, 1), 0.0001, 1e308) is None
    # strict interval
    assert oracle.sphere_hit((0.0, 0.0, -3.0), 1.0, (0, 0, 0), (0, 0, -1), 0.0, 2.0) is None
    assert oracle.sphere_hit((0.0, 0.0, -3.0), 1.0, (0, 0, 0), (0, 0, -1), 2.0, 3.0) is None


def _tri(v0, v1, v2):
    t = abi.RtTriangle()
    e1, e2 = host_ref.sub(v1, v0), host_ref.sub(v2, v0)
    n = host_ref.normalized(host_ref.cross(e1, e2))
    for name, v in (("v0", v0), ("v0v1", e1), ("v0v2", e2), ("normal", n)):
        f = getattr(t, name)
        f.x, f.y, f.z = v
    return t


def _quad(q, u, v):
    r = abi.RtQuad()
    n = host_ref.cross(u, v)
    normal = host_ref.normalized(n)
    w = host_ref.divf(n, host_ref.dot(n, n))
    for name, val in (("q", q), ("u", u), ("v", v), ("normal", normal), ("w", w)):
        f = getattr(r, name)
        f.x, f.y, f.z = val
    r.d = host_ref.dot(normal, q)
    r.area = host_ref.length(n)
    return r


def test_triangle_hit_culls_back_faces_and_is_inclusive():
    tri = _tri((0.0, 0.0, 0.0), (1.0, 0.0, 0.0), (0.0, 1.0, 0.0))  # normal +z
    assert oracle.triangle_hit(tri, (0.25, 0.25, 1.0), (0, 0, -1), 0.0001, 1e308) == 1.0
    assert oracle.triangle_hit(tri, (0.25, 0.25, -1.0), (0, 0, 1), 0.0001, 1e308) is None  # back face (triangle.rs:81)
    assert oracle.triangle_hit(tri, (0.25, 0.25, 1.0), (0, 0, -1), 0.0001, 1.0) == 1.0      # t == tmax accepted
    assert oracle.triangle_hit(tri, (0.25, 0.25, 1.0), (0, 0, -1), 1.0, 5.0) == 1.0         # t == tmin accepted
    assert oracle.triangle_hit(tri, (0.0, 0.0, 1.0), (0, 0, -1), 0.0001, 1e308) == 1.0      # vertex: u = v = 0
    assert oracle.triangle_hit(tri, (0.75, 0.75, 1.0), (0, 0, -1), 0.0001, 1e308) is None   # u + v > 1


# ----------------------------------------------------------- render/interval.rs:42-82
def test_interval_known_answers():
    """Interval has no type of its own here: its three operations live in the box merge, the
    box padding and the primitives' range check.  The reference's own cases, replayed there."""
    # new_from_intervals((0.1, 0.2), (0.3, -0.4)) == (0.1, 0.2), interval.rs:54-62 -- per axis of Aabb::merge
    m = host_ref.Aabb.merge(host_ref.Aabb((0.1, 0.1, 0.1), (0.2, 0.2, 0.2)), host_ref.Aabb((0.3, 0.3, 0.3), (-0.4, -0.4, -0.4)))
    assert m.min == (0.1, 0.1, 0.1) and m.max == (0.2, 0.2, 0.2)
    # size() == max - min, interval.rs:65-68; expand(delta) moves each end by delta / 2, interval.rs:71-75
    # (Aabb::pad expands an axis thinner than 1e-4 by 1e-4, aabb.rs:122-152)
    b = host_ref.Aabb((0.1, 0.0, 0.0), (0.1, 1.0, 1.0)).pad()
    assert b.min[0] == 0.1 - 0.0001 / 2.0 and b.max[0] == 0.1 + 0.0001 / 2.0 and (b.min[1], b.max[1]) == (0.0, 1.0)
    # contains(): both ends inclusive, interval.rs:78-82 -- the [1, 2] interval of the reference's test as a ray interval
    tri = _tri((0.0, 0.0, 0.0), (1.0, 0.0, 0.0), (0.0, 1.0, 0.0))
    assert oracle.triangle_hit(tri, (0.25, 0.25, 1.5), (0, 0, -1), 1.0, 2.0) == 1.5
    assert oracle.triangle_hit(tri, (0.25, 0.25, 2.5), (0, 0, -1), 1.0, 2.0) is None
    assert oracle.triangle_hit(tri, (0.25, 0.25, 2.0), (0, 0, -1), 1.0, 2.0) == 2.0


def test_quad_hit_and_pdf():
    quad = _quad((343.0, 554.0, 332.0), (-130.0, 0.0, 0.0), (0.0, 0.0, -105.0))
    assert quad.area == 13650.0
    t = oracle.quad_hit(quad, (278.0, 0.0, 279.5), (0.0, 1.0, 0.0), 0.0001, 1e308)
    assert t == 554.0
    assert oracle.quad_hit(quad, (278.0, 600.0, 279.5), (0.0, -1.0, 0.0), 0.0001, 1e308) == 46.0  # two-sided
    assert oracle.quad_hit(quad, (100.0, 0.0, 279.5), (0.0, 1.0, 0.0), 0.0001, 1e308) is None       # outside alpha range
    assert oracle.quad_hit(quad, (278.0, 0.0, 279.5), (1.0, 0.0, 0.0), 0.0001, 1e308) is None       # parallel
    # pdf = distance^2 / (cos * area) straight up from below (quad.rs:155-167)
    assert oracle.quad_pdf_value(quad, (278.0, 0.0, 279.5), (0.0, 1.0, 0.0)) == 554.0 * 554.0 / 13650.0
    assert oracle.quad_pdf_value(quad, (278.0, 0.0, 279.5), (1.0, 0.0, 0.0)) == 0.0


def test_sphere_pdf():
    s = abi.RtSphere()
    s.center.x, s.center.y, s.center.z, s.radius = 0.0, 0.0, 10.0, 2.0
    cos_max = math.sqrt(1.0 - 4.0 / 100.0)
    assert oracle.sphere_pdf_value(s, (0, 0, 0), (0, 0, 1)) == 1.0 / (2.0 * math.pi * (1.0 - cos_max))
    assert oracle.sphere_pdf_value(s, (0, 0, 0), (0, 1, 0)) == 0.0


# --------------------------------------------------------------------- third-party restatements
def test_sobol_is_a_scrambled_stratified_sequence():
    """sobol_burley::sample is restated from the published algorithm (parity unpinned).  Whatever
    the exact hash constants, an Owen-scrambled Sobol point set keeps the (0,2)-net property:
    the first 2^k points put exactly one point in every elementary interval of volume 2^-k."""
    for seed in (0, 1, 7, 600 * 599):
        pts = np.array([[oracle.sobol_sample(i, d, seed) for d in (0, 1)] for i in range(64)])
        assert pts.min() >= 0.0 and pts.max() < 1.0
        for n in (16, 64):
            p = pts[:n]
            k = int(math.log2(n))
            for a in range(k + 1):
                cells = (np.floor(p[:, 0] * 2 ** a).astype(int) * 2 ** (k - a) + np.floor(p[:, 1] * 2 ** (k - a)).astype(int))
                assert len(set(cells)) == n, (seed, n, a)
    a = [oracle.sobol_sample(i, 0, 1) for i in range(16)]
    b = [oracle.sobol_sample(i, 0, 2) for i in range(16)]
    assert a != b  # the seed changes the sequence
    assert a == [oracle.sobol_sample(i, 0, 1) for i in range(16)]  # and it is a pure function


def test_sample_offset_matches_integrator_formula():
    # integrator.rs:275, 286-291: seed = x * y (product), offset = sobol * radius - radius / 2
    sx, sy = oracle.sample_offset(2.0, 17, 23, 5)
    assert sx == oracle.sobol_sample(5, 0, 17 * 23) * 2.0 - 1.0
    assert sy == oracle.sobol_sample(5, 1, 17 * 23) * 2.0 - 1.0
    assert -1.0 <= sx < 1.0 and -1.0 <= sy < 1.0
    # first row and first column share seed 0
    assert oracle.sample_offset(2.0, 0, 9, 3) == oracle.sample_offset(2.0, 31, 0, 3)


def test_mitchell_filter():
    # B = C = 1/3, filter.rs:45-67: value at the centre is ((6 - 2B) / 6)^2, first branch only for |x| < r/2
    b = c = 1.0 / 3.0
    centre = (6.0 - 2.0 * b) * (1.0 / 6.0)
    assert oracle.filter_evaluate(0.0, 0.0, 2.0) == centre * centre
    for sx in (-0.999, -0.3, 0.5, 0.99):
        fx = abs(sx * 0.5) * 2.0
        m = ((12.0 - 9.0 * b - 6.0 * c) * fx * fx * fx + (-18.0 + 12.0 * b + 6.0 * c) * fx * fx + (6.0 - 2.0 * b)) * (1.0 / 6.0)
        assert oracle.filter_evaluate(sx, 0.0, 2.0) == m * centre
        assert oracle.filter_evaluate(sx, 0.0, 2.0) > 0.0
    # the zero tail (fx >= 2)
    assert oracle.filter_evaluate(2.0, 0.0, 2.0) == 0.0


def test_xoshiro256pp_known_answer():
    """SmallRng::seed_from_u64(0): the state is the first four SplitMix64 outputs for seed 0
    (published test vector of Vigna's splitmix64.c), and the stream is xoshiro256++'s recurrence
    applied to that state, written out independently here."""
    u = oracle.rng_stream(abi.RT_RNG_REF, 0, 0, 4)
    # SplitMix64(0) outputs (Vigna): e220a8397b1dcdaf, 6e789e6aa1b965f4, 06c45d188009454f, f88bb8a8724c81ec
    s = [0xe220a8397b1dcdaf, 0x6e789e6aa1b965f4, 0x06c45d188009454f, 0xf88bb8a8724c81ec]

    def rotl(x, k):
        return ((x << k) | (x >> (64 - k))) & (2 ** 64 - 1)

    out = []
    for _ in range(4):
        out.append((rotl((s[0] + s[3]) & (2 ** 64 - 1), 23) + s[0]) & (2 ** 64 - 1))
        t = (s[1] << 17) & (2 ** 64 - 1)
        s[2] ^= s[0]; s[3] ^= s[1]; s[1] ^= s[2]; s[0] ^= s[3]; s[2] ^= t; s[3] = rotl(s[3], 45)
    assert u == out


def test_rng_conversions():
    f = oracle.rng_stream(abi.RT_RNG_REF, 5, 1, 2000, lo=0.0, hi=1.0)
    assert min(f) >= 0.0 and max(f) < 1.0 and 0.45 < np.mean(f) < 0.55
    g = oracle.rng_stream(abi.RT_RNG_REF, 5, 1, 2000, lo=-1.0, hi=1.0)
    assert min(g) >= -1.0 and max(g) < 1.0 and abs(np.mean(g)) < 0.06
    # same underlying u64 stream: gen_range(-1,1) == 2 * gen_range(0,1) - 1 exactly
    assert g == [2.0 * x - 1.0 for x in f]
    h = oracle.rng_stream(abi.RT_RNG_REF, 5, 2, 100)
    assert min(h) >= 0.0 and max(h) < 1.0
    idx = oracle.rng_stream(abi.RT_RNG_REF, 9, 3, 3000, n=3)
    assert set(idx) == {0, 1, 2}
    assert oracle.rng_stream(abi.RT_RNG_REF, 9, 3, 50, n=1) == [0] * 50
    # counter mode: different paths get different streams, same path the same stream
    a = oracle.rng_stream(abi.RT_RNG_CTR, 1, 0, 8)
    assert a == oracle.rng_stream(abi.RT_RNG_CTR, 1, 0, 8) and a != oracle.rng_stream(abi.RT_RNG_CTR, 2, 0, 8)


def test_stats_match_the_definition():
    # stats.rs:21-51 on a hand-computed case
    colors = [(1.0, 0.0, 0.0), (0.0, 1.0, 0.0), (0.0, 0.0, 1.0)]
    weights = [1.0, 2.0, 1.0]
    mean, var, iv = oracle.stats(colors, weights)
    assert mean == [0.25, 0.5, 0.25]
    lum = [(2 * c[0] + c[2] + 3 * c[1]) / 6.0 for c in colors]
    ml = (0.25 + 0.25 + 0.25 + 0.5 + 0.5 + 0.5) / 6.0
    v = sum(w * (l - ml) ** 2 for w, l in zip(weights, lum)) / 4.0
    assert abs(var - v) < 1e-15 and abs(iv - 1.96 * math.sqrt(v) / math.sqrt(3.0)) < 1e-15
    assert oracle.stats([(0.3, 0.3, 0.3)], [5.0])[1] == 0.0  # a single batch has zero variance


def test_onb_is_orthonormal_and_follows_the_axis_rule():
    for w in ((0.0, 1.0, 0.0), (1.0, 0.0, 0.0), (0.95, 0.1, 0.0), (0.3, -0.4, 0.8)):
        m = oracle.onb(w)
        assert np.allclose(m @ m.T, np.eye(3), atol=1e-12)
        assert np.allclose(m[2], np.array(w) / np.linalg.norm(w), atol=1e-15)
    # |w.x| > 0.9 switches the helper axis to y: v = w x (0,1,0) has no y component (onb.rs:31-45)
    assert oracle.onb((1.0, 0.0, 0.0))[1][1] == 0.0
