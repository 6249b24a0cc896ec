"""Known answers at the level of `ray_color` (integrator.rs:344-470) that follow from the reference's rules
alone, whatever random numbers are drawn: the reference holds no test above the primitives, so these scenes are
built to have a radiance that can be written down by hand.

  light seen on its front face      -> its albedo            (diffuse_light.rs:20-29, integrator.rs:369-376)
  light seen from behind, or a miss -> black                 (integrator.rs:360-366)
  mirror (Metal, fuzz 0) to a light -> albedo_metal * albedo_light, two BVH queries per sample (metal.rs:21-53)
  two facing mirrors                -> black after exactly max_depth queries per sample (integrator.rs:350-353)
  Lambertian body in a furnace      -> albedo * L exactly without attractors (scattering_pdf / pdf = 1, integrator.rs:413),
                                       and albedo * L in the mean with a quad or a sphere attractor: the mixture density
                                       of pdf.rs:55-69 with Quad::pdf_value / Sphere::pdf_value is unbiased only if every
                                       one of those pieces is right

Each answer is checked on the CPU oracle here and, in the tests marked gpu, on the device."""
import numpy as np
import pytest

from raytracer_amd import RenderIntegrator, scenes
from oracle import oracle

L = (1.5, 1.0, 0.5)


def _quad(q, u, v, material, attractor=False):
    e = {"q": scenes._v(*q), "u": scenes._v(*u), "v": scenes._v(*v), "material": material}
    if attractor:
        e["attractor"] = True
    return {"JSONQuad": e}


def _far_speck():
    # a BVH wants more than one element; this one is behind the camera and never seen
    return _quad((5000, 5000, -5000), (0, 1, 0), (1, 0, 0), scenes.lambertian(0.5, 0.5, 0.5))


def _integrator(elements, camera, width=24, height=24, batch=8, max_batches=2, max_depth=8):
    cfg = scenes.config(width, height, batch, max_batches, max_batches, max_depth=max_depth)
    return RenderIntegrator.new({"camera": camera, "elements": elements}, cfg).prepare()


def _render(ri, where):
    cfg = ri.config.copy()
    if where == "oracle":
        out, st = oracle.render(ri.scene, cfg, 8)
    else:
        out, st = ri.resident(0).render(cfg)
    return out["linear"], st


WHERE = ["oracle", pytest.param("device", marks=pytest.mark.gpu)]
CAM = dict(center=(0.0, 0.0, -10.0), look_at=(0.0, 0.0, 0.0), fov=5.0, focus_dist=10.0)


# a quad's normal is u x v (quad.rs:69-89) and is never flipped; front face = the side the normal points to
def _light_facing_camera(albedo=L):
    return _quad((-50, -50, 0), (0, 100, 0), (100, 0, 0), scenes.diffuse_light(*albedo))     # (0,100,0) x (100,0,0) = -z


def _light_facing_away(albedo=L):
    return _quad((-50, -50, 0), (100, 0, 0), (0, 100, 0), scenes.diffuse_light(*albedo))     # +z


@pytest.mark.parametrize("where", WHERE)
def test_a_light_is_seen_on_its_front_face_only(where):
    ri = _integrator([_light_facing_camera(), _far_speck()], scenes.camera(**CAM))
    img, st = _render(ri, where)
    assert np.allclose(img, np.array(L), rtol=1e-13, atol=0), np.abs(img - np.array(L)).max()
    assert st.rays == st.samples                       # one BVH query per sample: an emitter ends the path
    ri = _integrator([_light_facing_away(), _far_speck()], scenes.camera(**CAM))
    img, st = _render(ri, where)
    assert (img == 0.0).all() and st.rays == st.samples


@pytest.mark.parametrize("where", WHERE)
def test_a_miss_is_black(where):
    ri = _integrator([_far_speck(), _far_speck()], scenes.camera(**CAM))
    img, st = _render(ri, where)
    assert (img == 0.0).all() and st.rays == st.samples


@pytest.mark.parametrize("where", WHERE)
def test_a_mirror_multiplies_by_its_albedo(where):
    am = (0.9, 0.5, 0.25)
    # mirror through the origin with normal (-1,0,-1)/sqrt(2): (0,s,0) x (s,0,-s) = (-s^2, 0, -s^2); rays along +z leave along -x
    s = 8.0
    mirror = _quad((-s / 2, -s / 2, s / 2), (0, s, 0), (s, 0, -s), scenes.metal(*am, 0.0))
    light = _quad((-50, -100, -100), (0, 200, 0), (0, 0, 200), scenes.diffuse_light(*L))     # normal +x, seen by rays along -x
    ri = _integrator([mirror, light, _far_speck()], scenes.camera(**CAM))
    img, st = _render(ri, where)
    want = np.array(am) * np.array(L)
    assert np.allclose(img, want, rtol=1e-13, atol=0), np.abs(img - want).max()
    assert st.rays == 2 * st.samples


@pytest.mark.parametrize("where", WHERE)
@pytest.mark.parametrize("max_depth", [1, 3, 8])
def test_facing_mirrors_end_black_after_max_depth_queries(where, max_depth):
    a = _quad((-500, -500, 10), (0, 1000, 0), (1000, 0, 0), scenes.metal(0.9, 0.9, 0.9, 0.0))    # normal -z, in front of the camera
    b = _quad((-500, -500, -10), (1000, 0, 0), (0, 1000, 0), scenes.metal(0.9, 0.9, 0.9, 0.0))   # normal +z, behind it
    cam = scenes.camera(center=(0.0, 0.0, 0.0), look_at=(0.0, 0.0, 1.0), fov=5.0, focus_dist=1.0)
    ri = _integrator([a, b], cam, max_depth=max_depth)
    img, st = _render(ri, where)
    assert (img == 0.0).all()
    assert st.rays == max_depth * st.samples           # ray_color(depth 0) returns black without a query


def _furnace(attractor=None):
    """A Lambertian sphere inside a closed cube whose six faces emit L towards the inside."""
    h = 50.0
    emit = scenes.diffuse_light(*L)
    faces = [
        _quad((-h, -h, h), (0, 2 * h, 0), (2 * h, 0, 0), emit),                       # z = +h, normal -z
        _quad((-h, -h, -h), (2 * h, 0, 0), (0, 2 * h, 0), emit, attractor == "quad"),  # z = -h, normal +z: the wall the body's visible side faces
        _quad((h, -h, -h), (0, 0, 2 * h), (0, 2 * h, 0), emit),                       # x = +h, normal -x
        _quad((-h, -h, -h), (0, 2 * h, 0), (0, 0, 2 * h), emit),                      # x = -h, normal +x
        _quad((-h, h, -h), (2 * h, 0, 0), (0, 0, 2 * h), emit),                       # y = +h, normal -y
        _quad((-h, -h, -h), (0, 0, 2 * h), (2 * h, 0, 0), emit),                      # y = -h, normal +y
    ]
    # the sampling offsets are added in world units on the focus plane (camera.rs:116), up to pixel_radius / 2 = 1 from the
    # pixel centre: the body is large enough that every sample still meets it
    body = scenes.sphere((0, 0, 0), 5.0, scenes.lambertian(0.2, 0.5, 0.8))
    els = [body] + faces
    if attractor == "sphere":  # an emitting ball behind the camera, in view of the body's visible side; seen from outside, i.e. on its front face
        els.append(scenes.sphere((8.0, 4.0, -25.0), 4.0, emit, attractor=True))
    return els


FURNACE_CAM = dict(center=(0.0, 0.0, -10.0), look_at=(0.0, 0.0, 0.0), fov=2.0, focus_dist=10.0)
BODY = np.array((0.2, 0.5, 0.8)) * np.array(L)


@pytest.mark.parametrize("where", WHERE)
def test_furnace_without_attractors_is_exact(where):
    ri = _integrator(_furnace(), scenes.camera(**FURNACE_CAM))
    assert ri.scene.contents.n_attractors == 0
    img, st = _render(ri, where)
    # every primary ray meets the body, every scattered ray a wall: albedo * L * (scattering_pdf / pdf), the ratio of a number to itself
    assert np.allclose(img, BODY, rtol=1e-12, atol=0), np.abs(img / BODY - 1).max()
    assert st.rays == 2 * st.samples


@pytest.mark.parametrize("where", WHERE)
@pytest.mark.parametrize("attractor", ["quad", "sphere"])
def test_furnace_with_an_attractor_is_unbiased(where, attractor):
    ri = _integrator(_furnace(attractor), scenes.camera(**FURNACE_CAM), width=48, height=48, batch=16, max_batches=4)
    assert ri.scene.contents.n_attractors == 1
    img, st = _render(ri, where)
    assert st.rays == 2 * st.samples
    # the weight scattering_pdf / (0.5 p_light + 0.5 p_cos) lies in [0, 2]: 147 456 samples put the mean within 0.3 % (1 sigma)
    mean = img.reshape(-1, 3).mean(axis=0)
    assert np.allclose(mean, BODY, rtol=0.015, atol=0), mean / BODY
    # and it is the mixture that was sampled: single samples do differ from the plain cosine answer
    assert np.abs(img / BODY - 1).max() > 1e-3


def _furnace_walls(h=50.0):
    return _furnace()[1:7]


@pytest.mark.parametrize("where", WHERE)
def test_a_fuzzy_mirror_in_a_furnace_returns_its_albedo_times_l(where):
    """Metal, fuzz 0.3, met at 45 degrees: n . (reflected + fuzz * unit) >= 0.707 - 0.3 > 0, so no sample is absorbed
    (metal.rs:43-51) and whatever direction the fuzz gives ends on an emitting wall."""
    am = (0.9, 0.5, 0.25)
    s = 16.0
    mirror = _quad((-s / 2, -s / 2, s / 2), (0, s, 0), (s, 0, -s), scenes.metal(*am, 0.3))   # normal (-1,0,-1)/sqrt(2)
    ri = _integrator([mirror] + _furnace_walls(), scenes.camera(**FURNACE_CAM))
    img, st = _render(ri, where)
    want = np.array(am) * np.array(L)
    assert np.allclose(img, want, rtol=1e-13, atol=0), np.abs(img / want - 1).max()
    assert st.rays == 2 * st.samples


@pytest.mark.parametrize("where", WHERE)
def test_a_glass_ball_in_a_furnace_is_invisible(where):
    """Dielectric attenuation is white (dielectric.rs:64): reflected or refracted, every path ends on a wall with throughput 1.
    (A ray refracted into the ball leaves it at once -- Sphere::hit only takes the nearer root, sphere.rs:120-124 -- so the
    depth limit is never reached.)"""
    ri = _integrator([scenes.sphere((0, 0, 0), 5.0, scenes.dielectric(1.5))] + _furnace_walls(), scenes.camera(**FURNACE_CAM))
    img, st = _render(ri, where)
    assert np.allclose(img, np.array(L), rtol=1e-13, atol=0), np.abs(img - np.array(L)).max()   # the weighted mean of equal samples rounds
    assert 2 * st.samples <= st.rays <= 4 * st.samples


@pytest.mark.parametrize("where", WHERE)
def test_a_glass_pane_reflects_by_schlick(where):
    """A dielectric quad seen head-on: cos(theta) ~ 1, so the path reflects with probability r0 + (1 - r0)(1 - cos)^5,
    r0 = ((1 - 1/1.5) / (1 + 1/1.5))^2 = 0.04 (dielectric.rs:41-60, metal.rs:70-76), back to a light behind the camera, and otherwise
    goes straight on (refract_ray at normal incidence) to a light behind the pane."""
    la, lb = (1.0, 0.5, 0.25), (0.0, 2.0, 4.0)
    pane = _quad((-50, -50, 0), (0, 100, 0), (100, 0, 0), scenes.dielectric(1.5))               # normal -z: front face to the camera
    behind_pane = _quad((-500, -500, 40), (0, 1000, 0), (1000, 0, 0), scenes.diffuse_light(*la))  # normal -z, met by rays along +z
    behind_camera = _quad((-500, -500, -40), (1000, 0, 0), (0, 1000, 0), scenes.diffuse_light(*lb))  # normal +z, met by rays along -z
    ri = _integrator([pane, behind_pane, behind_camera], scenes.camera(**CAM), width=64, height=64, batch=16, max_batches=4)
    img, st = _render(ri, where)
    assert st.rays == 2 * st.samples
    mean = img.reshape(-1, 3).mean(axis=0)
    # rays leave the camera within 2.5 degrees (+ the sampling offset) of the axis: (1 - cos)^5 < 1e-13
    r0 = ((1 - 1 / 1.5) / (1 + 1 / 1.5)) ** 2
    want = (1 - r0) * np.array(la) + r0 * np.array(lb)
    # 262 144 samples: the reflected share 0.04 is known to +-0.0004 (1 sigma)
    assert np.allclose(mean, want, rtol=0, atol=4 * 0.0004 * np.abs(np.array(lb) - np.array(la)).max()), (mean, want)
    share = (mean[2] - la[2]) / (lb[2] - la[2])
    assert abs(share - r0) < 0.002, share
