"""Random scenes of mixed primitives: the device's closest hits must equal the oracle's exactly.

Each seed draws spheres (some overlapping, some duplicated, some with a negative radius, which the reference's
bounding-box constructor allows, sphere.rs:73-74), axis-aligned and tilted quads (the latter stick out of the
reference's boxes, quad.rs:86: none, a few -- the two-pass walk -- or many -- the reference's rule for every
ray), triangles (some degenerate, some duplicated, some coplanar with a quad), one box, and both split methods."""
import numpy as np
import pytest

from helpers import assert_image_parity, random_rays
from oracle import oracle
from raytracer_amd import RenderIntegrator, scenes

pytestmark = pytest.mark.gpu


def _material(rng):
    k = rng.randint(4)
    if k == 0:
        return scenes.lambertian(*rng.uniform(0.1, 0.9, 3))
    if k == 1:
        return scenes.metal(*rng.uniform(0.1, 0.9, 3), rng.uniform(0.0, 0.5))
    if k == 2:
        return scenes.dielectric(rng.uniform(1.1, 2.0))
    return scenes.diffuse_light(*rng.uniform(1.0, 10.0, 3))


def _scene(seed):
    rng = np.random.RandomState(seed)
    els = []
    n_sph, n_tri, n_quad = rng.randint(0, 40), rng.randint(0, 200), rng.randint(1, 25)
    tilted = [0, 3, 40][seed % 3]
    for k in range(n_sph):
        c = rng.uniform(50, 500, 3)
        r = rng.uniform(5, 90) * (-1.0 if rng.uniform() < 0.1 else 1.0)
        els.append(scenes.sphere(c, r, _material(rng), attractor=bool(rng.uniform() < 0.2)))
        if rng.uniform() < 0.15:
            els.append(scenes.sphere(c, r, _material(rng)))  # exact duplicate
    for k in range(n_tri):
        v0 = rng.uniform(0, 555, 3)
        e1, e2 = rng.uniform(-120, 120, 3), rng.uniform(-120, 120, 3)
        if rng.uniform() < 0.05:
            e2 = e1 * rng.uniform(0.2, 2.0)  # zero area
        t = {"JSONTriangle": {"v0": scenes._v(*v0), "v1": scenes._v(*(v0 + e1)), "v2": scenes._v(*(v0 + e2)), "material": _material(rng)}}
        els.append(t)
        if rng.uniform() < 0.1:
            els.append(t)
    for k in range(n_quad + tilted):
        q = rng.uniform(0, 500, 3)
        if k < n_quad:  # axis-aligned: u and v along two different axes
            a, b = rng.choice(3, 2, replace=False)
            u, v = np.zeros(3), np.zeros(3)
            u[a], v[b] = rng.uniform(-200, 200), rng.uniform(-200, 200)
        else:
            u, v = rng.uniform(-150, 150, 3), rng.uniform(-150, 150, 3)
        quad = {"q": scenes._v(*q), "u": scenes._v(*u), "v": scenes._v(*v), "material": _material(rng)}
        if rng.uniform() < 0.3:
            quad["attractor"] = True
        els.append({"JSONQuad": quad})
        if k == 0:  # a triangle lying in the first quad's plane, sharing its corner
            els.append({"JSONTriangle": {"v0": scenes._v(*q), "v1": scenes._v(*(q + u)), "v2": scenes._v(*(q + v)), "material": _material(rng)}})
    els.append({"JSONBox": {"a": scenes._v(0, 0, 0), "b": scenes._v(*rng.uniform(40, 200, 3)), "material": _material(rng),
                            "transpose": scenes._v(*rng.uniform(0, 350, 3)),
                            "rotate": {"theta_x": rng.uniform(-1, 1), "theta_y": rng.uniform(-1, 1), "theta_z": rng.uniform(-1, 1)}}})
    rng.shuffle(els)
    split = "SAH" if seed % 2 == 0 else "Mid"
    cam = scenes.camera(defocus_angle=0.8 if seed % 4 == 1 else 0.0, focus_dist=900.0)  # thin lens on some seeds
    depth = [8, 1, 3, 12][seed % 4]
    return RenderIntegrator.new({"camera": cam, "elements": els}, scenes.config(40, 40, 4, 1, 1, split=split, max_depth=depth)).prepare()


@pytest.mark.parametrize("seed", range(24))
def test_random_scene_closest_hits(seed):
    ri = _scene(seed)
    res = ri.resident(0)
    o, d = random_rays(60_000, seed=100 + seed)
    pg, tg = res.trace_rays(o, d)
    po, to = oracle.trace_rays(ri.scene, o, d)
    assert (po >= 0).mean() > 0.05
    bad = np.flatnonzero((pg != po) | (tg != to))
    assert bad.size == 0, f"seed {seed}: {bad.size} rays differ, first {bad[:5]}: device {pg[bad[:5]]} {tg[bad[:5]]} oracle {po[bad[:5]]} {to[bad[:5]]}"
    cfg = ri.config.copy()
    pg, tg = res.primary_hits(cfg)
    po, to = oracle.primary_hits(ri.scene, cfg)
    assert np.array_equal(pg, po) and np.array_equal(tg, to)


@pytest.mark.parametrize("seed", range(9))
def test_random_scene_images(seed):
    """Whole renders of the same scenes (several attractors of both kinds, every material, three batches with the
    z-test active): same per-path streams on both sides, so the images agree except where a last-bit difference
    in sin/cos moves a path across an edge."""
    ri = _scene(seed)
    cfg = ri.config.copy()
    cfg.sample_batch_size, cfg.max_sample_batches, cfg.min_sample_batches = 8, 4, 0
    out_g, st_g = ri.resident(0).render(cfg)
    out_o, st_o = oracle.render(ri.scene, cfg, 8)
    # 1 600 pixels at 32 spp under lights of radiance up to 10: ONE flipped path that ends on a light is a tenth of
    # so small an image's whole noise, hence the wider share here (the gates on pixel and sample counts stay)
    assert_image_parity(out_g, out_o, f"seed {seed}", ri.resident(0), cfg, noise_share=0.25)
    assert abs(int(st_g.rays) - int(st_o.rays)) <= 1e-4 * st_o.rays + 8
