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
