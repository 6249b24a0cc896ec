"""BASELINE configs[0] at its stated size and configs[4] in its own regime.

configs[0]  the Cornell scene with spheres, 400x400, fixed 64 spp (one batch of 64): the whole image's primary hits and
            evenly spaced rows against the oracle, the rest by properties.
configs[4]  the triangle soup with glass and metal triangles, a thin lens and the quad light.  What sets that configuration
            apart on the device is chosen by SIZE, not by a switch: above 256 MB of binary node records the tree is stored as
            four-wide nodes, above 256 MB of triangle records those are stored one per 128-byte line, and the wide walk keeps
            22 stack levels in LDS and spills the rest.  8.6 M triangles cross both thresholds with the library's defaults
            (no RT_HIP_* variable set); the image is small (96 x 54, one batch of 8) because the oracle, which walks the
            tree by the reference's rule -- every box along the whole line of a ray -- needs about a millisecond per ray here.
"""
import os

import numpy as np
import pytest

from raytracer_amd import RenderIntegrator, abi, scenes
from oracle import oracle

from helpers import assert_image_parity, c1

pytestmark = pytest.mark.gpu

SOUP_TRIANGLES = 8_600_000


# ------------------------------------------------------------------ configs[0]
@pytest.fixture(scope="module")
def cornell():
    ri = RenderIntegrator.new(scenes.scene_c1(), scenes.config_c1(400, 400)).prepare()
    res = ri.resident(0)
    cfg = ri.config.copy()
    out, st = res.render(cfg)
    yield ri, res, cfg, out, st
    ri.close()


def test_c1_at_its_stated_size_properties(cornell):
    ri, res, cfg, out, st = cornell
    assert (cfg.img_width, cfg.img_height, cfg.sample_batch_size, cfg.max_sample_batches, cfg.min_sample_batches) == (400, 400, 64, 1, 1)
    assert (out["samples"] == 64).all() and st.samples == 400 * 400 * 64
    assert st.rays == st.rays_traced                                   # one mandatory batch: nothing speculative
    assert st.samples <= st.rays <= cfg.max_depth * st.samples
    # heat map of a pixel that took its single batch (integrator.rs:333-337): (64 - 64) / 64 = 0
    assert (out["sample_pixels"][..., 0] == 0.0).all() and (out["sample_pixels"][..., 1] == 1.0).all()
    assert np.allclose(out["pixels"], np.clip(out["linear"], 0.0, 1.0) ** (1.0 / cfg.gamma_correction), rtol=1e-12, atol=0)
    again, st2 = res.render(cfg)
    assert np.array_equal(again["linear"], out["linear"]) and st2.rays == st.rays


def test_c1_at_its_stated_size_against_the_oracle(cornell):
    ri, res, cfg, out, st = cornell
    pg, tg = res.primary_hits(cfg)
    po, to = oracle.primary_hits_threads(ri.scene, cfg, 16)
    assert pg.shape == (400, 400, 64)
    assert np.array_equal(pg, po) and np.array_equal(tg, to)
    rows = cfg.copy()
    rows.row_first, rows.row_stride = 10, 20                           # 20 rows from top to bottom
    o, so = oracle.render(ri.scene, rows, 16)
    g = {k: out[k][10::20] for k in ("linear", "samples", "pixels", "sample_pixels")}
    assert_image_parity(g, o, "C1 rows 10::20", res, rows)
    assert np.array_equal(g["sample_pixels"], o["sample_pixels"])


# ------------------------------------------------------------------ configs[4]
@pytest.fixture(scope="module")
def soup(tmp_path_factory):
    for name in ("RT_HIP_WIDE", "RT_HIP_TRI_PAD", "RT_HIP_NODE_ORDER", "RT_HIP_SPILL_ABOVE", "RT_HIP_NO_GRID"):
        assert name not in os.environ, f"{name} is set: this test is about what the library chooses by itself"
    d = str(tmp_path_factory.mktemp("soup"))
    sp, cp = scenes.materialize_soup(d, SOUP_TRIANGLES, 96, 54, batch=8, max_batches=1, min_batches=1)
    ri = RenderIntegrator.new_from_json(sp, cp).prepare()
    res = ri.resident(0)
    yield ri, res
    ri.close()


def test_c5_regime_is_chosen_by_size(soup):
    ri, res = soup
    sc = ri.scene.contents
    assert sc.n_triangles == SOUP_TRIANGLES + 24 and sc.n_quads == 1 and sc.n_attractors == 1
    internal = (sc.n_nodes - 1) // 2
    assert internal * 32 > (256 << 20), "binary node records must exceed the wide-node threshold"
    assert sc.n_triangles * 80 > (256 << 20), "triangle records must exceed the padding threshold"
    assert sc.camera.defocus_angle > 0.0
    kinds = {ri.scene.contents.materials[i].kind for i in range(sc.n_materials)}
    assert {abi.RT_MAT_LAMBERTIAN, abi.RT_MAT_METAL, abi.RT_MAT_DIELECTRIC, abi.RT_MAT_DIFFUSE_LIGHT} <= kinds
    # the walk on wide nodes fetches far fewer node records per ray than a binary walk can: at most one per two levels
    cfg = ri.config.copy()
    cfg.flags = abi.RT_FLAG_COUNT_VISITS
    _, st = res.render(cfg)
    assert st.node_visits > 0 and st.leaf_tests > 0
    o = np.array([[278.0, 278.0, -800.0]])
    prim, t = res.trace_rays(o, np.array([[0.0, 0.0, 1.0]]))
    assert prim[0] >= 0


def test_c5_regime_hits_match_the_oracle(soup):
    ri, res = soup
    cfg = ri.config.copy()
    pg, tg = res.primary_hits(cfg)                                     # with the lens: origins drawn per sample
    po, to = oracle.primary_hits_threads(ri.scene, cfg, 16)
    assert pg.shape == (54, 96, 8)
    assert np.array_equal(pg, po) and np.array_equal(tg, to)
    assert 0.4 < (po >= 0).mean() < 0.7                                # the 16:9 frame is wider than the room's opening
    rng = np.random.RandomState(5)
    n = 200_000
    o = rng.uniform(5.0, 550.0, size=(n, 3))
    d = rng.normal(size=(n, 3))
    rg, sg = res.trace_rays(o, d)
    ro, so = oracle.trace_rays(ri.scene, o, d)
    assert np.array_equal(rg, ro), f"{(rg != ro).sum()} of {n} rays end on another element"
    assert np.array_equal(sg, so)
    assert len(np.unique(ro)) > 100_000                                # the rays really spread over the soup


def test_c5_regime_image_matches_the_oracle(soup):
    ri, res = soup
    cfg = ri.config.copy()
    out_g, st = res.render(cfg)
    out_o, so = oracle.render(ri.scene, cfg, 16)
    assert_image_parity(out_g, out_o, "8.6 M-triangle soup, 96 x 54 x 8, lens + glass", res, cfg, noise_share=0.25)
    assert st.samples == 96 * 54 * 8 and abs(int(st.rays) - int(so.rays)) <= 1e-3 * so.rays
