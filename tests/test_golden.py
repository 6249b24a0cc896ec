"""Golden fixtures (tests/golden/*.npz, written by tests/golden/make_golden.py).

They are regression pins produced by the CPU oracle -- the reference itself cannot be
run here and its tests hold no vectors for these stages.  On CPU they guard the oracle
against accidental change (sqrt, division and the bit operations are exact, so primary
hits and the integer / sampler vectors must match to the bit; images go through libm's
sin / cos / pow, so they are compared to 1e-12).  On the GPU the same vectors are
checked through the C ABI without running the oracle."""
import os

import numpy as np
import pytest

from raytracer_amd import abi
from oracle import oracle

from helpers import GOLDEN, assert_image_parity, c1, knot_scene


def _load(name):
    return np.load(os.path.join(GOLDEN, name))


def test_oracle_primary_hits_golden():
    g = _load("c1_primary_32x32x4.npz")
    ri = c1(width=32, height=32, batch=4)
    prim, t = oracle.primary_hits(ri.scene, ri.config.copy())
    assert np.array_equal(prim, g["prim"]) and np.array_equal(t, g["t"])
    assert (prim >= 0).mean() > 0.7 and len(np.unique(prim)) > 10


def test_oracle_render_golden():
    g = _load("c1_render_24x24x16.npz")
    ri = c1(width=24, height=24, batch=16)
    cfg = ri.config.copy()
    ctr, st = oracle.render(ri.scene, cfg, 3)
    assert np.allclose(ctr["linear"], g["ctr_linear"], rtol=1e-12, atol=1e-12)
    assert np.allclose(ctr["pixels"], g["ctr_pixels"], rtol=1e-12, atol=1e-12)
    assert st.rays == int(g["ctr_rays"])
    cfg.rng_mode = abi.RT_RNG_REF
    ref, st = oracle.render(ri.scene, cfg, 2)  # thread count must not matter: one RNG per row
    assert np.allclose(ref["linear"], g["ref_linear"], rtol=1e-12, atol=1e-12)
    assert st.rays == int(g["ref_rays"])


def test_oracle_adaptive_golden(tmp_scene_dir):
    g = _load("knot600_adaptive_20x20.npz")
    ri = knot_scene(tmp_scene_dir, 600, 20, 20, 4, 10, 1)
    out, st = oracle.render(ri.scene, ri.config.copy(), 4)
    assert np.array_equal(out["samples"], g["samples"])
    assert np.allclose(out["linear"], g["linear"], rtol=1e-12, atol=1e-12)
    assert np.array_equal(out["sample_pixels"], g["sample_pixels"])
    # heat-map formula, integrator.rs:333-337
    factor = (out["samples"].astype(np.float64) - 1 * 4) / (4 * 10)
    assert np.array_equal(out["sample_pixels"][..., 0], factor) and np.array_equal(out["sample_pixels"][..., 1], 1.0 - factor)


def test_oracle_sampler_golden():
    g = _load("samplers.npz")
    sob = np.array([[[oracle.sobol_sample(i, d, s) for d in (0, 1)] for i in range(32)] for s in (0, 1, 12345, 358801)], dtype=np.float32)
    assert np.array_equal(sob, g["sobol"])
    assert oracle.rng_stream(abi.RT_RNG_REF, 42, 0, 16) == [int(x) for x in g["xoshiro_seed42"]]
    assert oracle.rng_stream(abi.RT_RNG_CTR, 42, 0, 16) == [int(x) for x in g["counter_pixel42"]]
    assert oracle.rng_stream(abi.RT_RNG_REF, 42, 1, 16) == list(g["gen_range_seed42"])


def test_native_build_of_the_oracle_matches_the_goldens(tmp_scene_dir):
    """The benchmark's CPU baseline runs the oracle compiled with -march=native on the machine it measures; the flags
    license no other rounding (no contraction, no fast-math), so every golden vector must come out the same."""
    with oracle.native_build():
        assert "-march=native" in oracle.flags(), "this machine has gcc: the native build must succeed here"
        test_oracle_primary_hits_golden()
        test_oracle_render_golden()
        test_oracle_adaptive_golden(tmp_scene_dir)
        test_oracle_sampler_golden()
        g = _load("c1_render_24x24x16.npz")
        ri = c1(width=24, height=24, batch=16)
        ctr, _ = oracle.render(ri.scene, ri.config.copy(), 3)
    base, _ = oracle.render(ri.scene, ri.config.copy(), 3)
    assert "-march=native" not in oracle.flags()
    for k in base:  # and bit for bit the in-tree build's image (libm is the same library either way)
        assert np.array_equal(base[k], ctr[k]), k


@pytest.mark.gpu
def test_gpu_primary_hits_golden():
    g = _load("c1_primary_32x32x4.npz")
    ri = c1(width=32, height=32, batch=4)
    prim, t = ri.resident(0).primary_hits(ri.config.copy())
    assert np.array_equal(prim, g["prim"]) and np.array_equal(t, g["t"])


@pytest.mark.gpu
def test_gpu_render_golden():
    g = _load("c1_render_24x24x16.npz")
    ri = c1(width=24, height=24, batch=16)
    cfg = ri.config.copy()
    out, st = ri.resident(0).render(cfg)
    assert_image_parity(out, {"linear": g["ctr_linear"]}, "golden C1 24x24x16", ri.resident(0), cfg)
    assert abs(int(st.rays) - int(g["ctr_rays"])) <= 1e-4 * int(g["ctr_rays"])
    # against the reference-RNG image only the statistics can agree
    lum = lambda a: (a * np.array([0.2126, 0.7152, 0.0722])).sum(-1)
    noise = np.sqrt(((lum(g["ctr_linear"]) - lum(g["ref_linear"])) ** 2).mean())
    rmse = np.sqrt(((lum(out["linear"]) - lum(g["ref_linear"])) ** 2).mean())
    assert rmse <= 1.05 * noise + 1e-9


@pytest.mark.gpu
def test_gpu_adaptive_golden(tmp_scene_dir):
    g = _load("knot600_adaptive_20x20.npz")
    ri = knot_scene(tmp_scene_dir, 600, 20, 20, 4, 10, 1)
    cfg = ri.config.copy()
    out, st = ri.resident(0).render(cfg)
    assert_image_parity(out, {"linear": g["linear"], "samples": g["samples"]}, "golden adaptive knot 20x20", ri.resident(0), cfg)
