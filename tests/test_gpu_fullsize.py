"""Properties of the device path at the benchmark's full size (1200x1200, batches of 20, up to
100 batches, 100 000-triangle mesh), where the CPU oracle would need minutes: they do not depend
on the size, so they are checked where the queues, groups of batches and counters are largest."""
import os

import numpy as np
import pytest

from raytracer_amd import RenderIntegrator, abi, parallel, scenes
from oracle import oracle

from helpers import assert_image_parity

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def full(tmp_path_factory):
    d = str(tmp_path_factory.mktemp("c3"))
    scene_path, config_path = scenes.materialize_c3(d)
    ri = RenderIntegrator.new_from_json(scene_path, config_path).prepare()
    res = ri.resident(0)
    cfg = ri.config.copy()
    out, st = res.render(cfg)
    yield ri, res, cfg, out, st
    ri.close()  # the scene's device memory goes with the module, not with the garbage collector


def test_full_size_sample_accounting(full):
    ri, res, cfg, out, st = full
    S, lo, hi = cfg.sample_batch_size, cfg.min_sample_batches + 2, cfg.max_sample_batches
    n = out["samples"].astype(np.int64)
    assert (n % S == 0).all() and n.min() >= lo * S and n.max() == hi * S      # first z-test after min+2 batches
    assert int(n.sum()) == st.samples
    assert st.rays <= st.rays_traced <= 1.05 * st.rays                          # speculative batches stay a few percent
    assert st.samples <= st.rays <= cfg.max_depth * st.samples
    assert int(sum(st.rays_by_depth)) == st.rays_traced and st.rays_by_depth[0] >= st.samples
    # heat map = (factor, 1-factor, 0) with factor from the sample count (integrator.rs:333-337)
    factor = (n - cfg.min_sample_batches * S) / float(S * hi)
    assert np.array_equal(out["sample_pixels"][..., 0], factor) and np.array_equal(out["sample_pixels"][..., 1], 1.0 - factor)
    assert (out["sample_pixels"][..., 2] == 0).all()
    # final pixels are the clamped, gamma-encoded linear means (integrator.rs:331)
    assert np.allclose(out["pixels"], np.clip(out["linear"], 0.0, 1.0) ** (1.0 / cfg.gamma_correction), rtol=1e-12, atol=0)
    assert np.isfinite(out["linear"]).all() and (out["linear"] >= 0).all()
    assert 0.85 * hi * S <= n.mean() <= hi * S                                  # most of this scene needs all its samples


def test_full_size_is_reproducible_and_shard_independent(full):
    ri, res, cfg, out, st = full
    again, st2 = res.render(cfg)
    for k in ("pixels", "sample_pixels", "linear", "samples"):
        assert np.array_equal(out[k], again[k]), k
    assert (st2.rays, st2.samples) == (st.rays, st.samples)
    # rows dealt to 3 ranks: each shard alone gives exactly its rows of the full image
    parts = []
    rays = 0
    for r in range(3):
        shard = parallel.shard_config(cfg, r, 3)
        o, s = res.render(shard)
        parts.append(o)
        rays += s.rays
        assert np.array_equal(o["linear"], out["linear"][r::3])
        assert np.array_equal(o["samples"], out["samples"][r::3])
    assert rays == st.rays


def test_full_size_reference_visit_rule_gives_the_same_image(full):
    """Walking the tree exactly as the reference does (left first, no culling) costs more node
    visits but must not change a single pixel."""
    ri, res, cfg, out, st = full
    slow = cfg.copy()
    slow.flags = abi.RT_FLAG_REFERENCE_VISIT_RULE | abi.RT_FLAG_COUNT_VISITS
    o, s = res.render(slow)
    assert np.array_equal(o["linear"], out["linear"]) and np.array_equal(o["samples"], out["samples"])
    assert s.rays == st.rays
    fast = cfg.copy()
    fast.flags = abi.RT_FLAG_COUNT_VISITS
    _, f = res.render(fast)
    assert f.node_visits < s.node_visits and f.leaf_tests < s.leaf_tests


def test_full_size_rows_match_the_oracle(full):
    """Three rows of the full-size image (through the mesh, the glass sphere and the floor) traced by
    the CPU oracle with the same per-path random streams: same sample counts, same pixels."""
    ri, res, cfg, out, st = full
    rows = cfg.copy()
    rows.row_first, rows.row_stride = 250, 400          # rows 250, 650, 1050
    o, so = oracle.render(ri.scene, rows, 16)
    g = {k: out[k][250::400] for k in ("linear", "samples", "sample_pixels")}
    assert_image_parity(g, o, "C3 rows 250, 650, 1050", res, rows)
    assert abs(int(so.rays) - int(st.rays) * 3 // 1200) < 0.2 * so.rays


def test_full_size_agrees_with_the_reference_rng_mode_within_noise(full):
    """The north star's third gate at the configuration the metric is quoted on: against the oracle in the
    reference's own random-number mode (one xoshiro256++ per image row, integrator.rs:263 -- a different sample
    sequence, so only statistics can agree) on 32 evenly spaced rows of the image (tests/golden/c3_ref_rows.npz,
    written by tests/golden/make_golden.py).  Bound: the device's image may differ from the reference-mode image
    by no more than 1.05 x what two device renders with different seeds differ by (pure Monte-Carlo noise at the
    same sample counts); mean radiance within 1e-3 relative; mean samples per pixel within 0.25 % (every pixel's
    stopping batch is a random variable of its sample sequence: over these 38 400 pixels the two modes differ by
    0.11 %, over the whole image by 0.002 %)."""
    from helpers import GOLDEN, noise_gate
    ri, res, cfg, out, st = full
    g = np.load(os.path.join(GOLDEN, "c3_ref_rows.npz"))
    first, stride = int(g["row_first"]), int(g["row_stride"])
    ref_linear, ref_samples = g["linear"].astype(np.float64), g["samples"].astype(np.int64)
    a = out["linear"][first::stride]
    assert a.shape == ref_linear.shape == (32, 1200, 3)
    other = cfg.copy()
    other.row_first, other.row_stride, other.seed = first, stride, 20261111
    b, _ = res.render(other)
    rmse_ref, rmse_self = noise_gate(a, b["linear"], ref_linear, 1.05)
    assert rmse_self > 1e-4                                   # the comparison is between noisy estimates, not zeros
    for c in range(3):
        assert abs(a[..., c].mean() - ref_linear[..., c].mean()) <= 1e-3 * ref_linear[..., c].mean(), c
    spp, spp_ref = out["samples"][first::stride].mean(), ref_samples.mean()
    assert abs(spp - spp_ref) <= 2.5e-3 * spp_ref, (spp, spp_ref)
