"""RT_FLAG_REUSE_PRIMARY_HITS: the reference restarts the Sobol index with every batch (integrator.rs:284-289), so without a
lens sample i of a pixel sends the same primary ray in every batch.  With the flag the device traces it once per round of
batches; everything a caller can see of the image must stay exactly as it is without the flag, and the counters must tell
what was really traced."""
import numpy as np
import pytest

from raytracer_amd import abi
from oracle import oracle

from helpers import assert_image_parity, c1, knot_scene, teapot_scene

pytestmark = pytest.mark.gpu


def both(res, cfg):
    plain, st_plain = res.render(cfg, want_rgb8=True)
    flagged = cfg.copy()
    flagged.flags |= abi.RT_FLAG_REUSE_PRIMARY_HITS
    reuse, st_reuse = res.render(flagged, want_rgb8=True)
    for k in plain:
        assert np.array_equal(plain[k], reuse[k], equal_nan=True), k
    assert st_reuse.rays == st_plain.rays and st_reuse.samples == st_plain.samples
    assert list(st_reuse.rays_by_depth)[1:] == list(st_plain.rays_by_depth)[1:]
    return plain, st_plain, reuse, st_reuse


def test_same_image_fewer_primary_queries_on_the_adaptive_scene(tmp_scene_dir):
    ri = knot_scene(tmp_scene_dir, 2000, 48, 40, 20, 30, 2)
    cfg = ri.config.copy()
    res = ri.resident(0)
    plain, st_plain, _, st_reuse = both(res, cfg)
    # every round traces its primary rays once: rounds x active pixels x S instead of batches x pixels x S
    assert st_reuse.rays_by_depth[0] < 0.5 * st_plain.rays_by_depth[0]
    assert st_reuse.rays_by_depth[0] >= 48 * 40 * 20  # at least once per pixel and sample
    assert st_reuse.rays_traced == st_plain.rays_traced - (st_plain.rays_by_depth[0] - st_reuse.rays_by_depth[0])
    flagged = cfg.copy()
    flagged.flags |= abi.RT_FLAG_REUSE_PRIMARY_HITS
    out_o, st_o = oracle.render(ri.scene, cfg, 8)
    out_g, _ = res.render(flagged)
    assert_image_parity(out_g, out_o, "reuse vs oracle", res, flagged)
    assert st_reuse.rays == st_o.rays  # the queries the reference makes


def test_round_shapes_and_chunks(tmp_scene_dir):
    """One batch per round (nothing to share), rounds cut small, pixel chunks inside a round, a round of 64 batches."""
    ri = knot_scene(tmp_scene_dir, 2000, 40, 40, 8, 40, 1)
    cfg = ri.config.copy()
    res = ri.resident(0)
    # (a working memory of 20 000 paths holds one batch of the 1 600 pixels x 8 samples at a time, 30 000 hold two)
    for shares, tune in ((False, dict(max_group=1, first_group=1)), (True, dict(max_group=5, first_group=3)), (True, dict(max_group=64, first_group=64)),
                         (False, dict(max_group=64, first_group=8, max_paths=20000)), (False, dict(max_paths=5000)), (True, dict(max_paths=30000))):
        res.tune(**tune)
        _, st_plain, _, st_reuse = both(res, cfg)
        if shares:
            assert st_plain.rays_by_depth[0] > st_reuse.rays_by_depth[0], tune
        else:  # rounds of one batch have nothing to share
            assert st_plain.rays_by_depth[0] == st_reuse.rays_by_depth[0], tune
    res.tune(max_group=64, first_group=8, max_paths=0)


def test_batches_larger_than_a_workgroup_share(tmp_scene_dir):
    """More than 512 samples per batch take the other ray-generation kernel."""
    ri = knot_scene(tmp_scene_dir, 600, 12, 10, 600, 4, 1)
    _, st_plain, _, st_reuse = both(ri.resident(0), ri.config.copy())
    assert st_reuse.rays_by_depth[0] < st_plain.rays_by_depth[0]


def test_fixed_sample_counts_spheres_and_the_teapot():
    ri = c1(width=40, height=36, batch=16, max_batches=6, min_batches=6)
    _, st_plain, _, st_reuse = both(ri.resident(0), ri.config.copy())
    assert st_reuse.rays_by_depth[0] == 40 * 36 * 16  # one round, every ray once
    assert st_plain.rays_by_depth[0] == 6 * 40 * 36 * 16
    rt = teapot_scene(64, 64, 16, 8, 2)
    both(rt.resident(0), rt.config.copy())


def test_a_lens_switches_it_off():
    """With a lens every batch draws its own origin (camera.rs:104-112): nothing is shared, the flag changes nothing."""
    ri = c1(width=32, height=32, batch=8, max_batches=4, min_batches=4, defocus=1.5)
    _, st_plain, _, st_reuse = both(ri.resident(0), ri.config.copy())
    assert st_reuse.rays_by_depth[0] == st_plain.rays_by_depth[0] and st_reuse.rays_traced == st_plain.rays_traced


def test_row_shards_and_the_reference_visit_rule(tmp_scene_dir):
    ri = knot_scene(tmp_scene_dir, 2000, 40, 37, 8, 12, 2)
    res = ri.resident(0)
    cfg = ri.config.copy()
    cfg.row_first, cfg.row_stride, cfg.row_block = 1, 3, 2
    both(res, cfg)
    cfg = ri.config.copy()
    cfg.flags |= abi.RT_FLAG_REFERENCE_VISIT_RULE | abi.RT_FLAG_COUNT_VISITS
    both(res, cfg)
    cfg = ri.config.copy()
    cfg.max_depth = 1
    both(res, cfg)
