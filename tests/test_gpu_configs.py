"""BASELINE configs[1], configs[2] (on the reference's own scene file) and configs[3] on the device, against the CPU oracle.

configs[1]  input/scene_teapot.json (the reference's own files, tests/golden/ref_inputs): teapot.obj
            import, 800x800, fixed 256 spp (16 batches of 16, all mandatory), light-PDF importance
            sampling through the quad light's attractor.
configs[2]  input/scene.json, the scene of the README render, as the reference ships it; the dragon.obj it names is not
            distributed, so a generated 20 000-triangle mesh stands under that name.  input/config.json's sampling (batches
            of 20, z-test after 2) at 240x240 with at most 12 batches.  (The full-size workload with 100 000 triangles is
            tests/test_gpu_fullsize.py.)
configs[3]  multi-million-triangle mesh (xl_dragon.obj is not distributed: the generated knot at
            2.2 M triangles, above the size at which the library stores its nodes in line-sized blocks),
            2048x2048, 1024 spp (64 batches of 16, z-test after 4).
"""
import os

import numpy as np
import pytest

from raytracer_amd import RenderIntegrator, abi, parallel, scenes
from oracle import oracle

from helpers import assert_image_parity, noise_gate, readme_scene, teapot_scene

pytestmark = pytest.mark.gpu


# ------------------------------------------------------------------ configs[1]
@pytest.fixture(scope="module")
def teapot():
    ri = teapot_scene(800, 800, 16, 16, 16)
    res = ri.resident(0)
    cfg = ri.config.copy()
    out, st = res.render(cfg)
    yield ri, res, cfg, out, st
    ri.close()  # the scene's device memory goes with the module, not with the garbage collector


def test_c2_teapot_primary_hits_of_the_whole_first_batch(teapot):
    ri, res, cfg, out, st = teapot
    pg, tg = res.primary_hits(cfg)
    po, to = oracle.primary_hits_threads(ri.scene, cfg, 16)
    assert pg.shape == (800, 800, 16)
    assert np.array_equal(pg, po), f"{(pg != po).sum()} of {pg.size} primary hits name another element"
    assert np.array_equal(tg, to)
    # the picture is what the scene file describes: teapot, box, walls and light are all seen
    seen = np.unique(po)
    assert (seen < 6320).sum() > 1500 and 6344 in seen and ((seen >= 6320) & (seen < 6332)).any() and (po >= 0).mean() > 0.85


def test_c2_teapot_rows_match_the_oracle(teapot):
    ri, res, cfg, out, st = teapot
    assert ri.scene.contents.n_attractors == 1                      # the mixture PDF is active (integrator.rs:381-391)
    assert (out["samples"] == 256).all() and st.samples == 800 * 800 * 256
    rows = cfg.copy()
    rows.row_first, rows.row_stride = 160, 200                      # rows 160, 360, 560, 760 (the last through the teapot)
    o, so = oracle.render(ri.scene, rows, 16)
    g = {k: out[k][160::200] for k in ("linear", "samples", "pixels", "sample_pixels")}
    p = assert_image_parity(g, o, "teapot rows 160::200", res, rows)
    assert np.isclose(g["pixels"], o["pixels"], rtol=0, atol=1e-9).all(axis=-1).mean() >= 0.999
    assert np.array_equal(g["sample_pixels"], o["sample_pixels"])
    # the same rows rendered as a shard of their own give the same pixels and exactly the oracle's ray count scale
    shard, ss = res.render(rows)
    assert np.array_equal(shard["linear"], g["linear"])
    assert abs(int(ss.rays) - int(so.rays)) <= 1e-4 * so.rays, (ss.rays, so.rays, p)


def test_c2_teapot_agrees_with_the_reference_rng_mode_within_noise(teapot):
    ri, res, cfg, out, st = teapot
    rows = cfg.copy()
    rows.row_first, rows.row_stride = 40, 80                        # 10 rows
    ref = rows.copy()
    ref.rng_mode = abi.RT_RNG_REF
    o, _ = oracle.render(ri.scene, ref, 16)
    other = rows.copy()
    other.seed = 777
    b, _ = res.render(other)
    a = out["linear"][40::80]
    noise_gate(a, b["linear"], o["linear"], 1.10)
    assert abs(a.mean() - o["linear"].mean()) <= 0.01 * o["linear"].mean()


# ------------------------------------------------------------------ configs[2], the reference's scene file
def test_c3_scene_file_of_the_reference(tmp_scene_dir):
    ri = readme_scene(tmp_scene_dir, 20000, 240, 240, 20, 12, 2)
    res = ri.resident(0)
    cfg = ri.config.copy()
    assert ri.scene.contents.n_attractors == 1 and ri.scene.contents.n_quads == 1
    pg, tg = res.primary_hits(cfg)
    po, to = oracle.primary_hits_threads(ri.scene, cfg, 16)
    assert np.array_equal(pg, po), f"{(pg != po).sum()} of {pg.size} primary hits name another element"
    assert np.array_equal(tg, to)
    n_mesh = ri.scene.contents.n_triangles - 24
    seen = np.unique(po)
    assert (seen < n_mesh).sum() > 500 and (po >= 0).mean() > 0.85  # the mesh is in the picture
    out, st = res.render(cfg)
    o, so = oracle.render(ri.scene, cfg, 16)
    p = assert_image_parity(out, o, "input/scene.json", res, cfg)
    assert np.array_equal(out["sample_pixels"], o["sample_pixels"]) and st.samples == so.samples
    assert 2 * 20 < out["samples"].mean() < 12 * 20 and (out["samples"] < 12 * 20).any()  # the z-test stopped pixels early
    ri.close()


# ------------------------------------------------------------------ configs[3]
C4_TRIANGLES = 2_200_000


@pytest.fixture(scope="module")
def big(tmp_path_factory):
    d = str(tmp_path_factory.mktemp("c4"))
    scene_path, config_path = scenes.materialize_c3(d, C4_TRIANGLES, 2048, 2048, batch=16, max_batches=64, min_batches=2)
    ri = RenderIntegrator.new_from_json(scene_path, config_path).prepare()
    res = ri.resident(0)
    cfg = ri.config.copy()
    out, st = res.render(cfg)
    yield ri, res, cfg, out, st
    ri.close()


def test_c4_shape_uses_the_block_node_order_by_default(big):
    ri, res, cfg, out, st = big
    sc = ri.scene.contents
    assert sc.n_triangles == C4_TRIANGLES + 24
    internal = (sc.n_nodes - 1) // 2
    assert internal * 32 > (64 << 20)                               # above the library's threshold for line-sized node blocks
    assert "RT_HIP_NODE_ORDER" not in os.environ
    assert 20 <= ri.bvh_depth() <= 64


def test_c4_shape_sample_accounting(big):
    ri, res, cfg, out, st = big
    S, lo, hi = cfg.sample_batch_size, cfg.min_sample_batches + 2, cfg.max_sample_batches
    n = out["samples"].astype(np.int64)
    assert n.shape == (2048, 2048)
    assert (n % S == 0).all() and n.min() >= lo * S and n.max() == hi * S
    assert int(n.sum()) == st.samples
    assert st.rays <= st.rays_traced <= 1.05 * st.rays
    assert st.samples <= st.rays <= cfg.max_depth * st.samples
    factor = (n - cfg.min_sample_batches * S) / float(S * hi)
    assert np.array_equal(out["sample_pixels"][..., 0], factor) and np.array_equal(out["sample_pixels"][..., 1], 1.0 - factor)
    assert np.allclose(out["pixels"], np.clip(out["linear"], 0.0, 1.0) ** (1.0 / cfg.gamma_correction), rtol=1e-12, atol=0)
    assert np.isfinite(out["linear"]).all() and (out["linear"] >= 0).all()


def test_c4_shape_rows_and_primary_hits_match_the_oracle(big):
    ri, res, cfg, out, st = big
    rows = cfg.copy()
    rows.row_first, rows.row_stride = 700, 800                      # rows 700 and 1500: through the mesh and the glass sphere
    o, so = oracle.render(ri.scene, rows, 16)
    g = {k: out[k][700::800] for k in ("linear", "samples")}
    assert_image_parity(g, o, "C4-shape rows 700, 1500", res, rows)
    assert o["samples"].min() < o["samples"].max()                  # the z-test is at work in these rows
    every = cfg.copy()
    every.row_first, every.row_stride = 5, 32                       # 64 rows of primary rays
    pg, tg = res.primary_hits(every)
    po, to = oracle.primary_hits_threads(ri.scene, every, 16)
    assert np.array_equal(pg, po) and np.array_equal(tg, to)
    assert ((po >= 0) & (po < C4_TRIANGLES)).mean() > 0.02          # the mesh is in view


def test_c4_shape_is_shard_independent(big):
    ri, res, cfg, out, st = big
    shard = parallel.shard_config(cfg, 5, 8)
    o, s = res.render(shard)
    assert np.array_equal(o["linear"], out["linear"][5::8]) and np.array_equal(o["samples"], out["samples"][5::8])


def test_c4_shape_on_wide_nodes_host_built_and_device_built(big, monkeypatch):
    """2.2 M internal nodes: the four-wide records (forced: by default only trees beyond the caches get them) are collapsed subtree
    by subtree on several host threads, from the uploaded tree and from the tree built on the device.  Hits and image rows as on
    the binary records."""
    from raytracer_amd.integrator import ResidentScene
    ri, res, cfg, out, st = big
    monkeypatch.setenv("RT_HIP_WIDE", "1")
    every = cfg.copy()
    every.row_first, every.row_stride = 5, 32
    p0, t0 = res.primary_hits(every)
    rng = np.random.RandomState(11)
    o = rng.uniform(5.0, 550.0, size=(300_000, 3))
    d = rng.normal(size=(300_000, 3))
    r0, s0 = res.trace_rays(o, d)
    wide = ResidentScene(ri.scene, 0)
    p1, t1 = wide.primary_hits(every)
    r1, s1 = wide.trace_rays(o, d)
    assert np.array_equal(p0, p1) and np.array_equal(t0, t1) and np.array_equal(r0, r1) and np.array_equal(s0, s1)
    shard = parallel.shard_config(cfg, 3, 64)
    a, sa = wide.render(shard)
    assert np.array_equal(a["linear"], out["linear"][3::64]) and np.array_equal(a["samples"], out["samples"][3::64])
    counted = shard.copy()
    counted.flags = abi.RT_FLAG_COUNT_VISITS
    _, sw = wide.render(counted)
    _, sb = res.render(counted)
    assert sw.node_visits / sw.rays_traced < 0.7 * sb.node_visits / sb.rays_traced
    wide.close()
    dev = ResidentScene(ri.scene, 0, build_on_device=True)
    r2, s2 = dev.trace_rays(o, d)
    assert np.array_equal(s0, s2) and np.array_equal(r0, r2)        # the device builds the host's tree: same hits, ids included
    b, _ = dev.render(shard)
    assert np.array_equal(b["linear"], a["linear"]) and np.array_equal(b["samples"], a["samples"])
    _, sd = dev.render(counted)
    assert sd.node_visits / sd.rays_traced < 0.7 * sb.node_visits / sb.rays_traced
    dev.close()
