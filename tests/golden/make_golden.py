#!/usr/bin/env python3
"""Regenerates tests/golden/*.npz with the CPU oracle.

The reference cannot be run where this repository is built (Rust, no toolchain),
and its own tests hold no vectors for traversal, shading, sampling or pixels, so
these fixtures are REGRESSION PINS produced by the oracle itself: they freeze the
oracle's behaviour (so an accidental change to it is caught on CPU) and give the
GPU tests fixed vectors that do not depend on running the oracle.  They are not
outputs of the reference.

    python tests/golden/make_golden.py [c3]      ("c3": only the headline-workload rows, the slow part)
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from helpers import c1, knot_scene  # noqa: E402
from raytracer_amd import abi  # noqa: E402
from oracle import oracle  # noqa: E402


C3_REF_ROWS = (11, 38)  # rows 11, 49, ..., 1189


def main():
    if "c3" not in sys.argv[1:]:
        small()
    headline()


def small():
    # 1. primary-hit map of the C1 scene, 32x32, 4 samples per pixel
    ri = c1(width=32, height=32, batch=4)
    cfg = ri.config.copy()
    prim, t = oracle.primary_hits(ri.scene, cfg)
    np.savez_compressed(os.path.join(HERE, "c1_primary_32x32x4.npz"), prim=prim, t=t)

    # 2. 24x24 renders of the C1 scene, 16 spp, both RNG modes
    ri = c1(width=24, height=24, batch=16)
    cfg = ri.config.copy()
    ctr, st_ctr = oracle.render(ri.scene, cfg, 4)
    ref_cfg = cfg.copy()
    ref_cfg.rng_mode = abi.RT_RNG_REF
    ref, st_ref = oracle.render(ri.scene, ref_cfg, 4)
    np.savez_compressed(os.path.join(HERE, "c1_render_24x24x16.npz"), ctr_linear=ctr["linear"], ctr_pixels=ctr["pixels"],
                        ref_linear=ref["linear"], ref_pixels=ref["pixels"], ctr_rays=np.int64(st_ctr.rays), ref_rays=np.int64(st_ref.rays))

    # 3. adaptive render (z-test) of a small mesh scene: sample counts and heat-map
    ri = knot_scene(os.path.join(HERE, "_tmp"), 600, 20, 20, 4, 10, 1)
    cfg = ri.config.copy()
    out, st = oracle.render(ri.scene, cfg, 4)
    np.savez_compressed(os.path.join(HERE, "knot600_adaptive_20x20.npz"), linear=out["linear"], samples=out["samples"],
                        sample_pixels=out["sample_pixels"], rays=np.int64(st.rays))

    # 4. sampler / RNG vectors
    sob = np.array([[[oracle.sobol_sample(i, d, s) for d in (0, 1)] for i in range(32)] for s in (0, 1, 12345, 358801)], dtype=np.float32)
    xo = np.array(oracle.rng_stream(abi.RT_RNG_REF, 42, 0, 16), dtype=np.uint64)
    ctr_u = np.array(oracle.rng_stream(abi.RT_RNG_CTR, 42, 0, 16), dtype=np.uint64)
    rng_f = np.array(oracle.rng_stream(abi.RT_RNG_REF, 42, 1, 16, lo=0.0, hi=1.0))
    np.savez_compressed(os.path.join(HERE, "samplers.npz"), sobol=sob, xoshiro_seed42=xo, counter_pixel42=ctr_u, gen_range_seed42=rng_f)


def headline():
    # 5. the headline workload in the reference's own random-number mode: 32 evenly spaced rows of the
    #    1200x1200 README-like scene (100 000-triangle mesh, batches of 20, up to 100, z-test), one
    #    xoshiro256++ per row as integrator.rs:263.  The statistical gate of the GPU tests compares against it.
    from raytracer_amd import RenderIntegrator, scenes  # noqa: E402
    scene_path, config_path = scenes.materialize_c3(os.path.join(HERE, "_tmp"))
    ri = RenderIntegrator.new_from_json(scene_path, config_path).prepare()
    rows = ri.config.copy()
    rows.rng_mode = abi.RT_RNG_REF
    rows.row_first, rows.row_stride = C3_REF_ROWS
    out, st = oracle.render(ri.scene, rows, os.cpu_count() or 4)
    np.savez_compressed(os.path.join(HERE, "c3_ref_rows.npz"), linear=out["linear"].astype(np.float32), samples=out["samples"].astype(np.uint16),
                        row_first=np.int32(rows.row_first), row_stride=np.int32(rows.row_stride), rays=np.int64(st.rays))
    import shutil
    shutil.rmtree(os.path.join(HERE, "_tmp"), ignore_errors=True)
    print("wrote", sorted(f for f in os.listdir(HERE) if f.endswith(".npz")))


if __name__ == "__main__":
    main()
