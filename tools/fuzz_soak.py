#!/usr/bin/env python3
"""Soak run of the random-scene parity check of tests/test_gpu_fuzz.py over many more seeds than the test suite holds:
closest hits of random rays and primary hits against the oracle, bit for bit, and a whole render every eighth seed.
tools/fuzz_soak.py FIRST LAST [IMAGE_EVERY=8] [--device-build]; with --device-build the tree is the one built on the
device (by default the reference's top-down build: the run then also counts the scenes whose device-built tree is not the
host's, node for node; RT_HIP_BUILD=0 / 1: the radix tree / the clustered one) and the oracle walks the downloaded copy of it
(rt_hip_scene_download_tree).  The RT_HIP_* environment (e.g. RT_HIP_WIDE=1 RT_HIP_TRI_PAD=1) picks the variant.
Pixels that differ in an image while every hit agrees are paths that a last-bit difference between the device's and libm's
sin / cos / pow sends to the other side of an edge; the run counts them."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from helpers import assert_image_parity, random_rays
from oracle import oracle
from test_gpu_fuzz import _scene

device_build = "--device-build" in sys.argv
argv = [a for a in sys.argv[1:] if not a.startswith("--")]
first, last = int(argv[0]), int(argv[1])
image_every = int(argv[2]) if len(argv) > 2 else 8
from helpers import image_parity
images = differing_pixels = images_with_differences = 0
t0 = time.time()
bad_seeds = []
other_tree = []  # scenes whose device-built tree differs from the host's
for seed in range(first, last):
    ri = _scene(seed)
    if device_build:
        import ctypes as C
        from raytracer_amd.integrator import ResidentScene
        from test_gpu_device_build import _scene_with_nodes
        res = ResidentScene(ri.scene, 0, build_on_device=True)
        nodes = res.download_tree()
        if os.environ.get("RT_HIP_BUILD", "-1") in ("-1", "2", "3"):
            from test_gpu_device_build import _preorder
            sc = ri.scene.contents
            if _preorder(nodes, len(nodes)) != _preorder(sc.nodes, sc.n_nodes):
                other_tree.append(seed)
        twin = _scene_with_nodes(ri, nodes)
        scene_for_oracle = C.byref(twin)
    else:
        res = ri.resident(0)
        scene_for_oracle = ri.scene
    o, d = random_rays(40_000, seed=1000 + seed)
    pg, tg = res.trace_rays(o, d)
    po, to = oracle.trace_rays(scene_for_oracle, o, d)
    cfg = ri.config.copy()
    pg2, tg2 = res.primary_hits(cfg)
    po2, to2 = oracle.primary_hits(scene_for_oracle, cfg)
    ok = np.array_equal(pg, po) and np.array_equal(tg, to) and np.array_equal(pg2, po2) and np.array_equal(tg2, to2)
    if ok and seed % image_every == 0:
        cfg.sample_batch_size, cfg.max_sample_batches, cfg.min_sample_batches = 8, 4, 0
        out_g, st_g = res.render(cfg)
        out_o, st_o = oracle.render(scene_for_oracle, cfg, 8)
        n_differ = image_parity(out_g, out_o)["n_differ"]
        images += 1
        differing_pixels += n_differ
        images_with_differences += n_differ > 0
        try:
            assert_image_parity(out_g, out_o, f"seed {seed}", res, cfg, noise_share=0.25)
        except AssertionError as e:
            ok = False
            print("image", seed, str(e)[:300], flush=True)
    if not ok:
        bad_seeds.append(seed)
        print("MISMATCH seed", seed, int(((pg != po) | (tg != to)).sum()), int(((pg2 != po2) | (tg2 != to2)).sum()), flush=True)
    if device_build:
        res.close()
    ri.close()
    if (seed - first) % 25 == 24:
        print(f"  {seed - first + 1} seeds, {time.time() - t0:.0f} s, mismatching so far: {bad_seeds}", flush=True)
print(f"seeds {first}..{last - 1}: {len(bad_seeds)} mismatching {bad_seeds}; {images} images of 1600 pixels rendered, {images_with_differences} with differing pixels, "
      f"{differing_pixels} such pixels in all; {time.time() - t0:.0f} s" + (f"; device-built tree differs from the host's on {len(other_tree)} scenes {other_tree[:20]}" if device_build else ""), flush=True)
sys.exit(1 if bad_seeds else 0)
