#!/usr/bin/env python3
"""Soak run over random render configurations of the small Cornell scene against the oracle: image shapes that are not
whole tiles, batches of 1 to 700 samples, 1 to 9 batches with and without the z-test, row shards with blocks, a lens on a
third of them, depths 1 to 12.  tools/config_soak.py FIRST LAST"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from helpers import c1, image_parity
from oracle import oracle

first, last = int(sys.argv[1]), int(sys.argv[2])
t0 = time.time()
bad = []
pixels = differing = 0
for seed in range(first, last):
    rng = np.random.RandomState(seed)
    w, h = int(rng.randint(1, 70)), int(rng.randint(1, 50))
    batch = int(rng.choice([1, 2, 3, 5, 8, 16, 20, 33, 64, 100, 255, 256, 257, 300, 512, 513, 700]))
    if batch > 64:
        w, h = min(w, 12), min(h, 9)
    max_b = int(rng.randint(1, 10))
    min_b = int(rng.randint(0, max_b + 1))
    ri = c1(width=w, height=h, batch=batch, max_batches=max_b, min_batches=min_b, defocus=0.7 if seed % 3 == 0 else 0.0,
            split="SAH" if seed % 2 else "Mid", max_depth=int(rng.choice([1, 2, 5, 8, 12])))
    cfg = ri.config.copy()
    stride = int(rng.randint(1, 5))
    block = int(rng.choice([0, 1, 2, 3, 8]))
    cfg.row_stride, cfg.row_block = stride, block
    n_blocks = (h + max(block, 1) - 1) // max(block, 1)
    cfg.row_first = int(rng.randint(0, min(stride, n_blocks)))
    cfg.seed = int(rng.randint(0, 2**31))
    if cfg.n_rows() <= 0:
        continue
    res = ri.resident(0)
    out_g, st_g = res.render(cfg)
    out_o, st_o = oracle.render(ri.scene, cfg, 8)
    p = image_parity(out_g, out_o)
    pixels += p["n"]
    differing += p["n_differ"]
    ok = (out_g["linear"].shape == out_o["linear"].shape and np.array_equal(out_g["samples"], out_o["samples"]) and st_g.samples == st_o.samples
          and p["n_differ"] <= max(1, p["n"] // 500) and p["rmse_equal_pixels"] <= 1e-9)
    if not ok:
        bad.append(seed)
        print("MISMATCH seed", seed, (w, h, batch, max_b, min_b, stride, block, cfg.row_first), p, flush=True)
    ri.close()
print(f"configurations {first}..{last - 1}: {len(bad)} mismatching {bad}; {differing} of {pixels} pixels differ; {time.time() - t0:.0f} s", flush=True)
sys.exit(1 if bad else 0)
