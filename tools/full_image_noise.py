#!/usr/bin/env python3
"""The converged benchmark image against the CPU oracle in the REFERENCE's random-number mode (one xoshiro256++
per image row, integrator.rs:263 -- a different sample sequence, so only the statistics can agree).
The yardstick is the device against itself with another seed: two independent estimates at the same sample
counts differ by the Monte-Carlo noise and nothing else.  Prints and stores RMSE and mean differences.
  tools/full_image_noise.py [width height [max_batches]]"""
import json
import os
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from raytracer_amd import RenderIntegrator, abi, scenes
from oracle import oracle

w = int(sys.argv[1]) if len(sys.argv) > 1 else 1200
h = int(sys.argv[2]) if len(sys.argv) > 2 else 1200
mb = int(sys.argv[3]) if len(sys.argv) > 3 else 100
d = os.path.join(tempfile.gettempdir(), f"rt_noise_{os.getuid()}")
sp, cp = scenes.materialize_c3(d, 100_000, w, h, max_batches=mb)
ri = RenderIntegrator.new_from_json(sp, cp).prepare()
res = ri.resident(0)
cfg = ri.config.copy()
a, st_a = res.render(cfg)
cfg2 = cfg.copy()
cfg2.seed = 1
b, st_b = res.render(cfg2)
ref_cfg = cfg.copy()
ref_cfg.rng_mode = abi.RT_RNG_REF
print(f"device: {st_a.rays/1e9:.2f} G rays, mean spp {st_a.samples/(w*h):.1f}; oracle in reference RNG mode running ...", flush=True)
r, st_r = oracle.render(ri.scene, ref_cfg, min(16, os.cpu_count() or 1))


def rmse(x, y):
    return float(np.sqrt(((x - y) ** 2).mean()))


out = {
    "image": f"{w}x{h}, batches of {cfg.sample_batch_size}, <= {cfg.max_sample_batches}",
    "mean_spp": {"device": st_a.samples / (w * h), "device_other_seed": st_b.samples / (w * h), "oracle_reference_rng": st_r.samples / (w * h)},
    "rays": {"device": int(st_a.rays), "oracle_reference_rng": int(st_r.rays)},
    "oracle_seconds": st_r.wall_ms / 1e3,
    "rmse_linear": {"device_vs_oracle_reference_rng": rmse(a["linear"], r["linear"]), "device_vs_device_other_seed": rmse(a["linear"], b["linear"])},
    "rmse_gamma_pixels": {"device_vs_oracle_reference_rng": rmse(a["pixels"], r["pixels"]), "device_vs_device_other_seed": rmse(a["pixels"], b["pixels"])},
    "mean_linear_rgb": {"device": a["linear"].mean(axis=(0, 1)).tolist(), "device_other_seed": b["linear"].mean(axis=(0, 1)).tolist(),
                        "oracle_reference_rng": r["linear"].mean(axis=(0, 1)).tolist()},
    "pixels_within_1_of_255_gamma": {"device_vs_oracle_reference_rng": float((np.abs(a["pixels"] - r["pixels"]) <= 1 / 255).all(axis=-1).mean()),
                                     "device_vs_device_other_seed": float((np.abs(a["pixels"] - b["pixels"]) <= 1 / 255).all(axis=-1).mean())},
    "sample_counts_equal_share": float((a["samples"] == r["samples"]).mean()),
}
print(json.dumps(out, indent=1), flush=True)
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
with open(os.path.join(ROOT, "gpurun_out", "full_image_noise.json"), "w") as f:
    json.dump(out, f, indent=1)
