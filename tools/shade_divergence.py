#!/usr/bin/env python3
"""How much of k_shade's time is material divergence?  The benchmark scene as it is, and with the metal box and
the glass sphere made Lambertian (every wave then runs one material branch only): k_shade time per shaded path."""
import json
import os
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from raytracer_amd import RenderIntegrator, abi, scenes

d = os.path.join(tempfile.gettempdir(), f"rt_div_{os.getuid()}")
sp, cp = scenes.materialize_c3(d, 100_000, 1200, 1200, batch=20, max_batches=20, min_batches=2)
with open(sp) as f:
    sc = json.load(f)


def all_lambertian(x):
    if isinstance(x, dict):
        if "material" in x and not any("DiffuseLight" in k for k in x["material"]):
            x["material"] = scenes.lambertian(0.6, 0.6, 0.6)
        for v in x.values():
            all_lambertian(v)
    elif isinstance(x, list):
        for v in x:
            all_lambertian(v)


for name in ("as is", "all Lambertian"):
    if name != "as is":
        all_lambertian(sc)
    ri = RenderIntegrator.new(sc, json.load(open(cp)), base_dir=d).prepare()
    res = ri.resident(0)
    cfg = ri.config.copy()
    cfg.flags = abi.RT_FLAG_TIME_KERNELS
    res.render(cfg, want_linear=False, want_samples=False)
    _, st = res.render(cfg, want_linear=False, want_samples=False)
    print(f"{name}: rays traced {st.rays_traced/1e6:.0f} M, extend {st.trace_kernel_ms:.1f} ms, shade {st.shade_kernel_ms:.1f} ms "
          f"= {st.shade_kernel_ms*1e6/st.rays_traced:.4f} ns per shaded path", flush=True)
