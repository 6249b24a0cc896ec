#!/usr/bin/env python3
"""Bounce 0 by the packet walk against bounce 0 by the general walk on the larger configurations:
  tools/packet_ab.py mesh 2000000 2048   |   tools/packet_ab.py soup 10000000 960"""
import os
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from raytracer_amd import RenderIntegrator, abi, scenes  # noqa: E402

kind, n, size = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
d = os.path.join(tempfile.gettempdir(), f"rt_packet_ab_{os.getuid()}")
if kind == "soup":
    sp, cp = scenes.materialize_soup(d, n, size, size * 9 // 16, batch=16, max_batches=4, min_batches=2)
else:
    sp, cp = scenes.materialize_c3(d, n, size, size, batch=16, max_batches=64, min_batches=2)
ri = RenderIntegrator.new_from_json(sp, cp).prepare()
res = ri.resident(0)
cfg = ri.config.copy()
cfg.flags = abi.RT_FLAG_TIME_KERNELS
res.render(cfg, want_linear=False, want_samples=False)
for packet in (1, 0, 1, 0):
    res.tune(packet=packet)
    _, st = res.render(cfg, want_linear=False, want_samples=False)
    print(f"{kind} {n} packet={packet}: extend {st.trace_kernel_ms:8.1f} shade {st.shade_kernel_ms:7.1f} ms  extend by bounce "
          + " ".join(f"{x:7.1f}" for x in list(st.trace_ms_by_depth)[:8]), flush=True)
