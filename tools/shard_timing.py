#!/usr/bin/env python3
"""Time one rank's share of the benchmark image for several shard counts on ONE GPU
(rows r, r+N, ... exactly as bench.py --gpus N deals them), to see how much of the
1-GPU time per image is fixed cost that does not shrink with the shard.  Optional argument: row-block sizes to
compare, comma separated (e.g. 1,2,8)."""
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from raytracer_amd import RenderIntegrator, abi, parallel, scenes

workdir = os.path.join(tempfile.gettempdir(), f"rt_bench_{os.getuid()}_0")
scene_path, config_path = scenes.materialize_c3(workdir, 100_000, 1200, 1200)
ri = RenderIntegrator.new_from_json(scene_path, config_path).prepare()
res = ri.resident(0)
dev = torch.device("cuda", 0)
blocks = [int(b) for b in sys.argv[1].split(",")] if len(sys.argv) > 1 else [parallel.ROW_BLOCK]
for n, block in [(n, b) for n in (1, 2, 4, 8) for b in (blocks if n > 1 else blocks[:1])]:
    cfg = parallel.shard_config(ri.config, 0, n, block)
    cfg.flags = abi.RT_FLAG_TIME_KERNELS
    rows, W = cfg.n_rows(), cfg.img_width
    out = {"pixels": torch.empty(rows, W, 3, dtype=torch.float64, device=dev),
           "sample_pixels": torch.empty(rows, W, 3, dtype=torch.float64, device=dev)}
    res.render(cfg, device_out=out)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    _, st = res.render(cfg, device_out=out)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    host = out["pixels"].cpu(); host2 = out["sample_pixels"].cpu()
    t2 = time.perf_counter()
    k = st.trace_kernel_ms + st.shade_kernel_ms + st.other_kernel_ms
    if n == 1:
        base = t1 - t0
    print(f"shards {n} (blocks of {block} rows, {rows} rows): render {1e3*(t1-t0):8.1f} ms (x{n} = {100*base/(n*(t1-t0)):5.1f} % of 1 shard)  extend {st.trace_kernel_ms:7.1f} shade {st.shade_kernel_ms:7.1f} "
          f"other {st.other_kernel_ms:6.1f}  rounds {st.batches}  traced/accepted {st.rays_traced/st.rays:.3f}  copy-out {1e3*(t2-t1):6.1f} ms", flush=True)
    print("   extend ms by depth x shards:", " ".join(f"{n * x:6.1f}" for x in list(st.trace_ms_by_depth)[:8]), flush=True)
