#!/usr/bin/env python3
"""Time EVERY rank's share of the benchmark image for 2, 4 and 8 shards on ONE GPU (rows dealt exactly as
bench.py --gpus N deals them).  An N-GPU render takes as long as its slowest rank (plus the gather), so the figure that
bounds the scaling efficiency is  t(1 shard) / (N * max over ranks of t(rank's shard)).

  tools/shard_timing.py [row blocks, comma separated] [--json out.json] [--reuse]
"""
import json
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from raytracer_amd import RenderIntegrator, abi, parallel, scenes  # noqa: E402

argv = sys.argv[1:]
out_json = None
if "--json" in argv:
    i = argv.index("--json")
    out_json = argv[i + 1]
    del argv[i:i + 2]
reuse = "--reuse" in argv
argv = [a for a in argv if a != "--reuse"]
blocks = [int(b) for b in argv[0].split(",")] if argv else [parallel.ROW_BLOCK]

workdir = os.path.join(tempfile.gettempdir(), f"rt_bench_{os.getuid()}_0")
scene_path, config_path = scenes.materialize_c3(workdir, 100_000, 1200, 1200)
ri = RenderIntegrator.new_from_json(scene_path, config_path).prepare()
res = ri.resident(0)
dev = torch.device("cuda", 0)
flags = abi.RT_FLAG_TIME_KERNELS | (abi.RT_FLAG_REUSE_PRIMARY_HITS if reuse else 0)


def render_ms(cfg, repeats=2):
    rows, W = cfg.n_rows(), cfg.img_width
    out = {"pixels": torch.empty(rows, W, 3, dtype=torch.float64, device=dev),
           "sample_pixels": torch.empty(rows, W, 3, dtype=torch.float64, device=dev)}
    res.render(cfg, device_out=out)
    torch.cuda.synchronize()
    best, st = None, None
    for _ in range(repeats):
        t0 = time.perf_counter()
        _, st = res.render(cfg, device_out=out)
        torch.cuda.synchronize()
        ms = 1e3 * (time.perf_counter() - t0)
        best = ms if best is None else min(best, ms)
    return best, st


report = {"workload": "bench.py headline image (1200 x 1200, batches of 20, <= 100 batches), one MI355X", "reuse_primary_hits": reuse, "shards": []}
cfg = parallel.shard_config(ri.config, 0, 1, blocks[0])
cfg.flags = flags
whole, st = render_ms(cfg)
print(f"whole image: {whole:8.1f} ms  rays {st.rays / 1e9:.3f} G  extend {st.trace_kernel_ms:.1f} shade {st.shade_kernel_ms:.1f} other {st.other_kernel_ms:.1f} "
      f"rounds {st.batches} traced/accepted {st.rays_traced / st.rays:.4f}", flush=True)
report["whole_image_ms"] = whole
report["whole_image_kernels"] = {"extend_ms": st.trace_kernel_ms, "shade_ms": st.shade_kernel_ms, "other_ms": st.other_kernel_ms, "rounds": int(st.batches),
                                 "traced_over_accepted": st.rays_traced / st.rays}
for n in (2, 4, 8):
    for block in blocks:
        times, rays, parts = [], [], []
        for r in range(n):
            cfg = parallel.shard_config(ri.config, r, n, block)
            cfg.flags = flags
            ms, st = render_ms(cfg)
            times.append(ms)
            rays.append(int(st.rays))
            parts.append({"extend_ms": round(st.trace_kernel_ms, 2), "shade_ms": round(st.shade_kernel_ms, 2), "other_ms": round(st.other_kernel_ms, 2),
                          "rounds": int(st.batches), "traced_over_accepted": round(st.rays_traced / st.rays, 4)})
        mx, mean = max(times), sum(times) / n
        row = {"shards": n, "row_block": block, "ms_per_rank": [round(t, 2) for t in times], "rays_per_rank": rays, "max_ms": mx, "mean_ms": mean,
               "max_over_mean": mx / mean, "kernels_per_rank": parts, "rays_max_over_mean": max(rays) / (sum(rays) / n),
               "efficiency_of_the_slowest_rank": whole / (n * mx), "efficiency_of_the_mean_rank": whole / (n * mean)}
        report["shards"].append(row)
        print(f"shards {n} (blocks of {block}): max {mx:7.1f} ms  mean {mean:7.1f} ms  max/mean {mx / mean:.3f}  rays max/mean {row['rays_max_over_mean']:.3f}  "
              f"slowest-rank efficiency {100 * whole / (n * mx):5.1f} %  mean-rank {100 * whole / (n * mean):5.1f} %  rank 0: {parts[0]}", flush=True)
if out_json:
    with open(out_json, "w") as f:
        json.dump(report, f, indent=1)
