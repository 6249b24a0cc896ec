#!/usr/bin/env python3
"""Time-to-image including scene preparation, host-built (reference-order SAH) versus device-built (LBVH) tree,
on a C4-shaped workload: generated mesh of --triangles triangles, --size^2 pixels, batches of 16, up to 64, z-test.

  python3 tools/device_build_timing.py [--triangles 2000000] [--size 2048]
"""
import argparse
import json
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--triangles", type=int, default=2_000_000)
    ap.add_argument("--size", type=int, default=2048)
    ap.add_argument("--max-batches", type=int, default=64)
    ap.add_argument("--soup", action="store_true", help="the triangle soup of BASELINE's largest configuration instead of the mesh (batches of 16, up to 4)")
    ap.add_argument("--no-count", action="store_true", help="time the production kernels (the visit-counting ones run at four waves per SIMD)")
    args = ap.parse_args()
    from raytracer_amd import RenderIntegrator, abi, scenes
    from raytracer_amd.integrator import ResidentScene
    d = os.path.join(tempfile.gettempdir(), f"rt_build_timing_{os.getuid()}")
    if args.soup:
        scene_path, config_path = scenes.materialize_soup(d, args.triangles, args.size, args.size * 9 // 16, batch=16, max_batches=4, min_batches=2)
    else:
        scene_path, config_path = scenes.materialize_c3(d, args.triangles, args.size, args.size, batch=16, max_batches=args.max_batches, min_batches=2)
    out = {"triangles": args.triangles, "image": [args.size, args.size]}
    # host path: load + expand + reference-order SAH build, upload, render
    t0 = time.perf_counter()
    ri = RenderIntegrator.new_from_json(scene_path, config_path).prepare()
    t1 = time.perf_counter()
    host = ResidentScene(ri.scene, 0)
    t2 = time.perf_counter()
    cfg = ri.config.copy()
    cfg.flags = abi.RT_FLAG_TIME_KERNELS | (0 if args.no_count else abi.RT_FLAG_COUNT_VISITS)
    host.render(cfg)  # first render pays for the working memory
    t3 = time.perf_counter()
    _, st = host.render(cfg)
    t4 = time.perf_counter()
    out["host_tree"] = {"load_expand_build_s": t1 - t0, "upload_s": t2 - t1, "first_render_s": t3 - t2, "render_s": t4 - t3,
                        "mrays_per_s": st.rays / (t4 - t3) / 1e6, "extend_ms": st.trace_kernel_ms, "shade_ms": st.shade_kernel_ms,
                        "node_visits_per_ray": st.node_visits / max(st.rays_traced, 1), "leaf_tests_per_ray": st.leaf_tests / max(st.rays_traced, 1),
                        "stack_block_moves_per_ray": st.stack_block_moves / max(st.rays_traced, 1),
                        "bvh_depth": ri.bvh_depth()}
    host.close()
    # device path: the same primitives (the host's tree is ignored), tree built on the GPU
    t5 = time.perf_counter()
    dev = ResidentScene(ri.scene, 0, build_on_device=True)
    t6 = time.perf_counter()
    dev.render(cfg)
    t7 = time.perf_counter()
    _, sd = dev.render(cfg)
    t8 = time.perf_counter()
    out["device_tree"] = {"copy_and_build_s": t6 - t5, "first_render_s": t7 - t6, "render_s": t8 - t7, "mrays_per_s": sd.rays / (t8 - t7) / 1e6,
                          "extend_ms": sd.trace_kernel_ms, "shade_ms": sd.shade_kernel_ms,
                          "node_visits_per_ray": sd.node_visits / max(sd.rays_traced, 1), "leaf_tests_per_ray": sd.leaf_tests / max(sd.rays_traced, 1),
                          "stack_block_moves_per_ray": sd.stack_block_moves / max(sd.rays_traced, 1), "bvh_depth": dev.bvh_depth() if hasattr(dev, "bvh_depth") else None}
    out["note"] = "load_expand_build_s includes OBJ parsing and element expansion, which the device path needs too (it starts from the expanded primitives)"
    print(json.dumps(out))


if __name__ == "__main__":
    main()
