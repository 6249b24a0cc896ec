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
    ap.add_argument("--split", default="SAH", help="the host build's bvh_split_method (SAH or Mid)")
    ap.add_argument("--methods", default="", help="comma-separated RT_HIP_BUILD values: the device leg once per method (-1 the scene's rule = the reference's top-down build, "
                                                 "0 radix tree, 1 clustering, 2 / 3 top-down bucketed / midpoint)")
    ap.add_argument("--only-device", action="store_true", help="skip the host-built tree's renders")
    ap.add_argument("--reupload", action="store_true", help="also: the device-built tree downloaded and uploaded again like a host-built one (tree shape apart from record layout)")
    args = ap.parse_args()
    from raytracer_amd import RenderIntegrator, abi, scenes
    from raytracer_amd.integrator import ResidentScene
    d = os.path.join(tempfile.gettempdir(), f"rt_build_timing_{os.getuid()}")
    if args.soup:
        scene_path, config_path = scenes.materialize_soup(d, args.triangles, args.size, args.size * 9 // 16, batch=16, max_batches=4, min_batches=2)
    else:
        scene_path, config_path = scenes.materialize_c3(d, args.triangles, args.size, args.size, batch=16, max_batches=args.max_batches, min_batches=2)
    if args.split != "SAH":
        with open(config_path) as f:
            cj = json.load(f)
        cj["bvh_split_method"] = args.split
        config_path = config_path.replace(".json", f"_{args.split}.json")
        with open(config_path, "w") as f:
            json.dump(cj, f)
    out = {"triangles": args.triangles, "image": [args.size, args.size], "host_split": args.split}
    # host path: load + expand + reference-order SAH build, upload, render
    t0 = time.perf_counter()
    ri = RenderIntegrator.new_from_json(scene_path, config_path).prepare()
    t1 = time.perf_counter()
    host = ResidentScene(ri.scene, 0)
    t2 = time.perf_counter()
    cfg = ri.config.copy()
    cfg.flags = abi.RT_FLAG_TIME_KERNELS | (0 if args.no_count else abi.RT_FLAG_COUNT_VISITS)
    if not args.only_device:
        host.render(cfg)  # first render pays for the working memory
    t3 = time.perf_counter()
    _, st = host.render(cfg) if not args.only_device else (None, abi.RtStats())
    t4 = time.perf_counter()
    out["host_tree"] = {"load_expand_build_s": t1 - t0, "upload_s": t2 - t1, "first_render_s": t3 - t2, "render_s": t4 - t3,
                        "mrays_per_s": st.rays / (t4 - t3) / 1e6, "extend_ms": st.trace_kernel_ms, "shade_ms": st.shade_kernel_ms,
                        "node_visits_per_ray": st.node_visits / max(st.rays_traced, 1), "leaf_tests_per_ray": st.leaf_tests / max(st.rays_traced, 1),
                        "stack_block_moves_per_ray": st.stack_block_moves / max(st.rays_traced, 1),
                        "bvh_depth": ri.bvh_depth()}
    host.close()
    # device path: the same primitives (the host's tree is ignored), tree built on the GPU
    def device_leg():
        t5 = time.perf_counter()
        dev = ResidentScene(ri.scene, 0, build_on_device=True)
        t6 = time.perf_counter()
        dev.render(cfg)
        t7 = time.perf_counter()
        _, sd = dev.render(cfg)
        t8 = time.perf_counter()
        leg = {"copy_and_build_s": t6 - t5, "first_render_s": t7 - t6, "render_s": t8 - t7, "mrays_per_s": sd.rays / (t8 - t7) / 1e6,
               "extend_ms": sd.trace_kernel_ms, "shade_ms": sd.shade_kernel_ms,
               "node_visits_per_ray": sd.node_visits / max(sd.rays_traced, 1), "leaf_tests_per_ray": sd.leaf_tests / max(sd.rays_traced, 1),
               "stack_block_moves_per_ray": sd.stack_block_moves / max(sd.rays_traced, 1)}
        return dev, sd, leg

    if args.methods:
        names = {"-1": "the scene's rule (the reference's top-down build)", "0": "radix tree over Morton codes", "1": "locally-ordered clustering",
                 "2": "the reference's top-down build, bucketed surface-area split", "3": "the reference's top-down build, midpoint split"}
        out["device_tree_by_method"] = {}
        for m in args.methods.split(","):
            os.environ["RT_HIP_BUILD"] = m
            ResidentScene(ri.scene, 0, build_on_device=True).close()  # (a first build of the process pays for its code objects)
            dev_m, _, leg = device_leg()
            leg["method"] = names.get(m, m)
            out["device_tree_by_method"][m] = leg
            dev_m.close()
        os.environ.pop("RT_HIP_BUILD", None)
    dev, sd, out["device_tree"] = device_leg()
    out["device_tree"]["bvh_depth"] = None
    if args.reupload:
        import ctypes as C
        import numpy as np
        raw = dev.download_tree()
        dt = np.dtype([("is_leaf", "<i4"), ("element", "<i4"), ("l", "<i4"), ("r", "<i4"), ("boxes", "<f8", (12,))])
        a = np.frombuffer(raw, dtype=dt).copy()
        # the upload wants children after their parents: relabel in preorder
        left, right, leaf = a["l"].tolist(), a["r"].tolist(), a["is_leaf"].tolist()
        order, stack = [], [0]
        while stack:
            i = stack.pop()
            order.append(i)
            if not leaf[i]:
                stack.append(right[i])
                stack.append(left[i])
        order = np.array(order, dtype=np.int64)
        rank = np.empty(len(a), dtype=np.int32)
        rank[order] = np.arange(len(a), dtype=np.int32)
        b = a[order]
        inner = b["is_leaf"] == 0
        b["l"][inner] = rank[b["l"][inner]]
        b["r"][inner] = rank[b["r"][inner]]
        nodes = (abi.RtNode * len(b)).from_buffer_copy(b.tobytes())
        twin = abi.RtScene()
        C.memmove(C.byref(twin), ri.scene, C.sizeof(abi.RtScene))
        twin.nodes = C.cast(nodes, C.POINTER(abi.RtNode))
        twin.n_nodes = len(nodes)
        t9 = time.perf_counter()
        re_up = ResidentScene(C.pointer(twin), 0)
        t10 = time.perf_counter()
        re_up.render(cfg)
        t11 = time.perf_counter()
        _, sr = re_up.render(cfg)
        t12 = time.perf_counter()
        out["device_tree_uploaded_by_the_host_path"] = {"upload_s": t10 - t9, "render_s": t12 - t11, "extend_ms": sr.trace_kernel_ms, "shade_ms": sr.shade_kernel_ms,
                                                        "node_visits_per_ray": sr.node_visits / max(sr.rays_traced, 1), "leaf_tests_per_ray": sr.leaf_tests / max(sr.rays_traced, 1)}
        re_up.close()
    def waves(x):
        return {"wave_node_steps_per_ray": x.wave_node_steps / max(x.rays_traced, 1), "wave_leaf_steps_per_ray": x.wave_leaf_steps / max(x.rays_traced, 1),
                "node_lane_share": x.node_visits / max(64 * x.wave_node_steps, 1), "leaf_lane_share": x.leaf_tests / max(64 * x.wave_leaf_steps, 1),
                "cycles_refill_node_leaf": [int(x.wave_cycles_refill), int(x.wave_cycles_node), int(x.wave_cycles_leaf)]}
    out["host_tree"].update(waves(st))
    out["device_tree"].update(waves(sd))
    out["note"] = "load_expand_build_s includes OBJ parsing and element expansion, which the device path needs too (it starts from the expanded primitives)"
    print(json.dumps(out))


if __name__ == "__main__":
    main()
