#!/usr/bin/env python3
"""One row of measurements per benchmark configuration (C1..C5 of BASELINE.json `configs`), on one GPU:

  * the full-size render with the scene resident (time-to-image, Mrays/s, mean spp, kernel times);
  * node visits / leaf tests per ray by the reference's visit rule (a two-batch render with
    RT_FLAG_REFERENCE_VISIT_RULE | RT_FLAG_COUNT_VISITS) -> algorithmic bytes per ray;
  * parity on a sample of rows against the CPU oracle with the same per-path random streams:
    primary-hit element ids and t, and the linear image (RMSE, share of pixels equal to 1e-9);
  * the CPU oracle's Mrays/s on those rows (all host threads given with --threads).

C2 is the reference's own teapot scene (tests/golden/ref_inputs).  The reference's dragon meshes are not distributed, so
C3 / C4 use the knot mesh generator at 100 000 / N triangles, and C5 the seeded soup written as OBJ files.

  tools/config_table.py [--configs C1,C2,C3,C4,C5] [--c4-triangles N] [--c5-triangles N] [--rows K] [--out FILE]
"""
import argparse
import json
import os
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from raytracer_amd import RenderIntegrator, abi, scenes
from oracle import oracle

NODE_BYTES, TRI_BYTES, RAY_STATE_BYTES = 64, 80, 60


def log(*a):
    print(*a, flush=True)


def build(name, args, workdir):
    if name == "C1":
        return RenderIntegrator.new(scenes.scene_c1(), scenes.config_c1(400, 400)), "Cornell room, metal box, 3 spheres, quad light; 400x400, 64 spp fixed"
    if name == "C2":
        ref = os.path.join(ROOT, "tests", "golden", "ref_inputs")
        with open(os.path.join(ref, "scene_teapot.json")) as f:
            scene = json.load(f)
        with open(os.path.join(ref, "config.json")) as f:
            cfg = json.load(f)
        cfg.update(img_width=800.0, img_height=800.0, sample_batch_size=16, max_sample_batches=16, min_sample_batches=16)
        return RenderIntegrator.new(scene, cfg, base_dir=ref), "the reference's input/scene_teapot.json (teapot.obj, 6 320 triangles); 800x800, 256 spp fixed"
    if name == "C3":
        sp, cp = scenes.materialize_c3(os.path.join(workdir, "c3"), 100_000, 1200, 1200)
        return RenderIntegrator.new_from_json(sp, cp), "100 000-triangle mesh in the README scene; 1200x1200, 20 x <=100 batches, z-test"
    if name == "C4":
        n = args.c4_triangles
        sp, cp = scenes.materialize_c3(os.path.join(workdir, "c4"), n, 2048, 2048, batch=16, max_batches=64, min_batches=2)
        return RenderIntegrator.new_from_json(sp, cp), f"{n}-triangle mesh in the README scene; 2048x2048, 16 x <=64 batches, z-test"
    if name == "C5":
        n = args.c5_triangles
        sp, cp = scenes.materialize_soup(os.path.join(workdir, "c5"), n, 3840, 2160, batch=64, max_batches=64, min_batches=2)
        return RenderIntegrator.new_from_json(sp, cp), f"{n}-triangle soup (10 % glass, 10 % metal), thin lens 0.6 deg; 3840x2160, 64 x <=64 batches, z-test"
    raise SystemExit(f"unknown config {name}")


def measure(name, args, workdir):
    t0 = time.time()
    ri, what = build(name, args, workdir)
    t_files = time.time() - t0
    t0 = time.time()
    ri.prepare()
    t_host = time.time() - t0
    t0 = time.time()
    res = ri.resident(0)
    t_upload = time.time() - t0
    sc = ri.scene.contents
    cfg = ri.config.copy()
    cfg.rng_mode = abi.RT_RNG_CTR
    H, W = cfg.img_height, cfg.img_width
    row = {"config": name, "what": what, "elements": int(sc.n_elements), "bvh_nodes": int(sc.n_nodes), "bvh_depth": int(ri.bvh_depth()),
           "input_files_s": t_files, "host_load_build_s": t_host, "upload_s": t_upload}
    log(f"[{name}] {what}: {sc.n_elements} elements, host {t_host:.2f} s, upload {t_upload:.2f} s")

    # ---- full render, scene resident; the second of two runs is reported (the first sizes the buffers)
    cfg.flags = abi.RT_FLAG_TIME_KERNELS
    best = None
    for _ in range(1 if name == "C5" else 2):
        t0 = time.perf_counter()
        out, st = res.render(cfg, want_linear=False, want_samples=False)
        wall = time.perf_counter() - t0
        best = (wall, st)
    wall, st = best
    row.update({"time_to_image_s": wall, "rays": int(st.rays), "mrays_per_s": st.rays / wall / 1e6, "mean_spp": st.samples / (W * H),
                "rays_per_sample": st.rays / st.samples, "extend_ms": st.trace_kernel_ms, "shade_ms": st.shade_kernel_ms,
                "other_ms": st.other_kernel_ms})
    log(f"[{name}] render {wall:.3f} s, {row['mrays_per_s']:.0f} Mrays/s, mean spp {row['mean_spp']:.1f}, extend {st.trace_kernel_ms:.0f} "
        f"shade {st.shade_kernel_ms:.0f} other {st.other_kernel_ms:.0f} ms")
    del out

    # ---- visits by the reference's rule on a two-batch render of the same image
    cnt = cfg.copy()
    cnt.max_sample_batches, cnt.min_sample_batches = min(2, cfg.max_sample_batches), min(2, cfg.min_sample_batches)
    cnt.flags = abi.RT_FLAG_REFERENCE_VISIT_RULE | abi.RT_FLAG_COUNT_VISITS
    _, sv = res.render(cnt, want_linear=False, want_samples=False)
    v_int, v_leaf = sv.node_visits / max(sv.rays_traced, 1), sv.leaf_tests / max(sv.rays_traced, 1)
    cnt.flags = abi.RT_FLAG_COUNT_VISITS
    _, sd = res.render(cnt, want_linear=False, want_samples=False)
    b_ray = v_int * NODE_BYTES + v_leaf * TRI_BYTES + RAY_STATE_BYTES
    achieved = st.rays * b_ray / (st.trace_kernel_ms * 1e-3) / 1e9
    row.update({"node_visits_per_ray": v_int, "leaf_tests_per_ray": v_leaf, "algorithmic_bytes_per_ray": b_ray, "algorithmic_gb_s": achieved,
                "working_set_bytes": int(sc.n_nodes // 2 * 32 + sc.n_triangles * 112), "node_visits_per_ray_device": sd.node_visits / max(sd.rays_traced, 1),
                "leaf_tests_per_ray_device": sd.leaf_tests / max(sd.rays_traced, 1)})
    log(f"[{name}] reference rule: {v_int:.1f} node visits, {v_leaf:.1f} leaf tests per ray -> {b_ray:.0f} algorithmic B/ray, {achieved:.0f} GB/s "
        f"(not an HBM figure while the working set, {row['working_set_bytes'] / 1e6:.0f} MB, is cache resident); device walk {row['node_visits_per_ray_device']:.1f} / {row['leaf_tests_per_ray_device']:.1f}")

    # ---- parity on a row sample: same per-path streams on both sides.  The sample is sized by the oracle's
    # speed on this scene (one probe row of primary rays): about --cpu-seconds per leg.
    probe = cfg.copy()
    probe.flags = 0
    probe.row_first, probe.row_stride = H // 2, H
    t0 = time.time()
    oracle.primary_hits(ri.scene, probe)
    rate1 = W * cfg.sample_batch_size / max(time.time() - t0, 1e-3)  # primary rays per second on one thread
    log(f"[{name}] oracle probe: {rate1 / 1e6:.3f} M primary rays/s on one thread")
    budget = min(args.cpu_rays, rate1 * args.threads * args.cpu_seconds)
    rays_per_row = st.rays / H
    k = int(max(args.threads, min(H, budget / rays_per_row)))
    k = -(-k // args.threads) * args.threads  # whole rounds of rows over the oracle's threads
    k = max(1, min(k, H))
    rows = cfg.copy()
    rows.flags = 0
    rows.row_stride = H // k
    rows.row_first = rows.row_stride // 2
    if rays_per_row * k > 1.5 * budget:  # even one round of rows is too much: compare the first batches only
        keep = max(1, int(cfg.max_sample_batches * budget / (rays_per_row * k)))
        rows.max_sample_batches = keep
        rows.min_sample_batches = min(cfg.min_sample_batches, keep)
        log(f"[{name}] parity sample limited to the first {keep} batches of {k} rows")
    prim = rows.copy()
    kp = int(max(1, min(k, args.cpu_seconds * rate1 / (W * cfg.sample_batch_size))))
    prim.row_stride = H // kp
    prim.row_first = prim.row_stride // 2
    pg, tg = res.primary_hits(prim)
    t0 = time.time()
    po, to = oracle.primary_hits(ri.scene, prim)
    t_prim = time.time() - t0
    same = pg == po
    hit = same & (po >= 0)
    rel = np.abs(tg[hit] - to[hit]) / np.abs(to[hit])
    row.update({"parity_rows": int(rows.n_rows()), "parity_batches": int(rows.max_sample_batches), "primary_rays_compared": int(po.size),
                "primary_id_equal": float(same.mean()), "primary_t_max_rel_err": float(rel.max()) if rel.size else 0.0,
                "primary_t_bit_equal": float((tg[same] == to[same]).mean())})
    log(f"[{name}] primary hits on {prim.n_rows()} rows ({po.size} rays, oracle {t_prim:.1f} s): ids equal {same.mean():.6f}, "
        f"max rel t err {row['primary_t_max_rel_err']:.2e}")
    og, sg = res.render(rows)
    oo, so = oracle.render(ri.scene, rows, args.threads)
    diff = og["linear"] - oo["linear"]
    close = np.isclose(og["linear"], oo["linear"], rtol=1e-9, atol=1e-9).all(axis=-1)
    row.update({"image_rmse_linear": float(np.sqrt((diff ** 2).mean())), "image_pixels_equal_1e9": float(close.mean()),
                "sample_counts_equal": float((og["samples"] == oo["samples"]).mean()),
                "image_rmse_gamma": float(np.sqrt(((og["pixels"] - oo["pixels"]) ** 2).mean())),
                "cpu_mrays_per_s": so.rays / so.wall_ms / 1e3, "cpu_threads": args.threads, "cpu_rays": int(so.rays), "cpu_seconds": so.wall_ms / 1e3,
                "cpu_time_to_image_s_extrapolated": row["rays"] / (so.rays / so.wall_ms * 1e3)})
    row["speedup_mrays"] = row["mrays_per_s"] / row["cpu_mrays_per_s"]
    log(f"[{name}] image on those rows: rmse {row['image_rmse_linear']:.3e} (linear), {close.mean():.4f} of pixels equal to 1e-9, "
        f"sample counts equal {row['sample_counts_equal']:.4f}; oracle {row['cpu_mrays_per_s']:.1f} Mrays/s on {args.threads} threads "
        f"({so.wall_ms / 1e3:.1f} s) -> {row['speedup_mrays']:.0f}x")
    return row


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--configs", default="C1,C2,C3,C4")
    ap.add_argument("--c4-triangles", type=int, default=2_000_000)
    ap.add_argument("--c5-triangles", type=int, default=10_000_000)
    ap.add_argument("--rows", type=int, default=4, help="fewest rows of each image compared with (and timed on) the CPU oracle")
    ap.add_argument("--cpu-rays", type=float, default=1.5e8, help="rays the oracle should trace per config where the image has them")
    ap.add_argument("--cpu-seconds", type=float, default=20.0, help="wall time aimed at for each oracle leg")
    ap.add_argument("--threads", type=int, default=min(16, os.cpu_count() or 1))
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "config_table.json"))
    args = ap.parse_args()
    workdir = os.path.join(tempfile.gettempdir(), f"rt_configs_{os.getuid()}")
    table = []
    if os.path.exists(args.out):
        with open(args.out) as f:
            table = json.load(f)
    for name in args.configs.split(","):
        r = measure(name, args, workdir)
        table = [x for x in table if x["config"] != name] + [r]
        os.makedirs(os.path.dirname(args.out), exist_ok=True)
        with open(args.out, "w") as f:
            json.dump(table, f, indent=1)


if __name__ == "__main__":
    main()
