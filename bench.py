#!/usr/bin/env python3
"""Benchmark of the MI355X rendering path on the headline workload:
the README-like scene (Cornell box + metal box + glass sphere + quad light +
a generated 100 000-triangle mesh standing in for the undistributed dragon),
1200x1200, batches of 20 samples, at most 100 batches (2000 spp), z-test
adaptive sampling -- BASELINE.json configs[2].

One "step" = one full render of that image (scene already resident in HBM):
time-to-image.  value = BVH closest-hit queries per second (Mrays/s) summed
over all ranks.  With --gpus N (launched by torch.distributed.run, one
process per GPU) the rows of the SAME image are dealt round-robin over the
ranks and gathered to rank 0 with RCCL: total work is fixed ("strong").

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak
NODE_BYTES = 64        # device BVH node: two f32 child boxes + two refs
TRI_BYTES = 80         # device triangle record: v0, e1, e2 (f64) + ids
RAY_STATE_BYTES = 48 + 12  # k_extend reads origin + direction (6 f64), writes t + ref
# HBM/fabric bytes per ray of k_extend measured with rocprofv3 PMC passes (FETCH_SIZE x 2 + WRITE_SIZE)
# on this same workload; produced by tools/summarize_profiles.py from separate counter runs.
TRAFFIC_JSON = os.path.join(ROOT, "profiles", "r01_traffic.json")


def rust_crate_check():
    """bindings/rust/rt-hip-sys is written without a Rust toolchain: if this machine has one, type-check it."""
    import shutil
    import subprocess
    cargo = shutil.which("cargo")
    if not cargo:
        return {"cargo": None, "checked": False}
    try:
        r = subprocess.run([cargo, "check", "--offline", "--manifest-path", os.path.join(ROOT, "bindings", "rust", "rt-hip-sys", "Cargo.toml")],
                           capture_output=True, text=True, timeout=300)
        return {"cargo": cargo, "checked": True, "ok": r.returncode == 0, "stderr_tail": r.stderr[-400:]}
    except Exception as e:  # a broken toolchain must not cost the benchmark line
        return {"cargo": cargo, "checked": False, "error": str(e)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--width", type=int, default=1200)
    ap.add_argument("--height", type=int, default=1200)
    ap.add_argument("--triangles", type=int, default=100_000)
    ap.add_argument("--batch", type=int, default=20, help="sample_batch_size (other sizes are not the headline workload)")
    ap.add_argument("--max-batches", type=int, default=100)
    ap.add_argument("--min-batches", type=int, default=2)
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="target wall time of the CPU-baseline sample")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--early-terminate", action="store_true", help="set RT_FLAG_EARLY_TERMINATE (same pixels, fewer rays)")
    ap.add_argument("--reference-visits", action="store_true", help="set RT_FLAG_REFERENCE_VISIT_RULE (same pixels, more node visits)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from raytracer_amd import RenderIntegrator, abi, parallel, scenes

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("--gpus N > 1 must be launched with torch.distributed.run --nproc-per-node N")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the rendering path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))

    # ---- inputs (not timed): generated mesh + scene/config JSON, then the reference's host steps
    workdir = os.path.join(tempfile.gettempdir(), f"rt_bench_{os.getuid()}_{rank}")
    scene_path, config_path = scenes.materialize_c3(workdir, args.triangles, args.width, args.height, batch=args.batch,
                                                    max_batches=args.max_batches, min_batches=args.min_batches)
    t_host0 = time.time()
    ri = RenderIntegrator.new_from_json(scene_path, config_path).prepare()
    t_host = time.time() - t_host0
    t_up0 = time.time()
    res = ri.resident(local_rank)
    t_upload = time.time() - t_up0
    sc = ri.scene.contents
    cfg = parallel.shard_config(ri.config, rank, world)
    opt_flags = (abi.RT_FLAG_EARLY_TERMINATE if args.early_terminate else 0) | (abi.RT_FLAG_REFERENCE_VISIT_RULE if args.reference_visits else 0)
    cfg.flags = abi.RT_FLAG_TIME_KERNELS | opt_flags
    H, W = cfg.img_height, cfg.img_width
    n_rows = cfg.n_rows()
    dev = torch.device("cuda", local_rank)
    out = {
        "pixels": torch.empty(n_rows, W, 3, dtype=torch.float64, device=dev),
        "sample_pixels": torch.empty(n_rows, W, 3, dtype=torch.float64, device=dev),
        "samples": torch.empty(n_rows, W, dtype=torch.int32, device=dev),
    }

    # the finished image lands in page-locked host memory on rank 0
    if rank == 0:
        host_img = torch.empty(H, W, 3, dtype=torch.float64, pin_memory=True)
        host_heat = torch.empty(H, W, 3, dtype=torch.float64, pin_memory=True)

    def step():
        _, st = res.render(cfg, device_out=out)
        full = parallel.gather_rows(out["pixels"], H, rank, world)
        heat = parallel.gather_rows(out["sample_pixels"], H, rank, world)
        if rank == 0:
            host_img.copy_(full, non_blocking=True)  # final f64 image on the rank-0 host
            host_heat.copy_(heat, non_blocking=True)
        # every rank waits for its part of the exchange: the next render reuses the buffers it reads
        torch.cuda.current_stream().synchronize()
        if rank == 0:
            return st, host_img, host_heat
        return st, None, None

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    if world > 1:  # set up the gather's connections outside the timed region even with --warmup 0
        out["pixels"].zero_()
        parallel.gather_rows(out["pixels"], H, rank, world)
    for _ in range(args.warmup):
        step()
    sync()
    t0 = time.perf_counter()
    rays = trace_ms = launches = other_ms = shade_ms = 0.0
    ms_by_depth = [0.0] * 16
    samples = 0
    for _ in range(args.steps):
        st, img, heat = step()
        rays += st.rays
        trace_ms += st.trace_kernel_ms
        launches += st.trace_launches
        other_ms += st.other_kernel_ms
        shade_ms += st.shade_kernel_ms
        ms_by_depth = [a + b for a, b in zip(ms_by_depth, st.trace_ms_by_depth)]
        samples += st.samples
    sync()
    elapsed = time.perf_counter() - t0
    rays_by_depth = [int(x) for x in st.rays_by_depth]

    tot = torch.tensor([elapsed, rays, trace_ms, launches, float(samples)], dtype=torch.float64, device=dev)
    if world > 1:
        mx = tot.clone()
        dist.all_reduce(mx, op=dist.ReduceOp.MAX)
        dist.all_reduce(tot, op=dist.ReduceOp.SUM)
        elapsed = float(mx[0])
        trace_ms_max = float(mx[2])
    else:
        trace_ms_max = trace_ms
    rays_all, launches_all, samples_all = float(tot[1]), float(tot[3]), float(tot[4])

    # ---- untimed instrumented render: device-side visit counts for the byte model
    cfg_cnt = cfg.copy()
    cfg_cnt.flags = abi.RT_FLAG_COUNT_VISITS | opt_flags
    _, st_cnt = res.render(cfg_cnt, device_out=out)
    v_int_dev = st_cnt.node_visits / max(st_cnt.rays_traced, 1)
    v_leaf_dev = st_cnt.leaf_tests / max(st_cnt.rays_traced, 1)
    # fraction of the 64 lanes doing useful work when a wave executes a node step / a leaf test
    lanes_node = st_cnt.node_visits / max(64 * st_cnt.wave_node_steps, 1)
    lanes_leaf = st_cnt.leaf_tests / max(64 * st_cnt.wave_leaf_steps, 1)

    # ---- CPU baseline: the oracle (a port of the reference's Rayon path) on a bounded row sample
    cpu = None
    v_int, v_leaf = v_int_dev, v_leaf_dev
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from oracle import oracle
        threads = min(16, os.cpu_count() or 1)
        ref_cfg = ri.config.copy()
        ref_cfg.rng_mode = abi.RT_RNG_REF
        probe = ref_cfg.copy()
        probe.row_first, probe.row_stride = H // 2, H  # one row through the middle of the image
        t_p = time.perf_counter()
        _, st_probe = oracle.render(ri.scene, probe, 1)
        # one row on one thread; `threads` rows run concurrently in the sample
        per_row = max(time.perf_counter() - t_p, 1e-3)
        rows_wanted = max(threads, int(args.cpu_seconds / per_row * threads))
        stride = max(1, H // min(H, rows_wanted))
        sample = ref_cfg.copy()
        sample.row_first, sample.row_stride = stride // 2, stride
        _, st_cpu = oracle.render(ri.scene, sample, threads)
        cpu_rows = sample.n_rows()
        v_int = st_cpu.node_visits / max(st_cpu.rays, 1)
        v_leaf = st_cpu.leaf_tests / max(st_cpu.rays, 1)
        cpu = {
            "value": st_cpu.rays / st_cpu.wall_ms / 1e3,
            "unit": "Mrays/s",
            "cores": threads,
            "kind": "port",
            "single_core_value": st_probe.rays / max(st_probe.wall_ms, 1e-6) / 1e3,
            "sample": f"rows {sample.row_first}::{stride} of the same {W}x{H} workload ({cpu_rows} rows, "
                      f"{int(st_cpu.rays)} rays, {st_cpu.wall_ms / 1e3:.1f} s, reference RNG mode)",
            "rays": int(st_cpu.rays),
            "seconds": st_cpu.wall_ms / 1e3,
            "time_to_image_s_extrapolated": st_cpu.wall_ms / 1e3 * H / cpu_rows,
            "node_visits_per_ray": v_int,
            "leaf_tests_per_ray": v_leaf,
        }

    if rank == 0:
        bytes_per_ray = v_int * NODE_BYTES + v_leaf * TRI_BYTES + RAY_STATE_BYTES
        # per-GPU figures for the dominant kernel (k_extend): this rank's rays over this rank's kernel time
        achieved = (rays * bytes_per_ray) / (trace_ms * 1e-3) / 1e9 if trace_ms > 0 else 0.0
        value = rays_all / elapsed / 1e6
        traffic = traffic_note = None
        headline = (args.triangles, W, H, args.batch, args.max_batches, args.min_batches) == (100_000, 1200, 1200, 20, 100, 2)
        if os.path.exists(TRAFFIC_JSON) and headline and not opt_flags:
            with open(TRAFFIC_JSON) as fh:
                tj = json.load(fh)
            if tj.get("hbm_bytes_per_ray"):
                traffic = tj["hbm_bytes_per_ray"] * rays / max(launches, 1)  # bytes per launch
                traffic_note = (f"{tj['hbm_bytes_per_ray']:.1f} B/ray from PMC passes (profiles/r01_traffic.json) "
                                f"x {rays / max(launches, 1):.0f} rays per launch")
        line = {
            "metric": "Mrays/s",
            "value": value,
            "unit": "Mrays/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "time_to_image_s": elapsed / args.steps,
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {
                "workload": f"README-like scene, generated {args.triangles}-triangle mesh + metal box + glass sphere + quad light "
                            f"(attractor), {W}x{H}, batch {cfg.sample_batch_size} x <= {cfg.max_sample_batches} batches "
                            f"(min {cfg.min_sample_batches}), depth {cfg.max_depth}, SAH BVH, adaptive z-test",
                "elements": sc.n_elements, "bvh_nodes": sc.n_nodes, "bvh_depth": ri.bvh_depth(),
                "rows_per_rank": n_rows, "parallelism": f"rows interleaved over {world} GPU(s), gather to rank 0",
                "early_terminate": bool(args.early_terminate), "reference_visit_rule": bool(args.reference_visits),
            },
            "rays_per_step": rays_all / args.steps,
            "mean_spp": samples_all / args.steps / (W * H),
            "rays_by_depth_rank0_last_step": rays_by_depth[:cfg.max_depth],
            "host_build_s": t_host,
            "upload_s": t_upload,
            "time_to_image_with_load_build_upload_s": elapsed / args.steps + t_host + t_upload,
            "roofline": {
                "bound": "hbm",
                "kernel": "k_extend",
                "achieved": achieved,
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS,
                "traffic": traffic,
                "traffic_unit": "bytes per launch",
                "traffic_source": traffic_note,
                "algorithmic_bytes_per_launch": rays * bytes_per_ray / max(launches, 1),
                "bytes_per_ray": bytes_per_ray,
                "node_visits_per_ray": v_int,
                "leaf_tests_per_ray": v_leaf,
                "node_visits_per_ray_device": v_int_dev,
                "leaf_tests_per_ray_device": v_leaf_dev,
                "simd_lane_utilisation": {"node_steps": lanes_node, "leaf_tests": lanes_leaf},
                "launches": launches / args.steps,
                "avg_launch_ms": trace_ms / max(launches, 1),
                "kernel_ms_per_step": trace_ms / args.steps,
                "shade_kernel_ms_per_step": shade_ms / args.steps,
                "other_kernels_ms_per_step": other_ms / args.steps,
                "kernel_ms_by_depth_per_step": [x / args.steps for x in ms_by_depth[:cfg.max_depth]],
                "max_rank_kernel_ms_per_step": trace_ms_max / args.steps,
            },
            "cpu_baseline": cpu,
            "rust_sys_crate": rust_crate_check(),
        }
        if cpu:
            line["speedup_vs_cpu_mrays"] = value / cpu["value"]
            line["speedup_vs_cpu_time_to_image"] = cpu["time_to_image_s_extrapolated"] / (elapsed / args.steps)
        print(json.dumps(line))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
