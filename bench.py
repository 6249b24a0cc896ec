#!/usr/bin/env python3
"""Benchmark of the MI355X rendering path on the headline workload:
the README-like scene (Cornell box + metal box + glass sphere + quad light +
a generated 100 000-triangle mesh standing in for the undistributed dragon),
1200x1200, batches of 20 samples, at most 100 batches (2000 spp), z-test
adaptive sampling -- BASELINE.json configs[2].

One "step" = one full render of that image (scene already resident in HBM):
time-to-image.  value = BVH closest-hit queries per second (Mrays/s) summed
over all ranks.  With --gpus N the rows of the SAME image are dealt over N
ranks (one process per GPU) and gathered to rank 0 with one RCCL collective:
total work is fixed ("strong").  `python bench.py --gpus N` typed as such starts
the N ranks itself (torch.distributed.run as a child process, before anything
touches the GPU); under torch.distributed.run it is one of the ranks.

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec peak
L2_PEAK_GBS = 34500.0   # aggregate L2 bandwidth of the 8 XCDs (measured figure of the MI355X guide)
N_SIMD = 1024           # 256 CUs x 4 SIMDs
NODE_BYTES = 64         # the byte model's node: two f32 child boxes + two refs (the device reads a 32-byte grid form)
TRI_BYTES = 80          # device triangle record: v0, e1, e2 (f64) + ids
RAY_STATE_BYTES = 48 + 12  # k_extend reads origin + direction (6 f64), writes t + ref
# Counter sums per kernel of ONE render of this workload (separate rocprofv3 --pmc passes of this file, condensed
# by tools/summarize_profiles.py): what the roofline figures that no event can measure are derived from.
ROOFLINE_INPUTS = os.path.join(ROOT, "profiles", "roofline_inputs.json")


def parse_args(argv=None):
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
    ap.add_argument("--row-block", type=int, default=None, help="rows dealt together to a rank (default: raytracer_amd.parallel.ROW_BLOCK)")
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="target wall time of the CPU-baseline sample")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--early-terminate", action="store_true", help="set RT_FLAG_EARLY_TERMINATE (same pixels, fewer rays)")
    ap.add_argument("--reference-visits", action="store_true", help="set RT_FLAG_REFERENCE_VISIT_RULE (same pixels, more node visits)")
    ap.add_argument("--profile-pass", action="store_true",
                    help="exactly one render and nothing else on the GPU (for rocprofv3 counter passes): implies --steps 1 --warmup 0, "
                         "no CPU baseline, no instrumented render")
    ap.add_argument("--no-reuse-leg", action="store_true", help="skip the extra renders with RT_FLAG_REUSE_PRIMARY_HITS")
    ap.add_argument("--c-gather", action="store_true",
                    help="gather inside librt_hip.so (rt_hip_render_ranks: RCCL sends and receives issued by the C library) instead of "
                         "torch.distributed's gather; the render and the sharding are the same")
    ap.add_argument("--dry-run", action="store_true", help="print the launch plan (child command and environment) and exit")
    args = ap.parse_args(argv)
    if args.profile_pass:
        args.steps, args.warmup, args.no_cpu_baseline = 1, 0, True
    return args


def spawn_plan(args, argv):
    """The child command that starts `args.gpus` ranks of this file, one per GPU, and the environment it adds."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + [a for a in argv if a != "--dry-run"]
    env = {"MASTER_ADDR": "127.0.0.1", "HSA_ENABLE_IPC_MODE_LEGACY": os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0")}
    return cmd, env


def launch_ranks(args, argv):
    """`python bench.py --gpus N` outside a launcher: run the ranks as children of a process that has not touched the
    GPU, relay rank 0's JSON line and the children's exit code."""
    cmd, env = spawn_plan(args, argv)
    if args.dry_run:
        print(json.dumps({"spawn": cmd, "env": env, "ranks": args.gpus}))
        return 0
    p = subprocess.run(cmd, env=dict(os.environ, **env), stdout=subprocess.PIPE, text=True)
    lines = [l for l in p.stdout.splitlines() if l.startswith("{") and '"metric"' in l]
    for l in p.stdout.splitlines():
        if l not in lines:
            print(l, file=sys.stderr)
    if lines:
        print(lines[-1])
    return p.returncode if p.returncode else (0 if lines else 1)


def host_cpu():
    """Cores this process may use (affinity mask, and OMP_NUM_THREADS where the box sets its share that way) and the CPU model."""
    try:
        visible = len(os.sched_getaffinity(0))
    except AttributeError:
        visible = os.cpu_count() or 1
    share = visible
    try:
        share = max(1, min(visible, int(os.environ.get("OMP_NUM_THREADS", visible))))
    except ValueError:
        pass
    model = None
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.lower().startswith("model name"):
                    model = line.split(":", 1)[1].strip()
                    break
    except OSError:
        pass
    return {"threads_used": share, "affinity_cpus": visible, "os_cpu_count": os.cpu_count(), "model": model}


def rust_crate_check():
    """bindings/rust/rt-hip-sys is written without a Rust toolchain: if this machine has one, type-check it."""
    import shutil
    cargo = shutil.which("cargo")
    if not cargo:
        return {"cargo": None, "checked": False}
    try:
        r = subprocess.run([cargo, "check", "--offline", "--manifest-path", os.path.join(ROOT, "bindings", "rust", "rt-hip-sys", "Cargo.toml")],
                           capture_output=True, text=True, timeout=300)
        return {"cargo": cargo, "checked": True, "ok": r.returncode == 0, "stderr_tail": r.stderr[-400:]}
    except Exception as e:  # a broken toolchain must not cost the benchmark line
        return {"cargo": cargo, "checked": False, "error": str(e)}


def load_roofline_inputs(path, source_hash):
    """The committed counter sums, if they were taken on THIS build of the kernels: (inputs, None) or (None, why not)."""
    if not os.path.exists(path):
        return None, f"{os.path.relpath(path, ROOT)} is missing"
    with open(path) as fh:
        inputs = json.load(fh)
    have = inputs.get("source_hash")
    if have != source_hash:
        return None, (f"{os.path.relpath(path, ROOT)} was measured on another build of the kernels (source hash {have}, this tree "
                      f"{source_hash}): run tools/profile_round.sh and tools/summarize_profiles.py again")
    return inputs, None


# the event-timed kernel time of this run may differ from the profiled one by this factor before the counters are refused
PROFILE_TIME_TOLERANCE = 0.15


def counter_roofline(inputs, rays, extend_ms, shade_ms, lanes_useful, extend_launches=None):
    """Roofline figures of the two dominant kernels from the committed counter sums (per ray of the profiled render)
    and this run's event times.  Formulas:
      VALU busy  = 4 * SQ_ACTIVE_INST_VALU / (N_SIMD * GRBM_GUI_ACTIVE / 8)        [SQ counts quad-cycles; GRBM sums 8 XCDs]
      k_extend   bound "valu-issue": frac = VALU busy * share of the lanes enabled in the issued VALU instructions
                 (SQ_THREAD_CYCLES_VALU / (64 * SQ_ACTIVE_INST_VALU); without that counter, the instrumented per-step share)
      fabric B/s = (FETCH_SIZE * fetch_factor + WRITE_SIZE) KiB * 1024 / kernel time  -> hbm_frac against 8 TB/s
      L2 B/s     = (TCC_HIT + TCC_MISS) * 128 B / kernel time                        -> l2_frac against 34.5 TB/s
    """
    out = {}
    prof_rays = inputs["rays"]
    ff = inputs.get("fetch_factor", 2.0)
    for name, ms in (("k_extend", extend_ms), ("k_shade", shade_ms)):
        c = inputs["kernels"].get(name)
        if not c or ms <= 0:
            continue
        scale = rays / prof_rays  # counters scale with the rays traced
        t = ms * 1e-3
        fabric = (c["FETCH_SIZE"] * ff + c["WRITE_SIZE"]) * 1024.0 * scale
        l2 = (c["TCC_HIT_sum"] + c["TCC_MISS_sum"]) * 128.0 * scale
        valu_busy = 4.0 * c["SQ_ACTIVE_INST_VALU"] / (N_SIMD * c["GRBM_GUI_ACTIVE"] / 8.0)
        time_ratio = None
        if name == "k_extend" and extend_launches and c.get("kernel_avg_launch_ms"):
            # The busy share above is that of the profiled run.  The same instructions issued over THIS run's kernel time
            # (another box of the pool can be 5 % slower) are busy in proportion; a larger gap means something else than
            # the box changed, and the counters are refused.
            time_ratio = c["kernel_avg_launch_ms"] / (ms / extend_launches)  # same workload, same launches per render
            if abs(time_ratio - 1.0) > PROFILE_TIME_TOLERANCE:
                out["refused"] = (f"k_extend took {ms / extend_launches:.3f} ms per launch here but {c['kernel_avg_launch_ms']:.3f} ms when "
                                  f"the counters were taken (more than {PROFILE_TIME_TOLERANCE:.0%} apart)")
                continue
            valu_busy *= time_ratio
        out[name] = {
            "profiled_over_this_run_kernel_time": time_ratio,
            "fabric_bytes": fabric, "fabric_GBs": fabric / t / 1e9, "hbm_frac": fabric / t / 1e9 / HBM_PEAK_GBS,
            "l2_GBs": l2 / t / 1e9, "l2_frac": l2 / t / 1e9 / L2_PEAK_GBS,
            "l2_hit_rate": c["TCC_HIT_sum"] / max(c["TCC_HIT_sum"] + c["TCC_MISS_sum"], 1.0),
            "valu_busy": valu_busy,
            "wait_share_of_wave_time": c["SQ_WAIT_ANY"] / max(c["SQ_WAVE_CYCLES"], 1.0),
            "profiled_avg_launch_ms": c.get("kernel_avg_launch_ms"), "this_run_kernel_ms": ms,
        }
        if c.get("SQ_THREAD_CYCLES_VALU"):
            # the hardware's own count of the lanes enabled in the VALU instructions it issued (rocprof's VALUUtilization);
            # lanes without work are masked off by the kernels' branches, so enabled = useful
            out[name]["valu_lane_share"] = c["SQ_THREAD_CYCLES_VALU"] / (64.0 * c["SQ_ACTIVE_INST_VALU"])
    if "k_extend" in out:
        # instrumented render: lanes with a node visit or a primitive test per wave step (every step weighs the same)
        out["k_extend"]["useful_lane_share_by_steps"] = lanes_useful
        share = out["k_extend"].get("valu_lane_share", lanes_useful)
        out["k_extend"]["useful_lane_share"] = share
        out["k_extend"]["valu_issue_frac"] = out["k_extend"]["valu_busy"] * share
    return out


def main():
    argv = sys.argv[1:]
    args = parse_args(argv)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args, argv))
    if args.dry_run:
        print(json.dumps({"spawn": None, "ranks": int(os.environ.get("WORLD_SIZE", "1"))}))
        return

    import torch
    import torch.distributed as dist
    from raytracer_amd import RenderIntegrator, abi, parallel, scenes

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but the launcher started {world} rank(s)")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the rendering path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
    row_block = parallel.ROW_BLOCK if args.row_block is None else max(1, args.row_block)

    # ---- inputs (not timed): generated mesh + scene/config JSON, then the reference's host steps
    workdir = os.path.join(tempfile.gettempdir(), f"rt_bench_{os.getuid()}_{rank}")
    scene_path, config_path = scenes.materialize_c3(workdir, args.triangles, args.width, args.height, batch=args.batch,
                                                    max_batches=args.max_batches, min_batches=args.min_batches)
    t_host0 = time.time()
    ri = RenderIntegrator.new_from_json(scene_path, config_path).prepare()
    t_host = time.time() - t_host0
    # The first upload of a process also starts the HIP runtime and loads the library's code object; a second one is what
    # a scene costs a process that is already running (both are reported).
    from raytracer_amd.integrator import ResidentScene
    t_up0 = time.time()
    ResidentScene(ri.scene, local_rank).close()
    t_upload_first = time.time() - t_up0
    t_up0 = time.time()
    res = ri.resident(local_rank)
    t_upload = time.time() - t_up0
    # beside it (opt-in, not what the timed steps use): the reference's tree built on the device from the primitives alone
    t_device_build = None
    if rank == 0:
        try:
            t_b0 = time.time()
            ResidentScene(ri.scene, local_rank, build_on_device=True).close()
            t_device_build = time.time() - t_b0
        except abi.RtError as e:  # a side figure must not take the benchmark line with it
            print(f"device build failed: {e}", file=sys.stderr)
    sc = ri.scene.contents
    cfg = parallel.shard_config(ri.config, rank, world, row_block)
    opt_flags = (abi.RT_FLAG_EARLY_TERMINATE if args.early_terminate else 0) | (abi.RT_FLAG_REFERENCE_VISIT_RULE if args.reference_visits else 0)
    cfg.flags = abi.RT_FLAG_TIME_KERNELS | opt_flags
    H, W = cfg.img_height, cfg.img_width
    n_rows = cfg.n_rows()
    dev = torch.device("cuda", local_rank)
    # image and heat-map are two planes of one buffer, so that one collective moves both
    planes = torch.empty(2, n_rows, W, 3, dtype=torch.float64, device=dev)
    out = {"pixels": planes[0], "sample_pixels": planes[1], "samples": torch.empty(n_rows, W, dtype=torch.int32, device=dev)}

    # the finished image lands in page-locked host memory on rank 0
    if rank == 0:
        host_planes = torch.empty(2, H, W, 3, dtype=torch.float64, pin_memory=True)

    comm = None
    if args.c_gather:  # the communicator of the C library: rank 0's id reaches the others through torch's store
        idt = torch.zeros(abi.RT_COMM_ID_BYTES, dtype=torch.uint8, device=dev)
        if rank == 0:
            idt.copy_(torch.frombuffer(bytearray(parallel.RankComm.new_id()), dtype=torch.uint8))
        if world > 1:
            dist.broadcast(idt, 0)
        comm = parallel.RankComm(idt.cpu().numpy().tobytes(), rank, world, local_rank)
        whole = torch.empty(2, H, W, 3, dtype=torch.float64, device=dev) if rank == 0 else None

    def step():
        if comm is not None:
            _, st = comm.render(res, cfg, device_out={"pixels": whole[0], "sample_pixels": whole[1]} if rank == 0 else None)
            if rank == 0:
                host_planes.copy_(whole, non_blocking=True)
            torch.cuda.current_stream().synchronize()
            return st
        _, st = res.render(cfg, device_out=out)  # (`cfg` is looked up at call time: the reuse leg below swaps it)
        full = parallel.gather_rows(planes, H, rank, world, row_block=row_block, row_dim=1)
        if rank == 0:
            host_planes.copy_(full, non_blocking=True)  # final f64 image + heat-map on the rank-0 host
        # every rank waits for its part of the exchange: the next render reuses the buffers it reads
        torch.cuda.current_stream().synchronize()
        return st

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    if world > 1:  # set up the gather's connections outside the timed region even with --warmup 0
        planes.zero_()
        parallel.gather_rows(planes, H, rank, world, row_block=row_block, row_dim=1)
    for _ in range(args.warmup):
        step()
    sync()
    t0 = time.perf_counter()
    rays = trace_ms = launches = other_ms = shade_ms = 0.0
    ms_by_depth = [0.0] * 16
    samples = 0
    for _ in range(args.steps):
        st = step()
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
    rays_traced = float(st.rays_traced) * args.steps

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
    if comm is not None:  # rt_hip_render_ranks already reports the job's query and sample counts on every rank
        rays_all, samples_all = float(rays), float(samples)

    # ---- beside the headline: the same image with RT_FLAG_REUSE_PRIMARY_HITS (every primary ray traced once per round
    # instead of once per batch: identical pixels, fewer queries executed than the reference makes).  Timed like the
    # headline steps but never part of `value`.
    reuse = None
    if not args.profile_pass and not args.no_reuse_leg:
        cfg_reuse = cfg.copy()
        cfg_reuse.flags = cfg.flags | abi.RT_FLAG_REUSE_PRIMARY_HITS
        saved, cfg = cfg, cfg_reuse
        n_reuse = max(1, min(args.steps, 3))
        step()
        sync()
        t_r0 = time.perf_counter()
        for _ in range(n_reuse):
            st_r = step()
        sync()
        t_reuse = (time.perf_counter() - t_r0) / n_reuse
        cfg = saved
        tr = torch.tensor([t_reuse, float(st_r.rays), float(st_r.rays_traced)], dtype=torch.float64, device=dev)
        if world > 1:
            mxr = tr.clone()
            dist.all_reduce(mxr, op=dist.ReduceOp.MAX)
            dist.all_reduce(tr, op=dist.ReduceOp.SUM)
            t_reuse = float(mxr[0])
            if comm is not None:
                tr[1], tr[2] = float(st_r.rays), float(st_r.rays_traced)
        reuse = {"flag": "RT_FLAG_REUSE_PRIMARY_HITS", "steps": n_reuse, "time_to_image_s": t_reuse,
                 "rays_reference_equivalent_per_step": float(tr[1]), "rays_executed_per_step": float(tr[2]),
                 "mrays_executed_per_s": float(tr[2]) / t_reuse / 1e6,
                 "extend_ms": st_r.trace_kernel_ms, "shade_ms": st_r.shade_kernel_ms, "other_ms": st_r.other_kernel_ms,
                 "note": "same pixels and sample counts as the headline render (tests/test_gpu_reuse.py); not part of `value`"}

    # ---- untimed instrumented render: device-side visit counts for the byte model and the lane utilisation
    cfg_cnt = cfg.copy()
    cfg_cnt.flags = abi.RT_FLAG_COUNT_VISITS | opt_flags
    if args.profile_pass:
        st_cnt = abi.RtStats()
    else:
        _, st_cnt = res.render(cfg_cnt, device_out=out)
    v_int_dev = st_cnt.node_visits / max(st_cnt.rays_traced, 1)
    v_leaf_dev = st_cnt.leaf_tests / max(st_cnt.rays_traced, 1)
    # fraction of the 64 lanes doing useful work when a wave executes a node step / a leaf test
    lanes_node = st_cnt.node_visits / max(64 * st_cnt.wave_node_steps, 1)
    lanes_leaf = st_cnt.leaf_tests / max(64 * st_cnt.wave_leaf_steps, 1)
    lanes_useful = (st_cnt.node_visits + st_cnt.leaf_tests) / max(64 * (st_cnt.wave_node_steps + st_cnt.wave_leaf_steps), 1)

    # ---- CPU baseline: the oracle (a port of the reference's Rayon path) on a bounded row sample
    cpu = None
    v_int, v_leaf = v_int_dev, v_leaf_dev
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from oracle import oracle
        hc = host_cpu()
        threads = hc["threads_used"]
        ref_cfg = ri.config.copy()
        ref_cfg.rng_mode = abi.RT_RNG_REF
        probe = ref_cfg.copy()
        probe.row_first, probe.row_stride = H // 2, H  # one row through the middle of the image
        # compiled on this machine for this machine (-O3 -march=native, still no contraction and no fast-math: same bits)
        with oracle.native_build():
            cpu_flags = oracle.flags()
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
            "compiler_flags": cpu_flags,
            "host": hc,
            "single_core_value": st_probe.rays / max(st_probe.wall_ms, 1e-6) / 1e3,
            "sample": f"rows {sample.row_first}::{stride} of the same {W}x{H} workload ({cpu_rows} rows, "
                      f"{int(st_cpu.rays)} rays, {st_cpu.wall_ms / 1e3:.1f} s, reference RNG mode, {threads} threads = every core "
                      f"this process may use)",
            "rays": int(st_cpu.rays),
            "seconds": st_cpu.wall_ms / 1e3,
            "time_to_image_s_extrapolated": st_cpu.wall_ms / 1e3 * H / cpu_rows,
            "node_visits_per_ray": v_int,
            "leaf_tests_per_ray": v_leaf,
        }

    if rank == 0:
        bytes_per_ray = v_int * NODE_BYTES + v_leaf * TRI_BYTES + RAY_STATE_BYTES
        value = rays_all / elapsed / 1e6
        headline = (args.triangles, W, H, args.batch, args.max_batches, args.min_batches) == (100_000, 1200, 1200, 20, 100, 2)
        working_set = sc.n_nodes // 2 * 32 + sc.n_triangles * (80 + 32)  # grid nodes + triangle test and shade records
        counters, counters_note = None, "no counter sums for this workload (profiles/roofline_inputs.json covers the headline configuration)"
        if headline and not opt_flags:
            inputs, counters_note = load_roofline_inputs(ROOFLINE_INPUTS, abi.kernel_source_hash())
            if inputs:
                counters = counter_roofline(inputs, rays_traced, trace_ms, shade_ms, lanes_useful, launches)
                counters_note = counters.get("refused")
        ext = (counters or {}).get("k_extend")
        roofline = {
            "kernel": "k_extend",
            "launches": launches / args.steps,
            "avg_launch_ms": trace_ms / max(launches, 1),
            "kernel_ms_per_step": trace_ms / args.steps,
            "shade_kernel_ms_per_step": shade_ms / args.steps,
            "other_kernels_ms_per_step": other_ms / args.steps,
            "kernel_ms_by_depth_per_step": [x / args.steps for x in ms_by_depth[:cfg.max_depth]],
            "max_rank_kernel_ms_per_step": trace_ms_max / args.steps,
            "algorithmic_bytes_per_ray": bytes_per_ray,
            "algorithmic_bytes_per_launch": rays * bytes_per_ray / max(launches, 1),
            "algorithmic_GBs": (rays * bytes_per_ray) / (trace_ms * 1e-3) / 1e9 if trace_ms > 0 else 0.0,
            "working_set_bytes": working_set,
            # the memory roofline the byte model belongs to: only a scene whose records leave the caches is bound by it
            "hbm_roofline": ({"algorithmic_GBs": (rays * bytes_per_ray) / (trace_ms * 1e-3) / 1e9 if trace_ms > 0 else 0.0, "frac": None,
                              "reason": f"working set {working_set / 1e6:.1f} MB is served by L2 / Infinity Cache: the algorithmic bytes never reach HBM "
                                        "(counter traffic is in hbm_frac)"} if working_set < (256 << 20) else None),
            "kernel_source_hash": abi.kernel_source_hash(),
            "node_visits_per_ray": v_int, "leaf_tests_per_ray": v_leaf,
            "node_visits_per_ray_device": v_int_dev, "leaf_tests_per_ray_device": v_leaf_dev,
            "simd_lane_utilisation": {"node_steps": lanes_node, "leaf_tests": lanes_leaf, "all_steps": lanes_useful},
            "wave_cycle_shares": {k: float(getattr(st_cnt, "wave_cycles_" + k)) / max(st_cnt.wave_cycles_refill + st_cnt.wave_cycles_node + st_cnt.wave_cycles_leaf, 1)
                                  for k in ("refill", "node", "leaf")},
        }
        if working_set < (256 << 20) and ext:
            # the BVH lives in L2 / Infinity Cache: the kernel is bound by VALU issue on divergent lanes, not by HBM
            roofline.update({
                "bound": "valu-issue",
                "achieved": 100.0 * ext["valu_issue_frac"], "peak": 100.0, "unit": "% of VALU issue cycles spent on useful lanes",
                "frac": ext["valu_issue_frac"],
                "valu_busy": ext["valu_busy"], "useful_lane_share": ext["useful_lane_share"],
                "wait_share_of_wave_time": ext["wait_share_of_wave_time"],
                "hbm_frac": ext["hbm_frac"], "hbm_GBs": ext["fabric_GBs"], "l2_frac": ext["l2_frac"], "l2_GBs": ext["l2_GBs"],
                "l2_hit_rate": ext["l2_hit_rate"],
                "traffic": ext["fabric_bytes"] / max(launches, 1), "traffic_unit": "bytes per launch",
                "traffic_source": "FETCH_SIZE x fetch_factor + WRITE_SIZE of profiles/roofline_inputs.json (separate --pmc passes), scaled by rays traced",
                "profiled_avg_launch_ms": ext["profiled_avg_launch_ms"],
            })
        elif ext:
            roofline.update({
                "bound": "hbm", "achieved": ext["fabric_GBs"], "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ext["hbm_frac"],
                "traffic": ext["fabric_bytes"] / max(launches, 1), "traffic_unit": "bytes per launch",
            })
        else:
            # no counter sums for this workload: only the byte model, which is NOT a fraction of the HBM roof when the
            # working set is cache resident
            roofline.update({"bound": "hbm" if working_set >= (256 << 20) else "valu-issue", "achieved": None, "peak": None,
                             "unit": None, "frac": None, "traffic": None, "note": counters_note})
        if counters and "k_shade" in counters:
            s = counters["k_shade"]
            roofline["second_kernel"] = {"kernel": "k_shade", "bound": "hbm", "achieved": s["fabric_GBs"], "peak": HBM_PEAK_GBS,
                                         "unit": "GB/s", "frac": s["hbm_frac"], "valu_busy": s["valu_busy"], "l2_hit_rate": s["l2_hit_rate"],
                                         "kernel_ms_per_step": shade_ms / args.steps}
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
                "rows_per_rank": n_rows, "row_block": row_block,
                "parallelism": f"blocks of {row_block} row(s) dealt over {world} GPU(s), one gather to rank 0"
                               + (" inside librt_hip.so (rt_hip_render_ranks: RCCL sends and receives)" if comm is not None else ""),
                "early_terminate": bool(args.early_terminate), "reference_visit_rule": bool(args.reference_visits),
            },
            "rays_per_step": rays_all / args.steps,
            "rays_traced_per_step_rank0": rays_traced / args.steps,
            "mean_spp": samples_all / args.steps / (W * H),
            "rays_by_depth_rank0_last_step": rays_by_depth[:cfg.max_depth],
            "host_build_s": t_host,
            "upload_s": t_upload,
            "first_upload_of_the_process_s": t_upload_first,
            "device_build_s": t_device_build,  # rt_hip_scene_build instead of the host's tree + upload_s: copies, build, records
            "time_to_image_with_load_build_upload_s": elapsed / args.steps + t_host + t_upload,
            "time_to_image_with_load_build_upload_in_a_fresh_process_s": elapsed / args.steps + t_host + t_upload_first,
            "roofline": roofline,
            "reuse_primary_hits": reuse,
            "time_to_image_reuse_s": reuse["time_to_image_s"] if reuse else None,
            "mrays_executed": reuse["mrays_executed_per_s"] if reuse else None,
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
