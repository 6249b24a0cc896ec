import os, sys, tempfile
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from raytracer_amd import RenderIntegrator, abi, parallel, scenes
workdir = os.path.join(tempfile.gettempdir(), f"rt_bench_{os.getuid()}_0")
sp, cp = scenes.materialize_c3(workdir, 100_000, 1200, 1200)
ri = RenderIntegrator.new_from_json(sp, cp).prepare()
res = ri.resident(0)
for n in (1, 8):
    cfg = parallel.shard_config(ri.config, 0, n, 1)
    cfg.flags = abi.RT_FLAG_TIME_KERNELS
    res.render(cfg, want_linear=False, want_samples=False)
    _, st = res.render(cfg, want_linear=False, want_samples=False)
    print(n, "rounds", st.batches, "extend by depth x n:", " ".join(f"{n*x:6.1f}" for x in list(st.trace_ms_by_depth)[:8]), "rays by depth x n (M):", " ".join(f"{n*x/1e6:6.0f}" for x in list(st.rays_by_depth)[:8]), "shade", n*st.shade_kernel_ms, "other", n*st.other_kernel_ms, flush=True)
