import os, sys, tempfile
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from raytracer_amd import RenderIntegrator, abi, scenes
workdir = os.path.join(tempfile.gettempdir(), f"rt_bench_{os.getuid()}_0")
sp, cp = scenes.materialize_c3(workdir, 100_000, 1200, 1200)
ri = RenderIntegrator.new_from_json(sp, cp).prepare()
res = ri.resident(0)
prev = (0, 0, 0)
for d in range(1, 5):
    cfg = ri.config.copy()
    cfg.max_depth = d
    cfg.max_sample_batches = 4; cfg.min_sample_batches = 4
    cfg.flags = abi.RT_FLAG_COUNT_VISITS
    _, st = res.render(cfg)
    cur = (st.rays_traced, st.node_visits, st.leaf_tests)
    dr, dn, dl = (cur[0]-prev[0], cur[1]-prev[1], cur[2]-prev[2])
    print(f"depth {d-1}: rays {dr/1e6:8.1f} M  node visits/ray {dn/dr:6.2f}  leaf tests/ray {dl/dr:5.2f}  lanes node {st.node_visits/(64*st.wave_node_steps):.3f} leaf {st.leaf_tests/(64*st.wave_leaf_steps):.3f}", flush=True)
    prev = cur
