"""rt_hip_trace_rays_noalloc: the closest-hit queries of rt_hip_trace_rays with every working
buffer in a fixed array of the library's code object.  Same ids and t, bit for bit, on two calls
in a row (the second smaller: the array still holds the first call's contents), no device memory
allocated by the calls, and a clear refusal beyond the stated maximum."""
import ctypes as C

import numpy as np
import pytest

from raytracer_amd import abi

from helpers import c1, knot_scene, random_rays

pytestmark = pytest.mark.gpu


def _device_call(torch, res, o, d):
    n = o.shape[0]
    d_o = torch.from_numpy(np.ascontiguousarray(o)).cuda()
    d_d = torch.from_numpy(np.ascontiguousarray(d)).cuda()
    d_id = torch.full((n,), -7, dtype=torch.int32, device="cuda")
    d_t = torch.full((n,), -7.0, dtype=torch.float64, device="cuda")
    torch.cuda.synchronize()
    free0 = torch.cuda.mem_get_info()[0]
    res.trace_rays_noalloc(d_o, d_d, d_id, d_t)
    free1 = torch.cuda.mem_get_info()[0]
    assert free1 == free0, "the call changed the amount of free device memory"
    return d_id.cpu().numpy(), d_t.cpu().numpy()


def test_noalloc_queries_equal_trace_rays_on_two_calls_in_a_row(tmp_scene_dir):
    torch = pytest.importorskip("torch")
    ri = knot_scene(tmp_scene_dir, 20000, 16, 16, 1, 1, 1)
    res = ri.resident(0)
    for n, seed in ((abi.RT_NOALLOC_MAX_RAYS, 41), (30_001, 42), (1, 43)):
        o, d = random_rays(n, seed=seed)
        want_id, want_t = res.trace_rays(o, d)
        got_id, got_t = _device_call(torch, res, o, d)
        assert np.array_equal(got_id, want_id), n
        assert np.array_equal(got_t, want_t), n
        assert (want_id >= 0).mean() > 0.5 or n == 1


def test_noalloc_on_a_tree_whose_stacks_leave_lds(monkeypatch, tmp_scene_dir):
    """Trees too high for on-chip stacks move blocks of stack entries to a buffer in device memory: it belongs to the scene
    (allocated at set-up), so the query still allocates nothing.  Forced here on a tree of moderate height."""
    torch = pytest.importorskip("torch")
    from raytracer_amd.integrator import ResidentScene
    ri = knot_scene(tmp_scene_dir, 20000, 16, 16, 1, 1, 1)
    assert ri.bvh_depth() > 16
    plain = ri.resident(0)  # uploaded before the variable is set: every stack in LDS
    monkeypatch.setenv("RT_HIP_SPILL_ABOVE", "16")
    res = ResidentScene(ri.scene, 0)
    for n, seed in ((100_000, 44), (5_000, 45)):
        o, d = random_rays(n, seed=seed)
        want_id, want_t = plain.trace_rays(o, d)
        got_id, got_t = _device_call(torch, res, o, d)
        assert np.array_equal(got_id, want_id) and np.array_equal(got_t, want_t)
    res.close()


def test_noalloc_follows_the_reference_visit_rule_where_the_scene_needs_it():
    """A scene with quads that stick out of their boxes is walked in two passes: both sets of claim counters are used."""
    torch = pytest.importorskip("torch")
    from test_gpu_parity import _tilted_quad_scene
    ri = _tilted_quad_scene()
    res = ri.resident(0)
    for n, seed in ((50_000, 5), (777, 6)):
        o, d = random_rays(n, seed=seed)
        want_id, want_t = res.trace_rays(o, d)
        got_id, got_t = _device_call(torch, res, o, d)
        assert np.array_equal(got_id, want_id) and np.array_equal(got_t, want_t)


def test_noalloc_rejects_sizes_beyond_its_maximum_and_null_buffers():
    ri = c1(width=8, height=8, batch=1)
    res = ri.resident(0)
    lib = abi.hip_lib()
    rc = lib.rt_hip_trace_rays_noalloc(res._h, abi.RT_NOALLOC_MAX_RAYS + 1, C.c_void_p(8), C.c_void_p(8), C.c_void_p(8), C.c_void_p(8))
    assert rc == abi.RT_ERR_UNSUPPORTED
    msg = lib.rt_hip_last_error().decode()
    assert str(abi.RT_NOALLOC_MAX_RAYS) in msg and "at most" in msg
    assert lib.rt_hip_trace_rays_noalloc(res._h, 5, None, None, None, None) == abi.RT_ERR_BAD_ARG
    assert lib.rt_hip_trace_rays_noalloc(res._h, 0, None, None, None, None) == abi.RT_OK


def test_noalloc_calls_from_two_threads_on_two_scenes_wait_for_each_other(tmp_scene_dir):
    """Both scenes' queries go through the GPU's one array: the library lets one call through at a time."""
    import threading
    torch = pytest.importorskip("torch")
    scenes_ = [knot_scene(tmp_scene_dir, 20000, 16, 16, 1, 1, 1), c1(width=8, height=8, batch=1)]
    res = [ri.resident(0) for ri in scenes_]
    rays = [random_rays(60_000, seed=51), random_rays(60_000, seed=52)]
    want = [res[k].trace_rays(*rays[k]) for k in range(2)]
    bufs = []
    for k in range(2):
        o, d = rays[k]
        bufs.append((torch.from_numpy(np.ascontiguousarray(o)).cuda(), torch.from_numpy(np.ascontiguousarray(d)).cuda(),
                     torch.empty(o.shape[0], dtype=torch.int32, device="cuda"), torch.empty(o.shape[0], dtype=torch.float64, device="cuda")))
    torch.cuda.synchronize()
    errors = []

    def work(k):
        try:
            for _ in range(20):
                res[k].trace_rays_noalloc(*bufs[k])
                if not (np.array_equal(bufs[k][2].cpu().numpy(), want[k][0]) and np.array_equal(bufs[k][3].cpu().numpy(), want[k][1])):
                    errors.append(k)
        except Exception as e:  # noqa: BLE001
            errors.append(repr(e))

    ts = [threading.Thread(target=work, args=(k,)) for k in range(2)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    assert not errors, errors


def test_noalloc_inputs_made_on_another_stream_must_be_finished_first(tmp_scene_dir):
    """The call runs on a stream of the library's own and waits for nobody else's work: inputs produced on a caller's
    (non-blocking) stream are synchronised by the caller first, as the header says; then the results are those of
    rt_hip_trace_rays."""
    torch = pytest.importorskip("torch")
    ri = knot_scene(tmp_scene_dir, 20000, 16, 16, 1, 1, 1)
    res = ri.resident(0)
    o, d = random_rays(90_000, seed=61)
    want_id, want_t = res.trace_rays(o, d)
    side = torch.cuda.Stream()  # non-blocking with respect to the legacy default stream
    pinned_o, pinned_d = torch.from_numpy(np.ascontiguousarray(o)).pin_memory(), torch.from_numpy(np.ascontiguousarray(d)).pin_memory()
    d_id = torch.full((o.shape[0],), -7, dtype=torch.int32, device="cuda")
    d_t = torch.full((o.shape[0],), -7.0, dtype=torch.float64, device="cuda")
    torch.cuda.synchronize()
    with torch.cuda.stream(side):
        d_o = pinned_o.to("cuda", non_blocking=True)
        d_d = (pinned_d.to("cuda", non_blocking=True) * 3.0) / 3.0 * 1.0  # a kernel on the side stream too
        d_d = pinned_d.to("cuda", non_blocking=True)
    side.synchronize()  # the contract: inputs complete before the call
    res.trace_rays_noalloc(d_o, d_d, d_id, d_t)
    # results are complete on return: read them on yet another stream without further synchronisation
    with torch.cuda.stream(torch.cuda.Stream()):
        got_id, got_t = d_id.cpu().numpy(), d_t.cpu().numpy()
    assert np.array_equal(got_id, want_id) and np.array_equal(got_t, want_t)


def test_noalloc_queries_are_not_disturbed_by_scene_set_up_on_the_same_gpu(tmp_scene_dir):
    """Scene set-up traces one ray through the same array (its ray sits in the array's last bytes, which a full-size
    query uses for hit references): it takes the GPU's lock like any query, so uploads overlapping full-size queries
    leave every result intact."""
    import threading
    torch = pytest.importorskip("torch")
    from raytracer_amd.integrator import ResidentScene
    ri = knot_scene(tmp_scene_dir, 20000, 16, 16, 1, 1, 1)
    small = c1(width=8, height=8, batch=1)
    res = ri.resident(0)
    o, d = random_rays(abi.RT_NOALLOC_MAX_RAYS, seed=71)
    want_id, want_t = res.trace_rays(o, d)
    buf = (torch.from_numpy(np.ascontiguousarray(o)).cuda(), torch.from_numpy(np.ascontiguousarray(d)).cuda(),
           torch.empty(o.shape[0], dtype=torch.int32, device="cuda"), torch.empty(o.shape[0], dtype=torch.float64, device="cuda"))
    torch.cuda.synchronize()
    errors, stop = [], threading.Event()

    def uploads():
        try:
            while not stop.is_set():
                ResidentScene(small.scene, 0).close()
                ResidentScene(small.scene, 0, build_on_device=True).close()
        except Exception as e:  # noqa: BLE001
            errors.append(repr(e))

    t = threading.Thread(target=uploads)
    t.start()
    try:
        for _ in range(25):
            res.trace_rays_noalloc(*buf)
            if not (np.array_equal(buf[2].cpu().numpy(), want_id) and np.array_equal(buf[3].cpu().numpy(), want_t)):
                errors.append("a query returned other hits while a scene was being set up")
                break
    finally:
        stop.set()
        t.join()
    assert not errors, errors
