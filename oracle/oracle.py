"""ctypes loader for the CPU oracle (oracle/rt_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, by the smoke check and by the
benchmark's CPU-baseline leg.  Nothing under raytracer_amd/ imports it.
"""
import ctypes as C
import os
import subprocess

import numpy as np

from raytracer_amd import abi

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "_build", "librt_oracle.so")
_lib = None


# -ffp-contract=off -fno-fast-math: the Rust reference never fuses a multiply-add and never reassociates.
BASE_FLAGS = ["-O3", "-std=gnu11", "-fPIC", "-shared", "-ffp-contract=off", "-fno-fast-math", "-Wall"]
_flags = " ".join(BASE_FLAGS)  # what the library in use was built with


def build(out=_LIB, extra_flags=()):
    os.makedirs(os.path.dirname(out), exist_ok=True)
    subprocess.check_call(["gcc"] + BASE_FLAGS + list(extra_flags) + ["-o", out, os.path.join(_HERE, "rt_oracle.c"), "-lm", "-lpthread"])


def _load(path):
    l = C.CDLL(path)
    l.rt_oracle_render.argtypes = [C.POINTER(abi.RtScene), C.POINTER(abi.RtConfig), C.POINTER(abi.RtImage),
                                   C.POINTER(abi.RtStats), C.c_int]
    l.rt_oracle_primary_hits.argtypes = [C.POINTER(abi.RtScene), C.POINTER(abi.RtConfig), C.c_void_p, C.c_void_p]
    l.rt_oracle_trace_rays.argtypes = [C.POINTER(abi.RtScene), C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    l.rt_oracle_sobol_sample.restype = C.c_float
    l.rt_oracle_sobol_sample.argtypes = [C.c_uint32, C.c_uint32, C.c_uint32]
    l.rt_oracle_filter_evaluate.restype = C.c_double
    l.rt_oracle_filter_evaluate.argtypes = [C.c_double] * 5
    l.rt_oracle_quad_pdf_value.restype = C.c_double
    l.rt_oracle_sphere_pdf_value.restype = C.c_double
    return l


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB):
            build()
        _lib = _load(_LIB)
    return _lib


def flags():
    """The compiler flags of the library lib() returns."""
    return "gcc " + _flags


class native_build:
    """Context manager: the oracle compiled ON THIS MACHINE with -march=native (the in-tree library is built portable,
    because it travels to machines with other CPUs), for the benchmark's CPU-baseline leg.  Results are the same to the
    bit -- nothing in the flags licenses another rounding -- which tests/test_golden.py checks.  If the compiler is
    missing or fails, the in-tree library stays in use and flags() says so."""

    def __enter__(self):
        global _lib, _flags
        import tempfile
        self._saved = (_lib, _flags)
        self._dir = tempfile.TemporaryDirectory(prefix="rt_oracle_native_")
        path = os.path.join(self._dir.name, "librt_oracle_native.so")
        try:
            build(path, ["-march=native"])
            _lib, _flags = _load(path), " ".join(BASE_FLAGS + ["-march=native"])
        except (OSError, subprocess.CalledProcessError):
            lib()
        return self

    def __exit__(self, *exc):
        global _lib, _flags
        _lib, _flags = self._saved
        self._dir.cleanup()
        return False


def _d3(v):
    return (C.c_double * 3)(*[float(x) for x in v])


def render(scene_ptr, cfg, n_threads=8):
    """rt_oracle_render -> (dict of numpy arrays, RtStats)."""
    n_rows, w = cfg.n_rows(), cfg.img_width
    out = {
        "pixels": np.zeros((n_rows, w, 3)), "sample_pixels": np.zeros((n_rows, w, 3)),
        "linear": np.zeros((n_rows, w, 3)), "samples": np.zeros((n_rows, w), dtype=np.uint32),
    }
    img = abi.RtImage()
    for k, a in out.items():
        setattr(img, k, a.ctypes.data)
    stats = abi.RtStats()
    rc = lib().rt_oracle_render(scene_ptr, C.byref(cfg), C.byref(img), C.byref(stats), int(n_threads))
    if rc != 0:
        raise RuntimeError(f"rt_oracle_render failed: {rc}")
    return out, stats


def primary_hits(scene_ptr, cfg):
    shape = (cfg.n_rows(), cfg.img_width, cfg.sample_batch_size)
    n = shape[0] * shape[1] * shape[2]
    prim = np.empty(n, dtype=np.int32)
    t = np.empty(n, dtype=np.float64)
    rc = lib().rt_oracle_primary_hits(scene_ptr, C.byref(cfg), prim.ctypes.data, t.ctypes.data)
    if rc != 0:
        raise RuntimeError(f"rt_oracle_primary_hits failed: {rc}")
    return prim.reshape(shape), t.reshape(shape)


def primary_hits_threads(scene_ptr, cfg, n_threads=8):
    """primary_hits with the shard's rows dealt to `n_threads` Python threads (the C call releases the GIL)."""
    from concurrent.futures import ThreadPoolExecutor
    n_rows = cfg.n_rows()
    n_threads = max(1, min(int(n_threads), n_rows))
    shape = (n_rows, cfg.img_width, cfg.sample_batch_size)
    prim = np.empty(shape, dtype=np.int32)
    t = np.empty(shape, dtype=np.float64)

    b = max(1, cfg.row_block)
    n_blocks = (n_rows + b - 1) // b  # blocks of the shard; thread k takes the shard's blocks k, k + n_threads, ...

    def part(k):
        c = cfg.copy()
        c.row_first = cfg.row_first + k * cfg.row_stride
        c.row_stride = cfg.row_stride * n_threads
        mine = [r for j in range(k, n_blocks, n_threads) for r in range(j * b, min((j + 1) * b, n_rows))]
        return mine, (primary_hits(scene_ptr, c) if mine else (None, None))

    with ThreadPoolExecutor(n_threads) as pool:
        for mine, (p, tt) in pool.map(part, range(n_threads)):
            if mine:
                prim[mine], t[mine] = p, tt
    return prim, t


def trace_rays(scene_ptr, origins, directions, want_visits=False):
    o = np.ascontiguousarray(origins, dtype=np.float64).reshape(-1, 3)
    d = np.ascontiguousarray(directions, dtype=np.float64).reshape(-1, 3)
    n = o.shape[0]
    prim = np.empty(n, dtype=np.int32)
    t = np.empty(n, dtype=np.float64)
    visits = np.zeros((n, 3), dtype=np.uint32) if want_visits else None
    rc = lib().rt_oracle_trace_rays(scene_ptr, n, o.ctypes.data, d.ctypes.data, prim.ctypes.data, t.ctypes.data,
                                    visits.ctypes.data if want_visits else None)
    if rc != 0:
        raise RuntimeError(f"rt_oracle_trace_rays failed: {rc}")
    return (prim, t, visits) if want_visits else (prim, t)


def aabb_hit(bmin, bmax, origin, direction):
    return bool(lib().rt_oracle_aabb_hit(_d3(bmin), _d3(bmax), _d3(origin), _d3(direction)))


def ray_new(origin, direction):
    out, signs = (C.c_double * 9)(), (C.c_int * 3)()
    lib().rt_oracle_ray_new(_d3(origin), _d3(direction), out, signs)
    return np.array(out[:]).reshape(3, 3), list(signs)


def ray_at(origin, direction, t):
    out = (C.c_double * 3)()
    lib().rt_oracle_ray_at(_d3(origin), _d3(direction), C.c_double(t), out)
    return np.array(out[:])


def vec3_ops(a, b):
    out = (C.c_double * 10)()
    lib().rt_oracle_vec3_ops(_d3(a), _d3(b), out)
    o = list(out)
    return {"dot": o[0], "cross": o[1:4], "length": o[4], "normalized": o[5:8], "distance": o[8], "length_squared": o[9]}


def color_ops(c, gamma):
    cl, g, rgb, lum = (C.c_double * 3)(), (C.c_double * 3)(), (C.c_uint8 * 3)(), (C.c_double * 2)()
    lib().rt_oracle_color_ops(_d3(c), C.c_double(gamma), cl, g, rgb, lum)
    return {"clamp": list(cl), "gamma": list(g), "rgb": list(rgb), "illuminance": lum[0], "illuminance_approx": lum[1]}


def sphere_hit(center, radius, origin, direction, tmin, tmax):
    t, p, n, ff = C.c_double(), (C.c_double * 3)(), (C.c_double * 3)(), C.c_int()
    ok = lib().rt_oracle_sphere_hit(_d3(center), C.c_double(radius), _d3(origin), _d3(direction), C.c_double(tmin),
                                    C.c_double(tmax), C.byref(t), p, n, C.byref(ff))
    return None if not ok else {"t": t.value, "point": list(p), "normal": list(n), "front_face": bool(ff.value)}


def triangle_hit(tri, origin, direction, tmin, tmax):
    t = C.c_double()
    ok = lib().rt_oracle_triangle_hit(C.byref(tri), _d3(origin), _d3(direction), C.c_double(tmin), C.c_double(tmax), C.byref(t))
    return t.value if ok else None


def quad_hit(quad, origin, direction, tmin, tmax):
    t = C.c_double()
    ok = lib().rt_oracle_quad_hit(C.byref(quad), _d3(origin), _d3(direction), C.c_double(tmin), C.c_double(tmax), C.byref(t))
    return t.value if ok else None


def quad_pdf_value(quad, origin, direction):
    return lib().rt_oracle_quad_pdf_value(C.byref(quad), _d3(origin), _d3(direction))


def sphere_pdf_value(sph, origin, direction):
    return lib().rt_oracle_sphere_pdf_value(C.byref(sph), _d3(origin), _d3(direction))


def sobol_sample(index, dim, seed):
    return float(lib().rt_oracle_sobol_sample(index, dim, seed))


def filter_evaluate(sx, sy, radius, b=1. / 3., c=1. / 3.):
    return lib().rt_oracle_filter_evaluate(sx, sy, radius, b, c)


def rng_stream(mode, seed, kind, count, lo=0.0, hi=1.0, n=1):
    u = (C.c_uint64 * count)()
    f = (C.c_double * count)()
    lib().rt_oracle_rng_stream(C.c_int(mode), C.c_uint64(seed), C.c_int(kind), C.c_double(lo), C.c_double(hi),
                               C.c_uint32(n), C.c_int(count), u, f)
    return list(f) if kind in (1, 2) else list(u)


def stats(colors, weights):
    c = np.ascontiguousarray(colors, dtype=np.float64)
    w = np.ascontiguousarray(weights, dtype=np.float64)
    mean, var, iv = (C.c_double * 3)(), C.c_double(), C.c_double()
    lib().rt_oracle_stats(c.ctypes.data_as(C.POINTER(C.c_double)), w.ctypes.data_as(C.POINTER(C.c_double)), C.c_int(len(w)),
                          mean, C.byref(var), C.byref(iv))
    return list(mean), var.value, iv.value


def sample_offset(pixel_radius, x, y, i):
    sx, sy = C.c_double(), C.c_double()
    lib().rt_oracle_sample_offset(C.c_double(pixel_radius), x, y, i, C.byref(sx), C.byref(sy))
    return sx.value, sy.value


def onb(w):
    out = (C.c_double * 9)()
    lib().rt_oracle_onb(_d3(w), out)
    return np.array(out[:]).reshape(3, 3)
