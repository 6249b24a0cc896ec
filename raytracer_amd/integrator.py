"""Python face of the native host: the reference's RenderIntegrator surface
(/root/reference/src/render/integrator.rs:38-249 -- new_from_json, render,
save_to_png) on top of librt_host.so / librt_hip.so.

Nothing here computes pixels: every method forwards to the C ABI.
"""
import ctypes as C
import json
import os

import numpy as np

from . import abi


class ResidentScene:
    """A scene uploaded to one GPU (rt_hip_scene_upload)."""

    def __init__(self, scene_ptr, device=0, build_on_device=False):
        """Upload the scene with the host's BVH, or (build_on_device) let the GPU build its own from the primitives."""
        self._h = C.c_void_p()
        fn = abi.hip_lib().rt_hip_scene_build if build_on_device else abi.hip_lib().rt_hip_scene_upload
        abi.check_hip(fn(scene_ptr, int(device), C.byref(self._h)))
        self.device = int(device)

    def download_tree(self):
        """The resident tree as an array of RtNode (rt_hip_scene_download_tree)."""
        n = C.c_int32()
        abi.check_hip(abi.hip_lib().rt_hip_scene_download_tree(self._h, None, 0, C.byref(n)))
        nodes = (abi.RtNode * n.value)()
        abi.check_hip(abi.hip_lib().rt_hip_scene_download_tree(self._h, nodes, n.value, C.byref(n)))
        return nodes

    def tune(self, **params):
        """Set launch / scheduling parameters of this handle (rt_hip_scene_tune): res.tune(max_group=5, first_group=3)."""
        for name, value in params.items():
            abi.check_hip(abi.hip_lib().rt_hip_scene_tune(self._h, name.encode(), int(value)))
        return self

    def close(self):
        if self._h:
            abi.hip_lib().rt_hip_scene_free(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def render(self, cfg, want_linear=True, want_samples=True, device_out=None, want_rgb8=False):
        """Render the row shard named by cfg.  Returns (dict of numpy arrays, RtStats).

        With `device_out` (a dict of torch CUDA tensors named pixels / sample_pixels /
        linear / samples) the kernels write straight into those tensors.
        """
        n_rows, w = cfg.n_rows(), cfg.img_width
        img = abi.RtImage()
        stats = abi.RtStats()
        if device_out is not None:
            img.memory = abi.RT_MEM_DEVICE
            for name in ("pixels", "sample_pixels", "linear", "samples", "rgb8", "sample_rgb8"):
                t = device_out.get(name)
                if t is not None:
                    setattr(img, name, t.data_ptr())
            abi.check_hip(abi.hip_lib().rt_hip_render_resident(self._h, C.byref(cfg), C.byref(img), C.byref(stats)))
            return device_out, stats
        out = {
            "pixels": np.empty((n_rows, w, 3), dtype=np.float64),
            "sample_pixels": np.empty((n_rows, w, 3), dtype=np.float64),
        }
        if want_linear:
            out["linear"] = np.empty((n_rows, w, 3), dtype=np.float64)
        if want_samples:
            out["samples"] = np.empty((n_rows, w), dtype=np.uint32)
        if want_rgb8:  # what save_to_png writes (color.rs:47-54), converted on the device
            out["rgb8"] = np.empty((n_rows, w, 3), dtype=np.uint8)
            out["sample_rgb8"] = np.empty((n_rows, w, 3), dtype=np.uint8)
        img.memory = abi.RT_MEM_HOST
        for name, arr in out.items():
            setattr(img, name, arr.ctypes.data)
        abi.check_hip(abi.hip_lib().rt_hip_render_resident(self._h, C.byref(cfg), C.byref(img), C.byref(stats)))
        return out, stats

    def set_progress(self, fn):
        """fn(fraction_done) after every round of a render on this scene (the reference's progress
        bar, integrator.rs:242); None removes it."""
        self._progress = abi.RtProgressFn(lambda _user, frac: fn(frac)) if fn else None  # keep the thunk alive
        ptr = C.cast(self._progress, C.c_void_p) if self._progress else None
        abi.check_hip(abi.hip_lib().rt_hip_set_progress(self._h, ptr, None))

    def primary_hits(self, cfg):
        n = cfg.n_rows() * cfg.img_width * cfg.sample_batch_size
        prim = np.empty(n, dtype=np.int32)
        t = np.empty(n, dtype=np.float64)
        abi.check_hip(abi.hip_lib().rt_hip_primary_hits(self._h, C.byref(cfg), prim.ctypes.data, t.ctypes.data))
        shape = (cfg.n_rows(), cfg.img_width, cfg.sample_batch_size)
        return prim.reshape(shape), t.reshape(shape)

    def trace_rays(self, origins, directions):
        o = np.ascontiguousarray(origins, dtype=np.float64).reshape(-1, 3)
        d = np.ascontiguousarray(directions, dtype=np.float64).reshape(-1, 3)
        assert o.shape == d.shape
        n = o.shape[0]
        prim = np.empty(n, dtype=np.int32)
        t = np.empty(n, dtype=np.float64)
        abi.check_hip(abi.hip_lib().rt_hip_trace_rays(self._h, n, o.ctypes.data, d.ctypes.data, prim.ctypes.data, t.ctypes.data))
        return prim, t


    def trace_rays_noalloc(self, d_origins, d_directions, d_prim_id, d_t):
        """rt_hip_trace_rays_noalloc on caller-owned device buffers (anything with data_ptr(): [n,3] f64
        origins and directions, [n] i32 ids, [n] f64 t, all on this scene's device)."""
        n = int(d_prim_id.numel())
        assert d_origins.numel() == 3 * n and d_directions.numel() == 3 * n and d_t.numel() == n
        abi.check_hip(abi.hip_lib().rt_hip_trace_rays_noalloc(self._h, n, d_origins.data_ptr(), d_directions.data_ptr(),
                                                              d_prim_id.data_ptr(), d_t.data_ptr()))


class RenderIntegrator:
    """Mirror of the reference's RenderIntegrator (integrator.rs:30-36)."""

    def __init__(self, handle):
        self._h = handle
        self._resident = {}

    # integrator.rs:63-82
    @classmethod
    def new_from_json(cls, scene_path, config_path, base_dir=None):
        h = C.c_void_p()
        abi.check_host(abi.host_lib().rt_host_new_from_json(
            os.fsencode(scene_path), os.fsencode(config_path), os.fsencode(base_dir) if base_dir else None, C.byref(h)))
        return cls(h)

    @classmethod
    def new(cls, json_scene, config, base_dir=None):
        """integrator.rs:39-61 with the scene / config given as Python dicts (or JSON text)."""
        s = json_scene if isinstance(json_scene, str) else json.dumps(json_scene)
        c = config if isinstance(config, str) else json.dumps(config)
        h = C.c_void_p()
        abi.check_host(abi.host_lib().rt_host_new_from_strings(
            s.encode(), c.encode(), os.fsencode(base_dir) if base_dir else None, C.byref(h)))
        return cls(h)

    def close(self):
        for r in self._resident.values():
            r.close()
        self._resident = {}
        if self._h:
            abi.host_lib().rt_host_free(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_device_build(self, on=True):
        """Before prepare(): leave the BVH build to the GPU (rt_host_set_device_build); render() then builds the same tree there."""
        abi.check_host(abi.host_lib().rt_host_set_device_build(self._h, int(bool(on))))
        return self

    # integrator.rs:163-192
    def prepare(self):
        abi.check_host(abi.host_lib().rt_host_prepare(self._h))
        return self

    @property
    def scene(self):
        """The RtScene POD (valid after prepare)."""
        p = abi.host_lib().rt_host_scene(self._h)
        if not p:
            raise RuntimeError("call prepare() first")
        return p

    @property
    def config(self):
        """The RtConfig POD (editable in place)."""
        return abi.host_lib().rt_host_config(self._h).contents

    def bvh_depth(self):
        return abi.host_lib().rt_host_bvh_depth(self._h)

    def element_bbox(self, i):
        mn, mx = (C.c_double * 3)(), (C.c_double * 3)()
        abi.check_host(abi.host_lib().rt_host_element_bbox(self._h, i, mn, mx))
        return tuple(mn), tuple(mx)

    def resident(self, device=0):
        """The scene resident on `device` (uploaded once, reused by repeated renders)."""
        self.prepare()
        if device not in self._resident:
            self._resident[device] = ResidentScene(self.scene, device)
        return self._resident[device]

    # integrator.rs:161-249
    def render(self, device=0):
        stats = abi.RtStats()
        abi.check_host(abi.host_lib().rt_host_render(self._h, int(device), C.byref(stats)))
        return stats

    def render_devices(self, devices):
        """render() split over several GPUs of this process: rows r::n go to devices[r], one host thread each."""
        stats = abi.RtStats()
        arr = (C.c_int * len(devices))(*[int(d) for d in devices])
        abi.check_host(abi.host_lib().rt_host_render_devices(self._h, arr, len(devices), C.byref(stats)))
        return stats

    def _buf(self, ptr):
        cfg = self.config
        n = cfg.img_width * cfg.img_height * 3
        return np.ctypeslib.as_array(ptr, shape=(n,)).reshape(cfg.img_height, cfg.img_width, 3)

    @property
    def pixels(self):
        return self._buf(abi.host_lib().rt_host_pixels(self._h))

    @property
    def sample_pixels(self):
        return self._buf(abi.host_lib().rt_host_sample_pixels(self._h))

    def set_pixels(self, pixels, sample_pixels=None):
        p = np.ascontiguousarray(pixels, dtype=np.float64)
        s = None if sample_pixels is None else np.ascontiguousarray(sample_pixels, dtype=np.float64)
        abi.check_host(abi.host_lib().rt_host_set_pixels(self._h, p.ctypes.data, None if s is None else s.ctypes.data))

    # integrator.rs:84-156
    def save_to_png(self, png_path, samples_path=None):
        abi.check_host(abi.host_lib().rt_host_save_to_png(
            self._h, os.fsencode(png_path), os.fsencode(samples_path) if samples_path else None))
