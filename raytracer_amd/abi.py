"""ctypes mirror of include/rt_hip.h and include/rt_host.h.

The product path is native: `librt_hip.so` (HIP kernels + C ABI) and
`librt_host.so` (C++ host: scene loading, BVH build, RenderIntegrator).  This
module only declares the C structures and loads the two libraries; it raises
if they are missing -- there is no Python or CPU fallback for rendering.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_DIR = os.path.join(_HERE, "lib")

RT_OK = 0
RT_ERR_BAD_ARG = -1
RT_ERR_HIP = -2
RT_ERR_OOM = -3
RT_ERR_UNSUPPORTED = -4

RT_PRIM_SPHERE, RT_PRIM_QUAD, RT_PRIM_TRIANGLE = 0, 1, 2
RT_MAT_LAMBERTIAN, RT_MAT_METAL, RT_MAT_DIELECTRIC, RT_MAT_DIFFUSE_LIGHT = 0, 1, 2, 3
RT_RNG_REF, RT_RNG_CTR = 0, 1
RT_FLAG_TIME_KERNELS, RT_FLAG_COUNT_VISITS, RT_FLAG_EARLY_TERMINATE, RT_FLAG_REFERENCE_VISIT_RULE = 1, 2, 4, 8
RT_FLAG_REUSE_PRIMARY_HITS = 16
RT_MEM_HOST, RT_MEM_DEVICE = 0, 1
RT_SPLIT_SAH, RT_SPLIT_MID = 0, 1  # RtScene.split_method
RT_MAX_DEPTH_STATS = 16
RT_NOALLOC_MAX_RAYS = 131072


class RtVec3(C.Structure):
    _fields_ = [("x", C.c_double), ("y", C.c_double), ("z", C.c_double)]

    def tuple(self):
        return (self.x, self.y, self.z)


class RtCamera(C.Structure):
    _fields_ = [
        ("camera_center", RtVec3),
        ("pixel00_loc", RtVec3),
        ("pixel_delta_u", RtVec3),
        ("pixel_delta_v", RtVec3),
        ("defocus_disk_u", RtVec3),
        ("defocus_disk_v", RtVec3),
        ("defocus_angle", C.c_double),
    ]


class RtMaterial(C.Structure):
    _fields_ = [("kind", C.c_int32), ("reserved", C.c_int32), ("albedo", C.c_double * 3), ("param", C.c_double)]


class RtTriangle(C.Structure):
    _fields_ = [("v0", RtVec3), ("v0v1", RtVec3), ("v0v2", RtVec3), ("normal", RtVec3),
                ("material", C.c_int32), ("reserved", C.c_int32)]


class RtQuad(C.Structure):
    _fields_ = [("q", RtVec3), ("u", RtVec3), ("v", RtVec3), ("normal", RtVec3), ("w", RtVec3),
                ("d", C.c_double), ("area", C.c_double), ("material", C.c_int32), ("attractor", C.c_int32)]


class RtSphere(C.Structure):
    _fields_ = [("center", RtVec3), ("radius", C.c_double), ("material", C.c_int32), ("attractor", C.c_int32)]


class RtElementRef(C.Structure):
    _fields_ = [("kind", C.c_int32), ("index", C.c_int32)]


class RtNode(C.Structure):
    _fields_ = [
        ("is_leaf", C.c_int32), ("element_index", C.c_int32), ("child_l_index", C.c_int32), ("child_r_index", C.c_int32),
        ("child_l_min", C.c_double * 3), ("child_l_max", C.c_double * 3),
        ("child_r_min", C.c_double * 3), ("child_r_max", C.c_double * 3),
    ]


class RtScene(C.Structure):
    _fields_ = [
        ("camera", RtCamera),
        ("n_nodes", C.c_int32), ("n_elements", C.c_int32), ("n_triangles", C.c_int32), ("n_quads", C.c_int32),
        ("n_spheres", C.c_int32), ("n_materials", C.c_int32), ("n_attractors", C.c_int32), ("split_method", C.c_int32),
        ("nodes", C.POINTER(RtNode)),
        ("elements", C.POINTER(RtElementRef)),
        ("triangles", C.POINTER(RtTriangle)),
        ("quads", C.POINTER(RtQuad)),
        ("spheres", C.POINTER(RtSphere)),
        ("materials", C.POINTER(RtMaterial)),
        ("attractors", C.POINTER(C.c_int32)),
    ]


class RtConfig(C.Structure):
    _fields_ = [
        ("img_width", C.c_int32), ("img_height", C.c_int32),
        ("sample_batch_size", C.c_int32), ("max_sample_batches", C.c_int32),
        ("min_sample_batches", C.c_int32), ("max_depth", C.c_int32),
        ("pixel_radius", C.c_double), ("gamma_correction", C.c_double),
        ("filter_b", C.c_double), ("filter_c", C.c_double),
        ("rng_mode", C.c_int32), ("flags", C.c_int32),
        ("seed", C.c_uint64),
        ("row_first", C.c_int32), ("row_stride", C.c_int32),
        ("row_block", C.c_int32), ("work_mib", C.c_int32),
    ]

    def copy(self):
        c = RtConfig()
        C.memmove(C.byref(c), C.byref(self), C.sizeof(RtConfig))
        return c

    def rows(self):
        """Image rows of the shard, in the order the output buffers hold them."""
        b = max(1, self.row_block)
        n_blocks = (self.img_height + b - 1) // b
        return [y for blk in range(self.row_first, n_blocks, self.row_stride) for y in range(blk * b, min((blk + 1) * b, self.img_height))]

    def n_rows(self):
        b = max(1, self.row_block)
        n_blocks = (self.img_height + b - 1) // b
        if self.row_first >= n_blocks:
            return 0
        mine = (n_blocks - self.row_first + self.row_stride - 1) // self.row_stride
        rows = mine * b
        if self.row_first + (mine - 1) * self.row_stride == n_blocks - 1 and self.img_height % b:
            rows -= b - self.img_height % b
        return rows


class RtImage(C.Structure):
    _fields_ = [
        ("pixels", C.c_void_p), ("sample_pixels", C.c_void_p), ("linear", C.c_void_p), ("samples", C.c_void_p),
        ("rgb8", C.c_void_p), ("sample_rgb8", C.c_void_p),
        ("memory", C.c_int32), ("reserved", C.c_int32),
    ]


class RtStats(C.Structure):
    _fields_ = [
        ("rays", C.c_uint64),
        ("rays_traced", C.c_uint64),
        ("rays_by_depth", C.c_uint64 * RT_MAX_DEPTH_STATS),
        ("samples", C.c_uint64),
        ("batches", C.c_uint64),
        ("node_visits", C.c_uint64),
        ("box_tests", C.c_uint64),
        ("leaf_tests", C.c_uint64),
        ("wall_ms", C.c_double),
        ("trace_kernel_ms", C.c_double),
        ("trace_launches", C.c_uint64),
        ("shade_kernel_ms", C.c_double),
        ("other_kernel_ms", C.c_double),
        ("trace_ms_by_depth", C.c_double * RT_MAX_DEPTH_STATS),
        ("wave_node_steps", C.c_uint64),
        ("wave_leaf_steps", C.c_uint64),
        ("wave_cycles_refill", C.c_uint64),
        ("wave_cycles_node", C.c_uint64),
        ("wave_cycles_leaf", C.c_uint64),
        ("stack_block_moves", C.c_uint64),
    ]


HIP_SYMBOLS = [
    "rt_hip_last_error", "rt_hip_device_count", "rt_hip_scene_upload", "rt_hip_scene_free",
    "rt_hip_render_resident", "rt_hip_render", "rt_hip_primary_hits", "rt_hip_trace_rays", "rt_hip_set_progress",
    "rt_hip_trace_rays_noalloc", "rt_hip_scene_build", "rt_hip_scene_download_tree", "rt_hip_scene_tune",
    "rt_hip_comm_id", "rt_hip_comm_init", "rt_hip_comm_free", "rt_hip_render_ranks",
]
RT_COMM_ID_BYTES = 128
RtProgressFn = C.CFUNCTYPE(None, C.c_void_p, C.c_double)
HOST_SYMBOLS = [
    "rt_host_last_error", "rt_host_new_from_json", "rt_host_new_from_strings", "rt_host_free", "rt_host_prepare",
    "rt_host_scene", "rt_host_config", "rt_host_bvh_depth", "rt_host_element_bbox", "rt_host_render", "rt_host_render_devices",
    "rt_host_pixels", "rt_host_sample_pixels", "rt_host_set_pixels", "rt_host_save_to_png", "rt_host_set_device_build",
]

_hip = None
_host = None


class RtError(RuntimeError):
    def __init__(self, code, message):
        super().__init__(f"[{code}] {message}")
        self.code = code


def hip_lib():
    """librt_hip.so; raises if the HIP extension has not been built."""
    global _hip
    if _hip is None:
        path = os.path.join(LIB_DIR, "librt_hip.so")
        if not os.path.exists(path):
            raise RuntimeError(f"{path} is missing: run `make` (or __graft_entry__.build()) first; there is no fallback path")
        lib = C.CDLL(path, mode=C.RTLD_GLOBAL)
        lib.rt_hip_last_error.restype = C.c_char_p
        lib.rt_hip_device_count.restype = C.c_int
        lib.rt_hip_scene_upload.argtypes = [C.POINTER(RtScene), C.c_int, C.POINTER(C.c_void_p)]
        lib.rt_hip_scene_free.argtypes = [C.c_void_p]
        lib.rt_hip_render_resident.argtypes = [C.c_void_p, C.POINTER(RtConfig), C.POINTER(RtImage), C.POINTER(RtStats)]
        lib.rt_hip_render.argtypes = [C.POINTER(RtScene), C.POINTER(RtConfig), C.POINTER(RtImage), C.POINTER(RtStats)]
        lib.rt_hip_primary_hits.argtypes = [C.c_void_p, C.POINTER(RtConfig), C.c_void_p, C.c_void_p]
        lib.rt_hip_trace_rays.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        lib.rt_hip_set_progress.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        lib.rt_hip_scene_build.argtypes = [C.POINTER(RtScene), C.c_int, C.POINTER(C.c_void_p)]
        lib.rt_hip_scene_download_tree.argtypes = [C.c_void_p, C.POINTER(RtNode), C.c_int32, C.POINTER(C.c_int32)]
        lib.rt_hip_trace_rays_noalloc.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        lib.rt_hip_scene_tune.argtypes = [C.c_void_p, C.c_char_p, C.c_int64]
        lib.rt_hip_comm_id.argtypes = [C.c_char_p]
        lib.rt_hip_comm_init.argtypes = [C.c_char_p, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_void_p)]
        lib.rt_hip_comm_free.argtypes = [C.c_void_p]
        lib.rt_hip_render_ranks.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(RtConfig), C.POINTER(RtImage), C.POINTER(RtStats)]
        _hip = lib
    return _hip


def host_lib():
    """librt_host.so (links librt_hip.so)."""
    global _host
    if _host is None:
        hip_lib()
        path = os.path.join(LIB_DIR, "librt_host.so")
        if not os.path.exists(path):
            raise RuntimeError(f"{path} is missing: run `make` (or __graft_entry__.build()) first")
        lib = C.CDLL(path)
        lib.rt_host_last_error.restype = C.c_char_p
        lib.rt_host_new_from_json.argtypes = [C.c_char_p, C.c_char_p, C.c_char_p, C.POINTER(C.c_void_p)]
        lib.rt_host_new_from_strings.argtypes = [C.c_char_p, C.c_char_p, C.c_char_p, C.POINTER(C.c_void_p)]
        lib.rt_host_free.argtypes = [C.c_void_p]
        lib.rt_host_free.restype = None
        lib.rt_host_prepare.argtypes = [C.c_void_p]
        lib.rt_host_scene.argtypes = [C.c_void_p]
        lib.rt_host_scene.restype = C.POINTER(RtScene)
        lib.rt_host_config.argtypes = [C.c_void_p]
        lib.rt_host_config.restype = C.POINTER(RtConfig)
        lib.rt_host_bvh_depth.argtypes = [C.c_void_p]
        lib.rt_host_element_bbox.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_double)]
        lib.rt_host_render.argtypes = [C.c_void_p, C.c_int, C.POINTER(RtStats)]
        lib.rt_host_render_devices.argtypes = [C.c_void_p, C.POINTER(C.c_int), C.c_int, C.POINTER(RtStats)]
        lib.rt_host_set_device_build.argtypes = [C.c_void_p, C.c_int]
        lib.rt_host_pixels.argtypes = [C.c_void_p]
        lib.rt_host_pixels.restype = C.POINTER(C.c_double)
        lib.rt_host_sample_pixels.argtypes = [C.c_void_p]
        lib.rt_host_sample_pixels.restype = C.POINTER(C.c_double)
        lib.rt_host_set_pixels.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        lib.rt_host_save_to_png.argtypes = [C.c_void_p, C.c_char_p, C.c_char_p]
        _host = lib
    return _host


KERNEL_SOURCES = ("csrc/rt_hip.hip", "csrc/rt_device.h", "csrc/rt_build_device.h", "csrc/rt_build_topdown.h")


def kernel_source_hash():
    """SHA-256 (first 16 hex digits) over the sources librt_hip.so is compiled from: what a profile was taken on."""
    import hashlib
    h = hashlib.sha256()
    for rel in KERNEL_SOURCES:
        with open(os.path.join(_HERE, rel), "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:16]


def check_hip(rc):
    if rc != RT_OK:
        raise RtError(rc, hip_lib().rt_hip_last_error().decode())


def check_host(rc):
    if rc != RT_OK:
        raise RtError(rc, host_lib().rt_host_last_error().decode())
