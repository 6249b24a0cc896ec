"""The C-ABI libraries load without a GPU and export every function their headers
declare; the ctypes mirror of the structures has the C compiler's layout."""
import ctypes as C
import os
import re
import subprocess

import pytest

from raytracer_amd import abi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared(header):
    text = open(os.path.join(ROOT, "include", header)).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(rt_(?:hip|host)_\w+)\s*\(", text)))


def test_hip_library_exports_every_declared_symbol():
    names = _declared("rt_hip.h")
    assert set(names) == set(abi.HIP_SYMBOLS)
    lib = abi.hip_lib()
    for n in names:
        assert getattr(lib, n) is not None, n


def test_host_library_exports_every_declared_symbol():
    names = _declared("rt_host.h")
    assert set(names) == set(abi.HOST_SYMBOLS)
    lib = abi.host_lib()
    for n in names:
        assert getattr(lib, n) is not None, n


def test_headers_compile_as_plain_c_and_struct_layout_matches_ctypes(tmp_path):
    src = tmp_path / "layout.c"
    fields = {
        "RtVec3": abi.RtVec3, "RtCamera": abi.RtCamera, "RtMaterial": abi.RtMaterial, "RtTriangle": abi.RtTriangle,
        "RtQuad": abi.RtQuad, "RtSphere": abi.RtSphere, "RtElementRef": abi.RtElementRef, "RtNode": abi.RtNode,
        "RtScene": abi.RtScene, "RtConfig": abi.RtConfig, "RtImage": abi.RtImage, "RtStats": abi.RtStats,
    }
    lines = ['#include <stdio.h>', '#include <stddef.h>', '#include "rt_host.h"', "int main(void) {"]
    for name, cls in fields.items():
        lines.append(f'  printf("{name} %zu\\n", sizeof({name}));')
        for fname, _ in cls._fields_:
            lines.append(f'  printf("{name}.{fname} %zu\\n", offsetof({name}, {fname}));')
    lines += ["  return 0;", "}"]
    src.write_text("\n".join(lines))
    exe = tmp_path / "layout"
    subprocess.check_call(["gcc", "-std=c99", "-pedantic", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), "-o", str(exe), str(src)])
    out = dict(l.split() for l in subprocess.check_output([str(exe)]).decode().splitlines())
    for name, cls in fields.items():
        assert int(out[name]) == C.sizeof(cls), name
        for fname, _ in cls._fields_:
            assert int(out[f"{name}.{fname}"]) == getattr(cls, fname).offset, f"{name}.{fname}"


def test_errors_come_back_as_codes_without_a_gpu():
    lib = abi.hip_lib()
    assert lib.rt_hip_scene_upload(None, 0, None) == abi.RT_ERR_BAD_ARG
    assert b"NULL" in lib.rt_hip_last_error()
    assert lib.rt_hip_render_resident(None, None, None, None) == abi.RT_ERR_BAD_ARG
    assert lib.rt_hip_primary_hits(None, None, None, None) == abi.RT_ERR_BAD_ARG
    assert lib.rt_hip_trace_rays(None, 0, None, None, None, None) == abi.RT_ERR_BAD_ARG
    assert lib.rt_hip_scene_free(None) == abi.RT_OK
    n = lib.rt_hip_device_count()
    assert n >= 0 or n in (abi.RT_ERR_HIP,)


def test_scene_validation_rejects_bad_indices_before_any_kernel():
    from helpers import c1
    ri = c1(width=8, height=8, batch=1)
    sc = ri.scene.contents
    lib = abi.hip_lib()
    h = C.c_void_p()

    def upload(scene):
        return lib.rt_hip_scene_upload(C.byref(scene), 0, C.byref(h))

    def copy():
        s = abi.RtScene()
        C.memmove(C.byref(s), C.byref(sc), C.sizeof(abi.RtScene))
        return s

    s = copy(); s.n_nodes = 0
    assert upload(s) == abi.RT_ERR_BAD_ARG
    nodes = (abi.RtNode * sc.n_nodes)()
    C.memmove(nodes, sc.nodes, C.sizeof(nodes))
    nodes[0].child_r_index = 0  # a cycle
    s = copy(); s.nodes = nodes
    assert upload(s) == abi.RT_ERR_BAD_ARG and b"child index" in lib.rt_hip_last_error()
    C.memmove(nodes, sc.nodes, C.sizeof(nodes))
    leaf = next(i for i in range(sc.n_nodes) if nodes[i].is_leaf)
    nodes[leaf].element_index = sc.n_elements
    assert upload(s) == abi.RT_ERR_BAD_ARG
    tris = (abi.RtTriangle * sc.n_triangles)()
    C.memmove(tris, sc.triangles, C.sizeof(tris))
    tris[3].material = 99
    s = copy(); s.triangles = tris
    assert upload(s) == abi.RT_ERR_BAD_ARG and b"material" in lib.rt_hip_last_error()
    att = (C.c_int32 * 1)(-5)
    s = copy(); s.attractors = att
    assert upload(s) == abi.RT_ERR_BAD_ARG
    # only spheres and quads can be attractors (elements/mod.rs:68-75): a triangle element is refused
    tri_element = next(i for i in range(sc.n_elements) if sc.elements[i].kind == abi.RT_PRIM_TRIANGLE)
    att = (C.c_int32 * 1)(tri_element)
    s = copy(); s.attractors = att
    assert upload(s) == abi.RT_ERR_BAD_ARG and b"not a quad or a sphere" in lib.rt_hip_last_error()


def test_the_package_has_no_cpu_rendering_path():
    """The product must not import the oracle or fall back to it."""
    pkg = os.path.join(ROOT, "raytracer_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".cpp", ".hpp", ".h", ".hip")):
                text = open(os.path.join(dirpath, f)).read()
                assert "oracle" not in text.replace("CPU oracle", "").replace("the oracle", ""), os.path.join(dirpath, f)


def test_rust_bindings_are_generated_from_the_same_structures():
    """bindings/rust/rt-hip-sys/src/lib.rs is written from the ctypes mirror whose layout the test
    above checks against gcc; the committed file must be what the generator writes today."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "gen_rust_bindings.py"), "--check"], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    text = open(os.path.join(root, "bindings", "rust", "rt-hip-sys", "src", "lib.rs")).read()
    for sym in abi.HIP_SYMBOLS:
        assert f"pub fn {sym}(" in text, sym
