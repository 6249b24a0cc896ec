"""Synthetic inputs in the reference's JSON schema (render/config.rs:8-20,
render/camera.rs:6-13, elements/mod.rs:48-55).

The reference's own large meshes (input/obj/dragon.obj, xl_dragon.obj) are
not distributed with it, so the benchmark scenes substitute a deterministic
generated mesh of the same triangle count in the dragon's place.  Everything
else -- camera, Cornell walls, the rotated metal box and the quad light with
`attractor: true` -- carries the values of the reference's input/scene.json.
"""
import json
import os

import numpy as np


def _v(x, y, z):
    return {"x": float(x), "y": float(y), "z": float(z)}


def _rgb(r, g, b):
    return {"r": float(r), "g": float(g), "b": float(b)}


def lambertian(r, g, b):
    return {"Lambertian": {"albedo": _rgb(r, g, b)}}


def metal(r, g, b, fuzz):
    return {"Metal": {"albedo": _rgb(r, g, b), "fuzz": float(fuzz)}}


def dielectric(ior):
    return {"Dielectric": {"index_of_refraction": float(ior)}}


def diffuse_light(r, g, b):
    return {"DiffuseLight": {"albedo": _rgb(r, g, b)}}


def camera(center=(278.0, 278.0, -800.0), look_at=(278.0, 278.0, 0.0), fov=40.0, defocus_angle=0.0, focus_dist=800.0):
    """input/scene.json:2-16"""
    return {
        "camera_center": _v(*center),
        "camera_look_at": _v(*look_at),
        "camera_fov_vertical": float(fov),
        "camera_defocus_angle": float(defocus_angle),
        "camera_focus_dist": float(focus_dist),
    }


GREEN, RED, WHITE = (0.12, 0.45, 0.15), (0.65, 0.05, 0.05), (0.73, 0.73, 0.73)

# The twelve wall / floor / ceiling triangles of input/scene.json:81-418, in
# file order (two of them are exact duplicates of earlier ones, which is what
# exercises the traversal's equal-t tie rule).
_CORNELL_TRIANGLES = [
    ((555, 0, 0), (555, 0, 555), (555, 555, 555), GREEN),
    ((555, 0, 0), (555, 555, 555), (555, 555, 0), GREEN),
    ((0, 0, 0), (0, 555, 555), (0, 0, 555), RED),
    ((0, 0, 0), (0, 555, 0), (0, 555, 555), RED),
    ((0, 555, 555), (555, 555, 0), (555, 555, 555), WHITE),
    ((0, 0, 0), (0, 555, 0), (0, 555, 555), RED),
    ((0, 0, 555), (555, 0, 555), (555, 0, 0), WHITE),
    ((0, 0, 555), (555, 0, 0), (0, 0, 0), WHITE),
    ((0, 555, 555), (555, 555, 0), (555, 555, 555), WHITE),
    ((0, 555, 555), (0, 555, 0), (555, 555, 0), WHITE),
    ((0, 0, 555), (0, 555, 555), (555, 0, 555), WHITE),
    ((555, 0, 555), (0, 555, 555), (555, 555, 555), WHITE),
]


def cornell_elements(light_attractor=True):
    """Box + walls + quad light of input/scene.json:47-447."""
    els = [{
        "JSONBox": {
            "a": _v(0, 0, 0), "b": _v(165, 330, 165),
            "material": metal(0.3, 0.3, 0.83, 0.1),
            "transpose": _v(265, 0, 295),
            "rotate": {"theta_x": 0.0, "theta_y": -0.25, "theta_z": 0.0},
        }
    }]
    for v0, v1, v2, col in _CORNELL_TRIANGLES:
        els.append({"JSONTriangle": {"v0": _v(*v0), "v1": _v(*v1), "v2": _v(*v2), "material": lambertian(*col)}})
    light = {"q": _v(343, 554, 332), "u": _v(-130, 0, 0), "v": _v(0, 0, -105), "material": diffuse_light(10, 8, 8)}
    if light_attractor:
        light["attractor"] = True
    els.append({"JSONQuad": light})
    return els


def sphere(center, radius, material, attractor=None, transpose=None):
    s = {"center": _v(*center), "radius": float(radius), "material": material}
    if attractor is not None:
        s["attractor"] = bool(attractor)
    if transpose is not None:
        s["transpose"] = _v(*transpose)
    return {"JSONSphere": s}


def obj_element(filepath, material, transpose=None, rotate=None, scale=None):
    o = {"filepath": filepath, "material": material}
    if transpose is not None:
        o["transpose"] = _v(*transpose)
    if rotate is not None:
        o["rotate"] = {"theta_x": float(rotate[0]), "theta_y": float(rotate[1]), "theta_z": float(rotate[2])}
    if scale is not None:
        o["scale"] = _v(*scale)
    return {"JSONObj": o}


def config(width, height, batch, max_batches, min_batches, max_depth=8, split="SAH", gamma=2.4, pixel_radius=2.0):
    """input/config.json with the sizes replaced."""
    c = {
        "img_width": float(width), "img_height": float(height),
        "sample_batch_size": int(batch), "max_sample_batches": int(max_batches), "min_sample_batches": int(min_batches),
        "max_depth": int(max_depth),
        "sky_color": _rgb(0.3, 0.5, 0.9),
        "pixel_radius": float(pixel_radius),
        "gamma_correction": float(gamma),
    }
    if split is not None:
        c["bvh_split_method"] = split
    return c


def scene_c1(defocus_angle=0.0):
    """Cornell box + Lambertian / Metal / Dielectric spheres standing where the
    reference's scene.json puts its (undistributed) dragon mesh."""
    els = [
        sphere((150, 80, 170), 80, lambertian(0.43, 0.53, 0.43)),
        sphere((310, 60, 90), 60, dielectric(1.5)),
        sphere((450, 70, 200), 70, metal(0.8, 0.7, 0.6, 0.2)),
    ] + cornell_elements()
    return {"camera": camera(defocus_angle=defocus_angle), "elements": els}


def config_c1(width=400, height=400):
    """Fixed 64 spp: one batch of 64, so the z-test never runs."""
    return config(width, height, 64, 1, 1)


# ------------------------------------------------------------- synthetic mesh
def knot_mesh(n_triangles=100_000, seed=7):
    """A closed, consistently wound tube around a (2,3) torus knot with a
    seeded sinusoidal displacement: exactly `n_triangles` triangles (must be
    even).  Returns (vertices [V,3] f64, faces [T,3] int, 0-based)."""
    if n_triangles % 2 or n_triangles < 24:
        raise ValueError("n_triangles must be even and >= 24")
    quads = n_triangles // 2
    # nu segments along the curve, nv around the tube, nu*nv == quads
    nv = 1
    for cand in range(int(np.sqrt(quads / 20.0)), 2, -1):
        if quads % cand == 0:
            nv = cand
            break
    if nv < 3:
        nv = 4 if quads % 4 == 0 else 3
        if quads % nv:
            raise ValueError("cannot factor the triangle count into a tube grid")
    nu = quads // nv
    rng = np.random.RandomState(seed)
    ph = rng.uniform(0, 2 * np.pi, size=4)
    u = np.arange(nu) * (2 * np.pi / nu)

    def curve(t):
        r = 2.0 + np.cos(3 * t)
        return np.stack([r * np.cos(2 * t), r * np.sin(2 * t), -1.2 * np.sin(3 * t)], axis=-1)

    h = 1e-4
    c = curve(u)
    d1 = (curve(u + h) - curve(u - h)) / (2 * h)
    d2 = (curve(u + h) - 2 * c + curve(u - h)) / (h * h)
    tan = d1 / np.linalg.norm(d1, axis=1, keepdims=True)
    nrm = d2 - (d2 * tan).sum(1, keepdims=True) * tan
    nrm /= np.linalg.norm(nrm, axis=1, keepdims=True)
    bin_ = np.cross(tan, nrm)
    v = np.arange(nv) * (2 * np.pi / nv)
    uu, vv = np.meshgrid(u, v, indexing="ij")
    radius = 0.42 * (1.0 + 0.18 * np.sin(9 * uu + ph[0]) * np.sin(4 * vv + ph[1]) + 0.07 * np.sin(31 * uu + 5 * vv + ph[2]))
    pos = c[:, None, :] + radius[..., None] * (np.cos(vv)[..., None] * nrm[:, None, :] + np.sin(vv)[..., None] * bin_[:, None, :])
    verts = pos.reshape(-1, 3)
    i = np.arange(nu)[:, None]
    j = np.arange(nv)[None, :]
    a = (i * nv + j).ravel()
    b = (((i + 1) % nu) * nv + j).ravel()
    cc = (((i + 1) % nu) * nv + (j + 1) % nv).ravel()
    d = (i * nv + (j + 1) % nv).ravel()
    faces = np.concatenate([np.stack([a, b, cc], 1), np.stack([a, cc, d], 1)], 0)
    # orient outward: the face normal must point away from the tube's centre line
    f0 = faces[0]
    fn = np.cross(verts[f0[1]] - verts[f0[0]], verts[f0[2]] - verts[f0[0]])
    out = verts[f0].mean(0) - c[0]
    if np.dot(fn, out) < 0:
        faces = faces[:, ::-1].copy()
    assert faces.shape[0] == n_triangles
    return verts, faces


def _write_obj_part(args):
    path, kind, rows = args
    with open(path, "w") as f:
        np.savetxt(f, rows, fmt="v %.6f %.6f %.6f" if kind == "v" else "f %d %d %d")
    return path


def write_obj(path, verts, faces, workers=None):
    """Plain `v` / `f` Wavefront file, 1-based indices, like input/obj/teapot.obj.  Files of millions of lines are
    formatted by several processes (pieces written side by side, then joined): formatting text is all the time there is."""
    lines = len(verts) + len(faces)
    if workers is None:
        workers = 1 if lines < 2_000_000 else max(1, min(8, len(os.sched_getaffinity(0))))
    if workers <= 1:
        with open(path, "w") as f:
            f.write("# generated by raytracer_amd.scenes.knot_mesh\n")
            np.savetxt(f, verts, fmt="v %.6f %.6f %.6f")
            np.savetxt(f, faces + 1, fmt="f %d %d %d")
        return
    import shutil
    from concurrent.futures import ProcessPoolExecutor
    jobs = []
    for kind, rows in (("v", verts), ("f", faces + 1)):
        for k, piece in enumerate(np.array_split(rows, 2 * workers)):
            jobs.append((f"{path}.part_{kind}{k:03d}", kind, piece))
    with ProcessPoolExecutor(max_workers=workers) as pool:
        parts = list(pool.map(_write_obj_part, jobs))
    with open(path, "wb") as out:
        out.write(b"# generated by raytracer_amd.scenes\n")
        for part in parts:
            with open(part, "rb") as f:
                shutil.copyfileobj(f, out, 1 << 24)
            os.remove(part)


def scene_c3(obj_path, defocus_angle=0.0):
    """README-like scene: Cornell box + metal box + quad light + glass sphere +
    a ~100k-triangle Lambertian mesh in the dragon's slot."""
    els = [
        obj_element(obj_path, lambertian(0.43, 0.53, 0.43), transpose=(190.0, 110.0, 170.0), rotate=(0.0, -3.2, 0.0),
                    scale=(30.0, 30.0, 30.0)),
        sphere((420, 70, 150), 70, dielectric(1.5)),
    ] + cornell_elements()
    return {"camera": camera(defocus_angle=defocus_angle), "elements": els}


def config_c3(width=1200, height=1200, batch=20, max_batches=100, min_batches=2):
    """README render: batches of 20, at most 100 batches (2000 spp), at least 2."""
    return config(width, height, batch, max_batches, min_batches)


def materialize_c3(directory, n_triangles=100_000, width=1200, height=1200, seed=7, batch=20, max_batches=100, min_batches=2):
    """Write mesh.obj + scene.json + config.json for the C3 workload; returns the two JSON paths."""
    os.makedirs(directory, exist_ok=True)
    obj_path = os.path.join(directory, f"knot_{n_triangles}.obj")
    if not os.path.exists(obj_path):
        verts, faces = knot_mesh(n_triangles, seed)
        tmp = obj_path + ".tmp"
        write_obj(tmp, verts, faces)
        os.replace(tmp, obj_path)
    scene_path = os.path.join(directory, "scene.json")
    config_path = os.path.join(directory, "config.json")
    with open(scene_path, "w") as f:
        json.dump(scene_c3(obj_path), f)
    with open(config_path, "w") as f:
        json.dump(config_c3(width, height, batch, max_batches, min_batches), f)
    return scene_path, config_path


# --------------------------------------------------------------- triangle soup
def soup_elements(n_triangles, seed=12345):
    """Random triangle soup: centres uniform in the Cornell interior [20, 535]^3, edge vectors
    uniform in [-4, 4]^3; 10 % dielectric (1.5), 10 % metal (fuzz U[0, 0.3]), 80 % Lambertian
    (albedo U[0.2, 0.9]^3).  Winding is whatever the edges give: about half of the triangles
    face away from any given ray and are culled (triangle.rs:88)."""
    rng = np.random.RandomState(seed)
    c = rng.uniform(20.0, 535.0, size=(n_triangles, 3))
    e1 = rng.uniform(-4.0, 4.0, size=(n_triangles, 3))
    e2 = rng.uniform(-4.0, 4.0, size=(n_triangles, 3))
    kind = rng.uniform(size=n_triangles)
    albedo = rng.uniform(0.2, 0.9, size=(n_triangles, 3))
    fuzz = rng.uniform(0.0, 0.3, size=n_triangles)
    els = []
    for i in range(n_triangles):
        if kind[i] < 0.1:
            m = dielectric(1.5)
        elif kind[i] < 0.2:
            m = metal(albedo[i, 0], albedo[i, 1], albedo[i, 2], fuzz[i])
        else:
            m = lambertian(albedo[i, 0], albedo[i, 1], albedo[i, 2])
        v0 = c[i] - (e1[i] + e2[i]) / 3.0
        els.append({"JSONTriangle": {"v0": _v(*v0), "v1": _v(*(v0 + e1[i])), "v2": _v(*(v0 + e2[i])), "material": m}})
    return els


def scene_soup(n_triangles, seed=12345, defocus_angle=0.6, focus_dist=1000.0):
    """Cornell room + quad light (attractor) filled with a triangle soup, thin-lens camera."""
    els = soup_elements(n_triangles, seed) + cornell_elements()
    return {"camera": camera(defocus_angle=defocus_angle, focus_dist=focus_dist), "elements": els}


def materialize_soup(directory, n_triangles, width, height, batch=16, max_batches=4, min_batches=2, seed=12345):
    """The triangle-soup scene at sizes JSON elements cannot carry: the soup of soup_elements
    (same distributions) written as three OBJ files, one per material class -- Lambertian
    (0.55, 0.55, 0.55), Metal (0.7, 0.7, 0.7) fuzz 0.15, Dielectric 1.5 -- because an OBJ element
    has one material.  Returns the scene and config JSON paths."""
    os.makedirs(directory, exist_ok=True)
    rng = np.random.RandomState(seed)
    c = rng.uniform(20.0, 535.0, size=(n_triangles, 3))
    e1 = rng.uniform(-4.0, 4.0, size=(n_triangles, 3))
    e2 = rng.uniform(-4.0, 4.0, size=(n_triangles, 3))
    kind = rng.uniform(size=n_triangles)
    v0 = c - (e1 + e2) / 3.0
    groups = [("lambertian", kind >= 0.2, lambertian(0.55, 0.55, 0.55)), ("metal", (kind >= 0.1) & (kind < 0.2), metal(0.7, 0.7, 0.7, 0.15)),
              ("glass", kind < 0.1, dielectric(1.5))]
    els = []
    for name, mask, material in groups:
        k = int(mask.sum())
        if k == 0:
            continue
        path = os.path.join(directory, f"soup_{n_triangles}_{name}.obj")
        if not os.path.exists(path):
            verts = np.empty((3 * k, 3))
            verts[0::3], verts[1::3], verts[2::3] = v0[mask], v0[mask] + e1[mask], v0[mask] + e2[mask]
            faces = np.arange(3 * k).reshape(k, 3)
            tmp = path + ".tmp"
            write_obj(tmp, verts, faces)
            os.replace(tmp, path)
        els.append(obj_element(path, material))
    scene = {"camera": camera(defocus_angle=0.6, focus_dist=1000.0), "elements": els + cornell_elements()}
    scene_path, config_path = os.path.join(directory, "scene_soup.json"), os.path.join(directory, "config_soup.json")
    with open(scene_path, "w") as f:
        json.dump(scene, f)
    with open(config_path, "w") as f:
        json.dump(config(width, height, batch, max_batches, min_batches), f)
    return scene_path, config_path
