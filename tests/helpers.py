"""Shared scene builders for the tests (small versions of the benchmark configs)."""
import os

import numpy as np

from raytracer_amd import RenderIntegrator, abi, scenes

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def c1(width=48, height=48, batch=8, max_batches=1, min_batches=1, defocus=0.0, split="SAH", max_depth=8):
    cfg = scenes.config(width, height, batch, max_batches, min_batches, max_depth=max_depth, split=split)
    return RenderIntegrator.new(scenes.scene_c1(defocus_angle=defocus), cfg).prepare()


def knot_scene(directory, n_triangles, width, height, batch, max_batches, min_batches, split="SAH", defocus=0.0):
    """C3-like scene with a small generated mesh."""
    os.makedirs(directory, exist_ok=True)
    obj = os.path.join(directory, f"knot_{n_triangles}.obj")
    if not os.path.exists(obj):
        v, f = scenes.knot_mesh(n_triangles)
        scenes.write_obj(obj, v, f)
    cfg = scenes.config(width, height, batch, max_batches, min_batches, split=split)
    return RenderIntegrator.new(scenes.scene_c3(obj, defocus_angle=defocus), cfg).prepare()


def random_rays(n, seed=0):
    """Rays from inside and around the Cornell room in random directions."""
    rng = np.random.RandomState(seed)
    o = rng.uniform(5.0, 550.0, size=(n, 3))
    d = rng.normal(size=(n, 3))
    # a share of camera-like rays from outside the room
    k = n // 3
    o[:k] = np.array([278.0, 278.0, -800.0]) + rng.normal(scale=3.0, size=(k, 3))
    d[:k] = rng.uniform(0.0, 555.0, size=(k, 3)) * np.array([1.0, 1.0, 0.0]) - o[:k]
    return o, d


# ------------------------------------------------------------ the reference's own input files
# Data the reference ships (input/scene.json, input/scene_teapot.json, input/config.json, input/obj/teapot.obj,
# input/obj/diamond.obj), kept under tests/golden/ref_inputs as inputs for the loader and the renders.
REF_INPUTS = os.path.join(GOLDEN, "ref_inputs")


def ref_config(**overrides):
    """input/config.json (600x600, batches of 20, up to 500, SAH, gamma 2.4) with fields replaced."""
    import json
    with open(os.path.join(REF_INPUTS, "config.json")) as f:
        cfg = json.load(f)
    cfg.update(overrides)
    return cfg


def soup_scene(n_triangles, width, height, batch, max_batches, min_batches, split="SAH"):
    """The triangle soup of the largest configuration in small: random triangles of mixed materials in the Cornell room, a lens."""
    return RenderIntegrator.new(scenes.scene_soup(n_triangles), scenes.config(width, height, batch, max_batches, min_batches, split=split)).prepare()


def teapot_scene(width, height, batch, max_batches, min_batches, split="SAH"):
    """input/scene_teapot.json: teapot.obj (6 320 triangles, scaled, rotated, moved) + metal box + walls +
    quad light (attractor), with input/config.json resized."""
    import json
    with open(os.path.join(REF_INPUTS, "scene_teapot.json")) as f:
        scene = json.load(f)
    cfg = ref_config(img_width=float(width), img_height=float(height), sample_batch_size=int(batch),
                     max_sample_batches=int(max_batches), min_sample_batches=int(min_batches), bvh_split_method=split)
    return RenderIntegrator.new(scene, cfg, base_dir=REF_INPUTS).prepare()


def readme_scene(directory, n_triangles, width, height, batch, max_batches, min_batches):
    """input/scene.json, the scene of the reference's README render: a dragon from input/obj/dragon.obj (scaled 22x, rotated,
    moved) + metal box + walls + quad light (attractor).  The reference does not distribute dragon.obj: a generated mesh
    of `n_triangles` is written under that name (scaled to the dragon's size: unit-sized coordinates), everything else is
    the reference's file as it stands, with input/config.json resized."""
    import json
    import shutil
    base = os.path.join(directory, f"readme_{n_triangles}")
    os.makedirs(os.path.join(base, "input", "obj"), exist_ok=True)
    obj = os.path.join(base, "input", "obj", "dragon.obj")
    if not os.path.exists(obj):
        v, f = scenes.knot_mesh(n_triangles)
        v = (v - v.mean(axis=0)) / np.abs(v - v.mean(axis=0)).max() * 5.0 + np.array([0.0, 5.0, 0.0])  # about 10 units across, standing on y = 0
        scenes.write_obj(obj + ".tmp", v, f)
        os.replace(obj + ".tmp", obj)
    shutil.copy(os.path.join(REF_INPUTS, "scene.json"), os.path.join(base, "scene.json"))
    with open(os.path.join(base, "scene.json")) as f:
        scene = json.load(f)
    cfg = ref_config(img_width=float(width), img_height=float(height), sample_batch_size=int(batch),
                     max_sample_batches=int(max_batches), min_sample_batches=int(min_batches))
    return RenderIntegrator.new(scene, cfg, base_dir=base).prepare()


# ------------------------------------------------------------------------------ image gates
# Device and oracle draw the same per-path random streams, so their linear images agree to rounding
# except where a last-bit difference in sin / cos / pow flips a path across an edge.  Measured on
# every BASELINE configuration: all pixels equal.  The gates leave room for one such pixel in a thousand
# and bound what it may do to the image.
PIXEL_GATE = 0.999        # share of pixels whose linear radiance equals the oracle's to 1e-9
COUNT_GATE = 0.999        # share of pixels that stop at the oracle's sample count
RMSE_EQUAL_GATE = 1e-9    # RMSE of the linear radiance over the pixels that agree
RMSE_NOISE_SHARE = 0.05   # RMSE over ALL pixels <= this share of the image's own Monte-Carlo noise, i.e. of the
                          # RMSE between two device renders with different seeds (a flipped path changes one
                          # sample of one pixel; the noise is what all samples of all pixels scatter by)


def image_parity(out_g, out_o):
    lg, lo = out_g["linear"], out_o["linear"]
    finite = np.isfinite(lo) & np.isfinite(lg)
    close = (np.isclose(lg, lo, rtol=1e-9, atol=1e-9, equal_nan=True) | ((lg == lo) & ~finite)).all(axis=-1)
    diff = np.where(finite, lg - lo, 0.0)
    same_counts = (out_g["samples"] == out_o["samples"]) if "samples" in out_g and "samples" in out_o else np.ones(close.shape, bool)
    rmse_equal = float(np.sqrt((diff[close] ** 2).mean())) if close.any() else 0.0
    return {"equal": float(close.mean()), "counts": float(same_counts.mean()), "rmse": float(np.sqrt((diff ** 2).mean())),
            "rmse_equal_pixels": rmse_equal, "worst": float(np.abs(diff).max()), "n": int(close.size), "n_differ": int((~close).sum())}


def assert_image_parity(out_g, out_o, what="", res=None, cfg=None, noise_share=RMSE_NOISE_SHARE):
    """The gates above.  `res`, `cfg`: the resident scene and the configuration `out_g` was rendered with;
    needed only when some pixel differs, to measure the image's noise with a second seed."""
    p = image_parity(out_g, out_o)
    msg = f"{what}: {p}"
    assert p["equal"] >= PIXEL_GATE, msg
    assert p["counts"] >= COUNT_GATE, msg
    assert p["rmse_equal_pixels"] <= RMSE_EQUAL_GATE, msg
    if p["rmse"] > RMSE_EQUAL_GATE:
        assert res is not None and cfg is not None, "pixels differ and no scene was given to measure the noise: " + msg
        other = cfg.copy()
        other.seed = cfg.seed ^ 0x5EED5EED
        out_b, _ = res.render(other)
        lg, lb = out_g["linear"], out_b["linear"]
        finite = np.isfinite(lg) & np.isfinite(lb)
        noise = float(np.sqrt((np.where(finite, lg - lb, 0.0) ** 2).mean()))
        p["noise_rmse"] = noise
        assert p["rmse"] <= noise_share * noise, f"{what}: {p}"
    return p


def noise_gate(gpu_a, gpu_b, ref, factor):
    """The statistical gate against the reference's own random-number mode (one xoshiro256++ per row: a
    different sample sequence, so only statistics can agree).  RMSE(gpu, ref) must not exceed `factor` x
    RMSE(gpu seed A, gpu seed B): two independent estimates at the same sample counts differ by pure
    Monte-Carlo noise, and an unbiased device image differs from the reference's by no more."""
    rmse_ref = float(np.sqrt(((gpu_a - ref) ** 2).mean()))
    rmse_self = float(np.sqrt(((gpu_a - gpu_b) ** 2).mean()))
    assert rmse_ref <= factor * rmse_self, (rmse_ref, rmse_self, factor)
    return rmse_ref, rmse_self
