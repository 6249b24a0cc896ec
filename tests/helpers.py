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
