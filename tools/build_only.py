#!/usr/bin/env python3
"""Builds the tree of a generated mesh on the device a few times (for rocprofv3 --kernel-trace --stats of the build's kernels).

  python3 tools/build_only.py [--triangles 2000000] [--repeat 3]
"""
import argparse
import os
import sys
import tempfile
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--triangles", type=int, default=2_000_000)
    ap.add_argument("--repeat", type=int, default=3)
    args = ap.parse_args()
    from raytracer_amd import RenderIntegrator, scenes
    from raytracer_amd.integrator import ResidentScene
    d = os.path.join(tempfile.gettempdir(), f"rt_build_timing_{os.getuid()}")
    sp, cp = scenes.materialize_c3(d, args.triangles, 256, 256, batch=16, max_batches=4, min_batches=2)
    ri = RenderIntegrator.new_from_json(sp, cp).prepare()
    for k in range(args.repeat):
        t0 = time.perf_counter()
        dev = ResidentScene(ri.scene, 0, build_on_device=True)
        print(f"build {k}: {1e3 * (time.perf_counter() - t0):.1f} ms", flush=True)
        dev.close()


if __name__ == "__main__":
    main()
