"""raytracer_amd -- MI355X-native per-pixel Monte-Carlo rendering path behind
the operator surface of scott223/raytracer's RenderIntegrator.

  abi         ctypes mirror of include/rt_hip.h / include/rt_host.h
  integrator  RenderIntegrator (new_from_json / render / save_to_png), ResidentScene
  scenes      synthetic scene / config JSON in the reference's schema
  parallel    row sharding over GPUs and the gather to rank 0
"""
from .integrator import RenderIntegrator, ResidentScene  # noqa: F401
