"""bench.py derives its counter-based roofline figures from profiles/roofline_inputs.json, a committed file of counter sums.
The file names the build of the kernels it was taken on; on any other build the figures are refused, not replayed."""
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from raytracer_amd import abi  # noqa: E402


def test_counter_sums_of_another_build_are_refused(tmp_path, monkeypatch):
    src = tmp_path / "raytracer_amd"
    shutil.copytree(os.path.join(ROOT, "raytracer_amd", "csrc"), src / "csrc", ignore=shutil.ignore_patterns("host"))
    monkeypatch.setattr(abi, "_HERE", str(src))
    before = abi.kernel_source_hash()
    inputs = tmp_path / "roofline_inputs.json"
    inputs.write_text(json.dumps({"source_hash": before, "rays": 1.0, "kernels": {}}))
    got, why = bench.load_roofline_inputs(str(inputs), abi.kernel_source_hash())
    assert got is not None and why is None
    with open(src / "csrc" / "rt_device.h", "a") as f:  # touch one of the kernel sources
        f.write("\n// a change\n")
    after = abi.kernel_source_hash()
    assert after != before
    got, why = bench.load_roofline_inputs(str(inputs), after)
    assert got is None and "another build" in why
    got, why = bench.load_roofline_inputs(str(tmp_path / "nothing.json"), after)
    assert got is None and "missing" in why


def test_counters_follow_this_runs_kernel_time_within_bounds():
    c = {"FETCH_SIZE": 1.0, "WRITE_SIZE": 1.0, "TCC_HIT_sum": 6.0, "TCC_MISS_sum": 4.0, "SQ_ACTIVE_INST_VALU": 0.5 * 1024 * 1e9 / 4,
         "GRBM_GUI_ACTIVE": 8e9, "SQ_WAIT_ANY": 1.0, "SQ_WAVE_CYCLES": 2.0, "SQ_THREAD_CYCLES_VALU": 0.5 * 64 * 0.5 * 1024 * 1e9 / 4,
         "kernel_avg_launch_ms": 10.0}
    inputs = {"rays": 100.0, "kernels": {"k_extend": c}}
    same = bench.counter_roofline(inputs, 100.0, 560.0, 0.0, 0.5, extend_launches=56)["k_extend"]
    assert abs(same["valu_busy"] - 0.5) < 1e-12 and abs(same["valu_issue_frac"] - 0.25) < 1e-12
    slower = bench.counter_roofline(inputs, 100.0, 560.0 * 1.05, 0.0, 0.5, extend_launches=56)["k_extend"]
    assert abs(slower["valu_busy"] - 0.5 / 1.05) < 1e-12  # the same instructions over 5 % more time
    off = bench.counter_roofline(inputs, 100.0, 560.0 * 1.3, 0.0, 0.5, extend_launches=56)
    assert "k_extend" not in off and "more than" in off["refused"]


def test_committed_counter_sums_name_their_build():
    with open(os.path.join(ROOT, "profiles", "roofline_inputs.json")) as f:
        inputs = json.load(f)
    assert "source_hash" in inputs and len(inputs["source_hash"]) == 16
    if inputs["source_hash"] != abi.kernel_source_hash():
        # not a failure of the code: bench.py then reports frac = null with the reason, which is the point of the hash
        import warnings
        warnings.warn("profiles/roofline_inputs.json was taken on another build of the kernels: run tools/profile_round.sh + tools/summarize_profiles.py")
