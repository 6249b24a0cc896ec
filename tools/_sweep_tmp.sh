set -e
timeout -k 10 300 python -u -m pytest tests/test_gpu_device_build.py -x -q --timeout 120 --timeout-method thread 2>&1 | tail -15
RT_HIP_VERBOSE=1 timeout -k 10 200 python3 tools/build_only.py --repeat 3 2>&1 | grep -E "rt_hip build|^build"
