set -e
timeout -k 10 600 python -u -m pytest tests/test_gpu_device_build.py tests/test_gpu_configs.py tests/test_zz_no_alloc.py -x -q --timeout 300 --timeout-method thread 2>&1 | tail -8
RT_HIP_VERBOSE=1 timeout -k 10 200 python3 tools/build_only.py --repeat 3 2>&1 | grep -E "rt_hip build|^build"
RT_HIP_VERBOSE=1 timeout -k 10 200 python3 tools/build_only.py --repeat 2 --triangles 100000 2>&1 | grep -E "rt_hip build|^build"
timeout -k 10 300 python3 tools/device_build_timing.py --no-count 2>/dev/null | python3 -c "import json,sys; d=json.load(sys.stdin); print('host extend %.1f render %.3f prep %.3f+%.3f; device extend %.1f render %.3f prep %.3f' % (d['host_tree']['extend_ms'], d['host_tree']['render_s'], d['host_tree']['load_expand_build_s'], d['host_tree']['upload_s'], d['device_tree']['extend_ms'], d['device_tree']['render_s'], d['device_tree']['copy_and_build_s']))"
