set -e
timeout -k 10 400 python -u -m pytest tests/test_gpu_device_build.py -x -q --timeout 120 --timeout-method thread 2>&1 | tail -15
run() { echo "== $*"; env "$@" RT_HIP_VERBOSE=1 timeout -k 10 200 python3 tools/device_build_timing.py --no-count --only-device 2>gpurun_out/_err.log | python3 -c "import json,sys; d=json.load(sys.stdin)['device_tree']; print('extend %.1f shade %.1f build %.4f' % (d['extend_ms'], d['shade_ms'], d['copy_and_build_s']))"; grep "rt_hip build" gpurun_out/_err.log; }
run RT_HIP_BUILD=2
run RT_HIP_BUILD=3
run RT_HIP_BUILD=0
