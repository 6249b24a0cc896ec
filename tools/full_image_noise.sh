# tools/full_image_noise.py with a heartbeat: the oracle is silent for four minutes.
R=$GRAFT_REPO_ROOT
( while sleep 60; do date >> $R/gpurun_out/heartbeat; done ) &
HB=$!
trap "kill $HB" EXIT
timeout -k 10 1000 python3 -u $R/tools/full_image_noise.py "$@" > $R/gpurun_out/full_image_noise.log 2>&1
