# The whole benchmark image (C3, 1200x1200, adaptive, ~1889 spp) against the CPU oracle with the same per-path
# random streams: every row.  About four minutes of 16 host threads.  bash tools/full_image_parity.sh
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
( while sleep 60; do date >> $O/heartbeat; done ) &
HB=$!
trap "kill $HB" EXIT
timeout -k 10 1000 python3 -u $R/tools/config_table.py --configs C3 --cpu-rays 1e11 --cpu-seconds 100000 --out $O/full_image_c3.json > $O/full_image_c3.log 2>&1
