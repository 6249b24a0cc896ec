# Counter passes over the 10 M-triangle soup (the scene that does not fit the caches): kernel times, fabric
# traffic and occupancy of k_extend.  Run on the GPU box: bash tools/profile_soup.sh [n_triangles]
set -e
R=$GRAFT_REPO_ROOT
N=${1:-10000000}
O=$R/gpurun_out/soup
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
( while sleep 60; do date >> $O/heartbeat; done ) &
HB=$!
trap "kill $HB" EXIT
timeout -k 10 600 python3 -u $R/tools/soup_run.py $N 960 540 --no-oracle > $O/plain.log
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $R/tools/soup_run.py $N 960 540 --no-oracle > $O/stats.log
timeout -s KILL 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch -- python3 $R/tools/soup_run.py $N 960 540 --no-oracle > $O/fetch.log
timeout -s KILL 300 rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $O/write -- python3 $R/tools/soup_run.py $N 960 540 --no-oracle > $O/write.log
timeout -s KILL 300 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY --output-format csv -d $O/sq -- python3 $R/tools/soup_run.py $N 960 540 --no-oracle > $O/sq.log
timeout -s KILL 300 rocprofv3 --pmc GRBM_GUI_ACTIVE GRBM_COUNT --output-format csv -d $O/grbm -- python3 $R/tools/soup_run.py $N 960 540 --no-oracle > $O/grbm.log
