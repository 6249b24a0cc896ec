# One round's profile set of the headline workload, on the GPU box (run through gpurun):
#   tools/profile_round.sh <out-subdir-of-gpurun_out>
# 1. the plain benchmark line, 2. rocprofv3 kernel trace + stats, 3. one --pmc pass per counter group of
# `bench.py --profile-pass` (exactly one render), 4. the FETCH_SIZE calibration.  tools/summarize_profiles.py
# then condenses the directories into profiles/.
set -e
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$1
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 python3 $R/bench.py > $O/bench.json 2> $O/bench.err
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-reuse-leg > $O/stats.json 2> $O/stats.err
pass() {  # name, counters...
  n=$1; shift
  timeout -s KILL 200 rocprofv3 --pmc "$@" --output-format csv -d $O/pmc_$n -- python3 $R/bench.py --profile-pass > $O/pmc_$n.json 2> $O/pmc_$n.err
}
pass fetch FETCH_SIZE
pass write WRITE_SIZE TCC_HIT_sum TCC_MISS_sum
pass sq SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY
pass sq2 SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS
pass sq3 SQ_THREAD_CYCLES_VALU
pass grbm GRBM_GUI_ACTIVE GRBM_COUNT
if [ -x $R/tools/calib/fetch_calibration ]; then
  timeout -k 10 120 $R/tools/calib/fetch_calibration > $O/calib_plain.jsonl
  timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/calib_fetch -- $R/tools/calib/fetch_calibration > $O/calib_fetch.jsonl
  timeout -s KILL 120 rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_REQ_sum TCC_MISS_sum --output-format csv -d $O/calib_req -- $R/tools/calib/fetch_calibration > $O/calib_req.jsonl || true
fi
echo profile set written to $O
