# k_extend scheduling knobs on the benchmark workload: one bench line per setting.  bash tools/sweep_extend.sh
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/sweep.log
: > $O
run() {
  echo "== $*" >> $O
  env "$@" timeout -k 10 200 python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | python3 $R/tools/show_bench.py /dev/stdin | head -1 >> $O || exit 1
}
run RT_HIP_REFILL=40
run RT_HIP_REFILL=32
run RT_HIP_REFILL=48
run RT_HIP_CLAIM=256
run RT_HIP_CLAIM=1024
run RT_HIP_NODE_MIN=8
run RT_HIP_NODE_MIN=24
run RT_HIP_MAX_GROUP=64
run RT_HIP_LEAFQ=3
