# k_extend scheduling knobs on the benchmark workload: one bench line per setting.
#   bash tools/sweep_extend.sh VAR=value [VAR=value ...]     (each argument is one run)
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/sweep.log
: > $O
for setting in "$@"; do
  echo "== $setting" >> $O
  env $setting timeout -k 10 200 python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | python3 $R/tools/show_bench.py /dev/stdin 2>/dev/null | sed -n 1p >> $O || exit 1
done
