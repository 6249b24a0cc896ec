for v in 3 0 1 2 3 0; do
  RT_HIP_TILE_HLOG=$v timeout -k 10 150 python bench.py --steps 3 --no-cpu-baseline --no-reuse-leg > gpurun_out/v_th$v.json 2> gpurun_out/v_th$v.err
  echo "tile_hlog=$v $(python tools/show_bench.py gpurun_out/v_th$v.json 2>/dev/null | head -1)"
done
