set -e
for m in "" "0" "0,8,9" "8,9" "" "0" "0,8,9" "0,2,4,8,9"; do
  echo "mask=$m" >> gpurun_out/ab_cfg.log
  CLAVA_F16X3_DISABLE=$m timeout -k 10 200 python3 tools/bench_chain.py 200 3 >> gpurun_out/ab_cfg.log 2>&1
done
