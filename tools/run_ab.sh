set -e
for m in "" "15" "16" "17" "18" "" "15,128,256" "17,128,256" "16,256,128" "18,256,128"; do
  echo "small=$m" >> gpurun_out/ab_s.log
  CLAVA_F16X3_SMALL=$m timeout -k 10 200 python3 tools/bench_chain.py 200 2 >> gpurun_out/ab_s.log 2>&1
done
