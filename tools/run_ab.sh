set -e
for v in "" old "" old; do
  CLAVA_LIB_VARIANT=$v timeout -k 10 300 python3 tools/bench_chain.py 2>&1 | tail -1 >> gpurun_out/ab_chain.log
done
cat gpurun_out/ab_chain.log
