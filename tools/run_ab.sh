set -e
for s in 128 64 96 128 64; do
  echo "bn_slabs=$s" >> gpurun_out/ab_bn.log
  CLAVA_BN_SLABS=$s timeout -k 10 200 python3 tools/bench_classifier_train.py 2>&1 | grep -E "graph=1|batchnorm" >> gpurun_out/ab_bn.log
done
cat gpurun_out/ab_bn.log
