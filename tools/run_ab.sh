set -e
timeout -k 10 600 python -u -m pytest -x -q --timeout 300 --timeout-method thread tests/test_gpu_classifier_training.py tests/test_gpu_pipeline.py > gpurun_out/t_clf.log 2>&1
tail -5 gpurun_out/t_clf.log
for m in 1 0; do
  echo "bn_split=$m" >> gpurun_out/ab_bn.log
  CLAVA_BN_SPLIT=$m timeout -k 10 200 python3 tools/bench_classifier_train.py >> gpurun_out/ab_bn.log 2>&1
done
