set -e
timeout -k 10 600 python -u -m pytest -x -q --timeout 300 --timeout-method thread tests/test_gpu_classifier_training.py tests/test_gpu_pipeline.py > gpurun_out/t_clf.log 2>&1 || { tail -40 gpurun_out/t_clf.log; exit 1; }
tail -2 gpurun_out/t_clf.log
for n in 1 2 3 1 2 3; do
  CLAVA_CLF_WGRAD_STREAMS=$n timeout -k 10 200 python3 tools/bench_classifier_train.py 2>&1 | grep -E "graph=1" | sed "s/^/wgrad_streams=$n /"
done
