# Scratch runner for A/B experiments on the GPU box (gpurun -- 'bash tools/run_ab.sh'): edit freely.
set -e
timeout -k 10 600 python -u -m pytest -x -q --timeout 200 --timeout-method thread tests/test_gpu_configs.py tests/test_gpu_f16x3.py tests/test_gpu_parity.py > gpurun_out/t.log 2>&1 || { tail -30 gpurun_out/t.log; exit 1; }
tail -2 gpurun_out/t.log
for v in prev new prev new; do
  if [ $v = new ]; then unset CLAVA_LIB_VARIANT; else export CLAVA_LIB_VARIANT=$v; fi
  echo "== $v"; timeout -k 10 150 python3 tools/bench_chain.py 300 3
done
