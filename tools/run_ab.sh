set -e
timeout -k 10 900 python -u -m pytest -x -q --timeout 300 --timeout-method thread tests/test_gpu_parity.py tests/test_gpu_configs.py tests/test_gpu_f16x3.py tests/test_gpu_narrow.py tests/test_gpu_pipeline.py > gpurun_out/t_par.log 2>&1 || { tail -40 gpurun_out/t_par.log; exit 1; }
tail -2 gpurun_out/t_par.log
for v in 1 0 1 0; do
  CLAVA_FUSE_GAUSS=$v timeout -k 10 300 python3 tools/bench_chain.py 2>&1 | tail -1 | sed "s/^/fuse_gauss=$v /" >> gpurun_out/ab_chain.log
done
cat gpurun_out/ab_chain.log
