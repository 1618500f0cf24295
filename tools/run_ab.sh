set -e
timeout -k 10 600 python -u -m pytest -x -q --timeout 300 --timeout-method thread tests/test_gpu_icdf.py > gpurun_out/t_icdf.log 2>&1 || true
tail -25 gpurun_out/t_icdf.log
for m in gumbel icdf; do
  echo "sampler=$m" >> gpurun_out/ab_icdf.log
  CLAVA_CAT_SAMPLER=$m timeout -k 10 200 python3 tools/bench_step.py >> gpurun_out/ab_icdf.log 2>&1
done
