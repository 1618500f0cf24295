set -e
timeout -k 10 900 python -u -m pytest -x -q --timeout 300 --timeout-method thread tests/test_gpu_f16x3.py tests/test_gpu_configs.py > gpurun_out/t_par.log 2>&1 || { tail -40 gpurun_out/t_par.log; exit 1; }
tail -2 gpurun_out/t_par.log
CLAVA_GEMM=f16x3 timeout -k 10 300 python3 tools/bench_scale_sweep.py 2>&1 | tail -25
