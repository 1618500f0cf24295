set -e
timeout -k 10 900 python -u -m pytest -x -q --timeout 300 --timeout-method thread tests/test_gpu_parity.py tests/test_gpu_configs.py tests/test_gpu_narrow.py tests/test_gpu_pipeline.py > gpurun_out/t_par.log 2>&1 || { tail -40 gpurun_out/t_par.log; exit 1; }
tail -3 gpurun_out/t_par.log
