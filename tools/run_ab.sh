# Scratch runner for A/B experiments on the GPU box (gpurun -- 'bash tools/run_ab.sh'): edit freely.
set -e
timeout -k 10 300 python -u -m pytest -x -q --timeout 200 --timeout-method thread tests/test_gpu_configs.py -k "fused" 2>&1 | tail -3
echo "fused"; timeout -k 10 150 python3 tools/bench_chain.py 300 3
echo "per-layer"; CLAVA_FUSED=0 timeout -k 10 150 python3 tools/bench_chain.py 300 3
echo "fused"; timeout -k 10 150 python3 tools/bench_chain.py 300 3
