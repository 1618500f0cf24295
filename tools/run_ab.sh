# Scratch runner for A/B experiments on the GPU box (gpurun -- 'bash tools/run_ab.sh'): edit freely.  As committed: the full
# GPU suite, the smoke check and the default bench, each under its own time limit.
set -e
timeout -k 10 1100 python -u -m pytest -x -q --timeout 300 --timeout-method thread tests -m gpu > gpurun_out/t_full.log 2>&1 || { tail -40 gpurun_out/t_full.log; exit 1; }
tail -3 gpurun_out/t_full.log
timeout -k 10 120 python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/smoke.log 2>&1; tail -1 gpurun_out/smoke.log
timeout -k 10 600 python3 bench.py > gpurun_out/bench_last.log 2> gpurun_out/bench_last.err
tail -c 300 gpurun_out/bench_last.log
