set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout -k 10 600 python -u -m pytest -x -q --timeout 300 --timeout-method thread tests/test_gpu_icdf.py tests/test_gpu_narrow.py tests/test_gpu_parity.py tests/test_gpu_configs.py > gpurun_out/t_icdf.log 2>&1 || { tail -40 gpurun_out/t_icdf.log; exit 1; }
tail -3 gpurun_out/t_icdf.log
timeout -k 10 200 python3 tools/bench_step.py > gpurun_out/ab_icdf.log 2>&1
O=gpurun_out/pmc_icdf
mkdir -p $O
export STEP_ONLY=multinomial_step_indices_only
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU -d $O/a -o p --output-format csv -- python3 tools/bench_step.py --profile > $O/a.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc SQ_BUSY_CYCLES SQ_INSTS_VALU_TRANS SQ_INST_CYCLES_VMEM SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_BRANCH SQ_ACTIVE_INST_SCA SQ_IFETCH -d $O/b -o p --output-format csv -- python3 tools/bench_step.py --profile > $O/b.log 2>&1 || true
python3 tools/pmc_kernel.py $(ls $O/a/*/p_counter_collection.csv $O/a/p_counter_collection.csv 2>/dev/null | head -1) "multinomial_step_kernel<0, true>" > gpurun_out/pmc_icdf.txt
python3 tools/pmc_kernel.py $(ls $O/b/*/p_counter_collection.csv $O/b/p_counter_collection.csv 2>/dev/null | head -1) "multinomial_step_kernel<0, true>" >> gpurun_out/pmc_icdf.txt || tail -5 $O/b.log
rm -rf $O
cat gpurun_out/pmc_icdf.txt
