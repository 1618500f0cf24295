#!/bin/bash
# Round-3 rocprofv3 artefacts (run on the GPU box from the repository root): kernel-trace summaries and PMC passes, each
# under its own time limit; stops at the first failure.  Outputs go to gpurun_out/r03/ and are copied into profiles/ by hand.
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r03
mkdir -p $O
stats() {   # name, command...
  local name=$1; shift
  timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/$name -o p --output-format csv -- "$@" > $O/$name.log 2>&1
  cp $O/$name/p_kernel_stats.csv $O/${name}_kernel_stats.csv
  rm -rf $O/$name            # the full kernel trace is tens of MB; only the summary is kept
}
pmc() {     # name, counter, command...
  local name=$1 ctr=$2; shift 2
  timeout -s KILL 120 rocprofv3 --pmc $ctr -d $O/${name}_$ctr -o p --output-format csv -- "$@" > $O/${name}_$ctr.log 2>&1
}
B="python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --train-steps 0 --no-f32-leg --no-gmm --clf-train-steps 0 --no-scale-sweep --no-t2000"
stats r03_bench_f16x3 $B
CLAVA_CHAIN_STREAMS=1 stats r03_bench_f16x3_serial $B
stats r03_step_kernels python3 tools/bench_step.py --profile
MODES=f16x3 stats r03_train_step python3 tools/bench_train.py
CLAVA_GEMM=f16x3 stats r03_scale_sweep python3 tools/bench_scale_sweep.py
stats r03_classifier_train python3 tools/bench_classifier_train.py
pmc gemm FETCH_SIZE python3 tools/pmc_f16x3.py 1024 1024 0
pmc gemm WRITE_SIZE python3 tools/pmc_f16x3.py 1024 1024 0
pmc step FETCH_SIZE python3 tools/bench_step.py --profile
pmc step WRITE_SIZE python3 tools/bench_step.py --profile
python3 tools/pmc_traffic.py $O/gemm_FETCH_SIZE/p_counter_collection.csv $O/gemm_WRITE_SIZE/p_counter_collection.csv gemm_f16x3_kernel $O/traffic_r03_f16x3.json
python3 tools/pmc_summary.py multinomial_step_kernel $O/step_multinomial_pmc.json $O/step_FETCH_SIZE/p_counter_collection.csv $O/step_WRITE_SIZE/p_counter_collection.csv
python3 tools/pmc_summary.py gauss_step_kernel $O/step_gauss_pmc.json $O/step_FETCH_SIZE/p_counter_collection.csv $O/step_WRITE_SIZE/p_counter_collection.csv
python3 tools/pmc_summary.py cat_input_kernel $O/cat_input_pmc.json $O/step_FETCH_SIZE/p_counter_collection.csv $O/step_WRITE_SIZE/p_counter_collection.csv || true
rm -rf $O/gemm_FETCH_SIZE $O/gemm_WRITE_SIZE $O/step_FETCH_SIZE $O/step_WRITE_SIZE
echo done
