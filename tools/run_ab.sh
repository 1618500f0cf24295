set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r03
mkdir -p $O
stats() {
  local name=$1; shift
  timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/$name -o p --output-format csv -- "$@" > $O/$name.log 2>&1
  cp $O/$name/p_kernel_stats.csv $O/${name}_kernel_stats.csv
  rm -rf $O/$name
}
pmc() {
  local name=$1 ctr=$2; shift 2
  timeout -s KILL 120 rocprofv3 --pmc $ctr -d $O/${name}_$ctr -o p --output-format csv -- "$@" > $O/${name}_$ctr.log 2>&1
}
stats r03_step_kernels python3 tools/bench_step.py --profile
CLAVA_GEMM=f16x3 stats r03_scale_sweep python3 tools/bench_scale_sweep.py
pmc step FETCH_SIZE python3 tools/bench_step.py --profile
pmc step WRITE_SIZE python3 tools/bench_step.py --profile
python3 tools/pmc_summary.py multinomial_step_kernel $O/step_multinomial_pmc.json $O/step_FETCH_SIZE/p_counter_collection.csv $O/step_WRITE_SIZE/p_counter_collection.csv
export STEP_ONLY=multinomial_step_indices_only
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU -d $O/sq -o p --output-format csv -- python3 tools/bench_step.py --profile > $O/sq.log 2>&1
python3 tools/pmc_kernel.py $(ls $O/sq/*/p_counter_collection.csv $O/sq/p_counter_collection.csv 2>/dev/null | head -1) "multinomial_step_kernel<0, true>" > $O/icdf_sq.txt
rm -rf $O/step_FETCH_SIZE $O/step_WRITE_SIZE $O/sq
cat $O/icdf_sq.txt; head -3 $O/r03_step_kernels_kernel_stats.csv
