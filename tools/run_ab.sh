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
B="python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --train-steps 0 --no-f32-leg --no-gmm --clf-train-steps 0 --no-scale-sweep --no-t2000"
stats r03_bench_f16x3 $B
CLAVA_CHAIN_STREAMS=1 stats r03_bench_f16x3_serial $B
stats r03_classifier_train python3 tools/bench_classifier_train.py
bash tools/collect_chain_timeline.sh chain_r03c > /dev/null 2>&1
head -30 gpurun_out/chain_r03c_timeline.txt
