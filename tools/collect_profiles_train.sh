#!/bin/bash
# Round-2 refresh of the TRAINING artefacts only (run on the GPU box from the repository root): kernel-trace summaries of
# the denoiser training step and the scale-sweep step, one step's kernel timeline of each, and the L2 / traffic counters of
# the batch-major weight-gradient kernel.  Outputs go to gpurun_out/r02t/ and are copied into profiles/ by hand.
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r02t
mkdir -p $O
stats() {   # name, command...
  local name=$1; shift
  timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/$name -o p --output-format csv -- "$@" > $O/$name.log 2>&1
  cp $O/$name/p_kernel_stats.csv $O/${name}_kernel_stats.csv
  python3 tools/trace_timeline.py $O/$name/p_kernel_trace.csv ${TL_BACK:-3} stage_fetch > $O/${name}_timeline.txt
  rm -rf $O/$name            # the full kernel trace is tens of MB; only the summary and one step's timeline are kept
}
pmc() {     # name, counters..., then -- command
  local name=$1; shift
  local ctrs=()
  while [ "$1" != "--" ]; do ctrs+=("$1"); shift; done
  shift
  timeout -s KILL 120 rocprofv3 --pmc "${ctrs[@]}" -d $O/$name -o p --output-format csv -- "$@" > $O/$name.log 2>&1
}
MODES=f16x3 stats r02_train_step python3 tools/bench_train.py
TL_BACK=8 CLAVA_GEMM=f16x3 stats r02_scale_sweep python3 tools/bench_scale_sweep.py
NSHAPES=1 pmc tn_l2 TCC_HIT_sum TCC_MISS_sum -- python3 tools/bench_wgrad_tn.py
NSHAPES=1 pmc tn_fetch FETCH_SIZE -- python3 tools/bench_wgrad_tn.py
NSHAPES=1 pmc tn_write WRITE_SIZE -- python3 tools/bench_wgrad_tn.py
for d in tn_l2 tn_fetch tn_write; do python3 tools/pmc_kernel.py $O/$d/p_counter_collection.csv gemm_f16x3_tn_kernel; done > $O/r02_gemm_f16x3_tn_pmc.txt
rm -rf $O/tn_l2 $O/tn_fetch $O/tn_write
echo done
