#!/bin/bash
# Kernel timeline of one reverse step of the bench chain as timed (two streams): rocprofv3 kernel trace of a short bench,
# one step's kernels between two gauss_step launches.  Output: gpurun_out/<name>_timeline.txt (+ the stats summary).
set -e
NAME=${1:-chain}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/$NAME
mkdir -p $O
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $O -o p --output-format csv -- python3 bench.py --steps 1 --warmup 1 \
  --no-cpu-baseline --train-steps 0 --no-f32-leg --no-gmm --clf-train-steps 0 --no-scale-sweep --no-t2000 > $O.log 2>&1
T=$(ls $O/*/p_kernel_trace.csv $O/p_kernel_trace.csv 2>/dev/null | head -1)
S=$(ls $O/*/p_kernel_stats.csv $O/p_kernel_stats.csv 2>/dev/null | head -1)
# the step's last kernel: the guidance's fused back kernel with the Gaussian step inside (two streams; MARKER="narrow_split_kernel<2, true>"
# with CLAVA_FUSED=0), gauss_step_kernel (one stream)
M=${MARKER:-"cls_back_kernel"}
python3 tools/trace_timeline.py $T 700 "$M" > gpurun_out/${NAME}_timeline.txt
python3 tools/trace_timeline.py $T 701 "$M" >> gpurun_out/${NAME}_timeline.txt
cp $S gpurun_out/${NAME}_kernel_stats.csv
rm -rf $O
echo done
