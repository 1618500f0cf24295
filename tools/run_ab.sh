set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/clf_tl
mkdir -p $O
timeout -k 10 300 rocprofv3 --kernel-trace -d $O -o p --output-format csv -- python3 tools/bench_classifier_train.py > $O.log 2>&1
T=$(ls $O/*/p_kernel_trace.csv $O/p_kernel_trace.csv 2>/dev/null | head -1)
python3 tools/trace_timeline.py $T 170 adamw_ema > gpurun_out/clf_timeline.txt
rm -rf $O
cat gpurun_out/clf_timeline.txt
