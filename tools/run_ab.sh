set -e
for args in "1024 1024 0" "512 1024 4" "256 512 1" "128 256 -1"; do
  echo "== N K cfg: $args" >> gpurun_out/stamps.log
  CLAVA_LIB_VARIANT=stamps timeout -k 10 120 python3 tools/dbg_gemm_stamps.py $args >> gpurun_out/stamps.log 2>&1 || true
done
cat gpurun_out/stamps.log
