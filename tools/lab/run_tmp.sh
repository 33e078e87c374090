cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 900 python -m pytest tests/test_gpu_rows3.py tests/test_gpu_rows.py tests/test_gpu_gemm.py -x -q 2>&1 | tail -3
ROUNDS=3 bash tools/lab/abn.sh PDAE_LIB tools/lab/lab_prebkn.so point_dae_amd/libpdae_hip.so
for v in prebkn cur; do
  lib=tools/lab/lab_$v.so; [ $v = cur ] && lib=point_dae_amd/libpdae_hip.so
  OUT=gpurun_out/tr_$v; rm -rf $OUT; mkdir -p $OUT
  PDAE_LIB=$lib rocprofv3 --kernel-trace --output-format csv -d $OUT/stats -o bench -- python bench.py --no-cpu-baseline --no-also --no-tvis-table --no-calibration --probe-steps 0 > $OUT/log.txt 2>&1
  t=$(find $OUT/stats -name '*kernel_trace.csv' | head -1)
  python tools/trace_summary.py "$t" 50 > gpurun_out/ks_$v.txt
  rm -rf $OUT/stats
  head -3 gpurun_out/ks_$v.txt | tail -2
done
