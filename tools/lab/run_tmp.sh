cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 1700 python -m pytest tests/test_gpu_gemm.py tests/test_gpu_sa.py tests/test_gpu_model.py -x -q 2>&1 | tail -4
ROUNDS=3 bash tools/lab/abn.sh PDAE_BN_FUSED 0 1
for v in 0 1 0 1; do
  line=$(PDAE_BN_FUSED=$v python bench.py --workload cfg2 --no-cpu-baseline --no-also --no-tvis-table --no-calibration --probe-steps 0 --steps 20 --warmup 5 2>/dev/null | grep '"metric"' | tail -1)
  echo "cfg2 PDAE_BN_FUSED=$v $(echo "$line" | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["ms_per_step"])' 2>/dev/null)"
done
for v in 1; do
  OUT=gpurun_out/tr_bn$v; rm -rf $OUT; mkdir -p $OUT
  PDAE_BN_FUSED=$v rocprofv3 --kernel-trace --output-format csv -d $OUT/stats -o bench -- python bench.py --no-cpu-baseline --no-also --no-tvis-table --no-calibration --probe-steps 0 > $OUT/log.txt 2>&1
  t=$(find $OUT/stats -name '*kernel_trace.csv' | head -1)
  python tools/trace_summary.py "$t" 50 > gpurun_out/ks_bn$v.txt
  rm -rf $OUT/stats
  OUT=gpurun_out/tr_c2bn$v; rm -rf $OUT; mkdir -p $OUT
  PDAE_BN_FUSED=$v rocprofv3 --kernel-trace --output-format csv -d $OUT/stats -o bench -- python bench.py --workload cfg2 --no-cpu-baseline --no-also --no-tvis-table --no-calibration --probe-steps 0 --steps 10 --warmup 3 > $OUT/log.txt 2>&1
  t=$(find $OUT/stats -name '*kernel_trace.csv' | head -1)
  python tools/trace_summary.py "$t" 10 > gpurun_out/ks_c2bn$v.txt
  rm -rf $OUT/stats
done
grep "Li5E\|finish\|reduce_kernel" gpurun_out/ks_bn1.txt gpurun_out/ks_c2bn1.txt
