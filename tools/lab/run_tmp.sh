cd "$GRAFT_REPO_ROOT"
timeout 600 python -m pytest tests/test_gpu_gemm.py tests/test_gpu_sa.py -x -q 2>&1 | tail -2
ROUNDS=3 bash tools/lab/abn.sh PDAE_BN_FUSED 0 1
for v in 0 1 0 1; do
  line=$(PDAE_BN_FUSED=$v python bench.py --workload cfg2 --no-cpu-baseline --no-also --no-tvis-table --no-calibration --probe-steps 0 --steps 20 --warmup 5 2>/dev/null | grep '"metric"' | tail -1)
  echo "cfg2 PDAE_BN_FUSED=$v $(echo "$line" | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["ms_per_step"])' 2>/dev/null)"
done
PDAE_GEMM=f32mfma timeout 1500 python -m pytest tests -m gpu -q > gpurun_out/full_tests_f32.txt 2>&1; echo "f32mfma pytest rc=$? $(tail -1 gpurun_out/full_tests_f32.txt)"
