cd "$GRAFT_REPO_ROOT"
timeout 1200 python -m pytest tests/test_gpu_rows3.py tests/test_gpu_rows.py tests/test_gpu_gemm.py -x -q 2>&1 | tail -8
python tools/lab/wgrad3_lab.py 2>&1 | tail -12
PDAE_WGRAD3=b python tools/lab/wgrad3_lab.py 2>&1 | tail -12
for v in b t b t; do
  line=$(PDAE_WGRAD3=$v python bench.py --no-cpu-baseline --no-also --no-tvis-table --no-calibration --probe-steps 0 --steps 40 --warmup 10 2>/dev/null | grep '"metric"' | tail -1)
  echo "PDAE_WGRAD3=$v $(echo "$line" | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["ms_per_step"])' 2>/dev/null)"
done
for v in tools/lab/lab_plain.so point_dae_amd/libpdae_hip.so tools/lab/lab_plain.so point_dae_amd/libpdae_hip.so; do
  line=$(PDAE_WGRAD3=b PDAE_LIB=$v python bench.py --no-cpu-baseline --no-also --no-tvis-table --no-calibration --probe-steps 0 --steps 40 --warmup 10 2>/dev/null | grep '"metric"' | tail -1)
  echo "PDAE_LIB=$v (wgrad b) $(echo "$line" | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["ms_per_step"])' 2>/dev/null)"
done
