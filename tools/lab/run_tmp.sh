cd "$GRAFT_REPO_ROOT"
timeout 900 python -m pytest tests/test_gpu_rows.py -k fold -x -q 2>&1 | tail -2
for v in e p e p; do
  line=$(PDAE_FOLD_INPUT=$v python bench.py --workload cfg2 --no-cpu-baseline --no-also --no-tvis-table --no-calibration --probe-steps 0 --steps 20 --warmup 5 2>/dev/null | grep '"metric"' | tail -1)
  echo "cfg2 PDAE_FOLD_INPUT=$v $(echo "$line" | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["ms_per_step"])' 2>/dev/null)"
done
