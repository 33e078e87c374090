cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_rows.py tests/test_gpu_block.py tests/test_gpu_model.py -x -q 2>&1 | tail -3
python tools/lab_rows.py 2>&1 | tail -5
for i in 1 2; do python bench.py --no-cpu-baseline --no-also --no-tvis-table --probe-steps 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bench ms/step', d['ms_per_step'])"; done
