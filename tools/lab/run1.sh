cd $GRAFT_REPO_ROOT
bash tools/xproc_repro.sh 2>&1 | grep -v amdgpu
ITERS=150 bash tools/lab/nondet_bisect.sh 2>&1 | head -4
python bench.py --no-cpu-baseline --no-also --no-tvis-table --probe-steps 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bench ms/step', d['ms_per_step'])"
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -5
