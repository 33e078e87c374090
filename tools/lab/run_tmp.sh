cd "$GRAFT_REPO_ROOT"
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/full_tests.txt 2>&1; echo "pytest rc=$? $(tail -1 gpurun_out/full_tests.txt)"
