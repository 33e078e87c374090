#!/bin/bash
# rocprofv3 kernel trace of the default bench (30 timed steps) -> gpurun_out/prof/bench_results.db
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/prof
rocprofv3 --kernel-trace --stats -d gpurun_out/prof -o bench -- python bench.py --no-cpu-baseline --no-also --steps 30 --warmup 5 --no-tvis-table --probe-steps 0 "$@" > gpurun_out/prof_bench.log 2>&1
grep '"metric"' gpurun_out/prof_bench.log | cut -c1-220
python tools/prof_db.py gpurun_out/prof/bench_results.db 30 60 > gpurun_out/prof_summary.txt
python tools/prof_db.py gpurun_out/prof/bench_results.db 30 40 pdae::gemm > gpurun_out/prof_gemm.txt
python tools/prof_db.py gpurun_out/prof/bench_results.db 30 80 pdae::rows > gpurun_out/prof_rows.txt
python tools/prof_db.py gpurun_out/prof/bench_results.db 30 0 SEQ > gpurun_out/prof_seq.txt
rm -f gpurun_out/prof/bench_results.db
