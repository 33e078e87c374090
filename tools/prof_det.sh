#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/profd
export PDAE_DETERMINISTIC=1
rocprofv3 --kernel-trace --stats -d gpurun_out/profd -o bench -- python bench.py --no-cpu-baseline --no-also --steps 20 --warmup 5 --no-tvis-table --no-calibration --probe-steps 0 > gpurun_out/profd.log 2>&1
grep '"metric"' gpurun_out/profd.log | cut -c1-200
python tools/prof_db.py gpurun_out/profd/bench_results.db 20 45 > gpurun_out/profd_summary.txt
rm -rf gpurun_out/profd
