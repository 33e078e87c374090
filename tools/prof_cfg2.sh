#!/bin/bash
# rocprofv3 kernel trace of the cfg2 bench -> gpurun_out/prof_cfg2_summary.txt
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/prof2
rocprofv3 --kernel-trace --stats -d gpurun_out/prof2 -o bench -- python bench.py --workload cfg2 --no-calibration --steps 10 --warmup 3 > gpurun_out/prof_cfg2.log 2>&1
grep '"metric"' gpurun_out/prof_cfg2.log | cut -c1-220
python tools/prof_db.py gpurun_out/prof2/bench_results.db 10 50 > gpurun_out/prof_cfg2_summary.txt
python tools/prof_db.py gpurun_out/prof2/bench_results.db 10 0 SEQ > gpurun_out/prof_cfg2_seq.txt
rm -rf gpurun_out/prof2
