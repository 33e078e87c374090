#!/bin/bash
# rocprofv3 kernel trace of the published Transformer variant's step -> gpurun_out/prof_published_summary.txt
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/prof3
rocprofv3 --kernel-trace --stats -d gpurun_out/prof3 -o bench -- python bench.py --model-name PointCAE_transformer_fc_global_folding_local --no-cpu-baseline --no-also --steps 10 --warmup 3 --no-tvis-table --probe-steps 0 > gpurun_out/prof_published.log 2>&1
grep '"metric"' gpurun_out/prof_published.log | cut -c1-220
python tools/prof_db.py gpurun_out/prof3/bench_results.db 10 50 > gpurun_out/prof_published_summary.txt
rm -rf gpurun_out/prof3
