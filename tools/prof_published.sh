#!/bin/bash
# rocprofv3 kernel trace of the published Transformer variant (cfg3 YAML, B=128) -> gpurun_out/prof_published_summary.txt
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/prof4
rocprofv3 --kernel-trace --stats -d gpurun_out/prof4 -o bench -- python bench.py --model-name PointCAE_transformer_fc_global_folding_local --steps 10 --warmup 3 --no-also --no-cpu-baseline --no-calibration --no-tvis-table --probe-steps 0 > gpurun_out/prof_published.log 2>&1
grep '"metric"' gpurun_out/prof_published.log | cut -c1-220
python tools/prof_db.py gpurun_out/prof4/bench_results.db 10 50 > gpurun_out/prof_published_summary.txt
python tools/prof_db.py gpurun_out/prof4/bench_results.db 10 0 SEQ > gpurun_out/prof_published_seq.txt
rm -rf gpurun_out/prof4
