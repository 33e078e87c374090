#!/bin/bash
# rocprofv3 kernel trace of the DGCNN auto-encoder's step (B=32 per GPU, N=1024) -> gpurun_out/prof_dgcnn_summary.txt
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/prof3
rocprofv3 --kernel-trace --stats -d gpurun_out/prof3 -o bench -- python bench.py --workload cfg2 --model-name Point_CAE_DGCNN_FCOnly --batch 32 --no-calibration --steps 10 --warmup 3 > gpurun_out/prof_dgcnn.log 2>&1
grep '"metric"' gpurun_out/prof_dgcnn.log | cut -c1-220
python tools/prof_db.py gpurun_out/prof3/bench_results.db 10 50 > gpurun_out/prof_dgcnn_summary.txt
python tools/prof_db.py gpurun_out/prof3/bench_results.db 10 0 SEQ > gpurun_out/prof_dgcnn_seq.txt
rm -rf gpurun_out/prof3
