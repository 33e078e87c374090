#!/bin/bash
# The round's profile artefacts (run on the GPU box through gpurun):
#   1. rocprofv3 --kernel-trace --stats of the default bench command  -> gpurun_out/round/stats_*
#   2. separate --pmc FETCH_SIZE and --pmc WRITE_SIZE passes           -> gpurun_out/round/pmc.json
# Summaries are then copied into profiles/ by hand (gpurun_out/ is scratch).
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/round
rm -rf $OUT gpurun_out/prof; mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o bench -- python bench.py --no-cpu-baseline --no-also --no-tvis-table --no-calibration --probe-steps 0 > $OUT/stats_bench.log 2>&1
grep '"metric"' $OUT/stats_bench.log > $OUT/bench_line_under_rocprof.json
f=$(find $OUT/stats -name '*kernel_stats.csv' | head -1); cp "$f" $OUT/kernel_stats.csv
t=$(find $OUT/stats -name '*kernel_trace.csv' | head -1)
python tools/trace_summary.py "$t" 50 > $OUT/kernel_summary.txt
rm -rf $OUT/stats
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/pmc_$c -o pmc -- python bench.py --no-cpu-baseline --no-also --no-tvis-table --no-calibration --steps 5 --warmup 2 --probe-steps 0 > $OUT/pmc_$c.log 2>&1
done
python tools/pmc_summary.py $OUT/pmc_FETCH_SIZE $OUT/pmc_WRITE_SIZE $OUT/kernel_summary.txt > $OUT/pmc.json
rm -rf $OUT/pmc_FETCH_SIZE $OUT/pmc_WRITE_SIZE
ls -la $OUT
bash tools/prof_mfma.sh
