#!/bin/bash
# The round's artefacts from ONE box; boxes differ by +-3 %: a short default bench
# first; above $1 ms per step the call ends there.   gpurun --timeout 2400 -- bash tools/lab/final_artifacts.sh 9.85
cd "$GRAFT_REPO_ROOT"
LIMIT=${1:-9.60}
ms=$(python bench.py --no-cpu-baseline --no-also --no-tvis-table --probe-steps 0 --steps 40 --warmup 10 2>/dev/null | grep '"metric"' | tail -1 | python -c 'import sys,json; print(json.loads(sys.stdin.read())["ms_per_step"])')
echo "probe ms/step $ms (limit $LIMIT)"
python -c "import sys; sys.exit(0 if float('$ms') < float('$LIMIT') else 1)" || { echo "slow box: skipped"; exit 0; }
bash tools/prof_round.sh > gpurun_out/round.log 2>&1
bash tools/prof_cfg2.sh > /dev/null 2>&1
bash tools/prof_dgcnn.sh > /dev/null 2>&1
python bench.py > gpurun_out/bench_final.json 2> gpurun_out/bench_final.err
echo "done: $(head -2 gpurun_out/round/kernel_summary.txt | tail -1)"
python - <<'PY'
import json
d = json.loads(open('gpurun_out/bench_final.json').read().strip().splitlines()[-1])
a = d['also']
print(d['ms_per_step'], d['value'], 'cfg2', a['cfg2']['ms_per_step'], 'pub', a['published_variant']['ms_per_step'], 'dgcnn', a['dgcnn']['ms_per_step'], 'cfg5', a['cfg5_shape']['ms_per_step'])
PY
