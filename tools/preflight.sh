#!/bin/bash
# What the driver runs at round end, in one call on the GPU box:  gpurun --timeout 1500 -- bash tools/preflight.sh
#   smoke() on cuda:0, the -m gpu suite, the default bench line (-> gpurun_out/preflight_*.{txt,json})
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 1200 python -m pytest tests -m gpu -x -q > gpurun_out/preflight_tests.txt 2>&1; echo "pytest rc=$? $(tail -1 gpurun_out/preflight_tests.txt)"
python bench.py > gpurun_out/preflight_bench.json 2> gpurun_out/preflight_bench.err; echo "bench rc=$?"
python - <<'PY'
import json
d = json.loads(open('gpurun_out/preflight_bench.json').read().strip().splitlines()[-1])
a = d['also']
print('ms/step %.3f  clouds/s %.0f  roofline %.3f (%.1f TFLOP/s-eq, %.2f ms of families)  best %.3f  whole-step %.1f TFLOP/s  cfg2 %.2f ms  published %.2f ms  '
      'dgcnn %.2f ms  cfg5-shape %.2f ms  cpu %.1f clouds/s' % (
          d['ms_per_step'], d['value'], d['roofline']['frac'], d['roofline']['achieved'], d['roofline']['ms_per_step'],
          d['roofline_best']['frac'], d['whole_step'].get('achieved', 0.0), a['cfg2']['ms_per_step'],
          a['published_variant']['ms_per_step'], a['dgcnn']['ms_per_step'], a['cfg5_shape']['ms_per_step'], d['cpu_baseline']['value']))
PY
