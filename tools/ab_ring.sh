#!/bin/bash
# A/B of the staging-ring depth on one box.
for r in 4 16 2 4; do
  PDAE_RING=$r python bench.py --no-cpu-baseline --steps 100 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('ring', $r, d['value'], d['ms_per_step'])"
done
