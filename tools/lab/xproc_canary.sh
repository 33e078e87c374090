#!/bin/bash
cd "$GRAFT_REPO_ROOT"
echo "--- canary alone"; tools/lab/lab_xproc C 2 | head -8
for v in 0 15 17; do
  echo "--- LDS canary next to a bare process running gemm3 lab variant $v"; tools/lab/lab_agg3 $v 5 & sleep 2; tools/lab/lab_xproc C 2 | head -44; wait
done
