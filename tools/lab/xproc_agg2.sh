#!/bin/bash
# victim (library-free) next to gemm3_kernel with ONE piece of side work left in its k-loop:
# 30 global loads | 29 split arithmetic | 27 LDS stores | 23 barrier | 15 fragment reads | 17 none | 0 all
cd "$GRAFT_REPO_ROOT"
X=tools/lab/lab_xproc
for v in ${VS:-30 29 27 23 15 17 0}; do
  echo "--- victim next to gemm3 lab variant $v"; V=$v SECS=7 python tools/lab/xproc_agg2.py 2>&1 | grep -v amdgpu & sleep 4; $X B 2 | tail -1; wait
done
