#!/bin/bash
cd "$GRAFT_REPO_ROOT"
X=tools/lab/lab_xproc
for v in 0 15 17; do
  echo "--- victim next to a BARE process running gemm3 lab variant $v"; tools/lab/lab_agg3 $v 6 & sleep 2; $X B 2 | tail -1; wait
done
echo "--- gemm3 (bare) and victim as two streams of ONE process: see xproc_inproc"
