#!/bin/bash
cd "$GRAFT_REPO_ROOT"
for b in lab_xproc lab_xproc_pad1 lab_xproc_pad3 lab_xproc_pad16 lab_xproc_pad64; do
  echo "--- victim binary $b next to a bare process running gemm3 (lab variant 0)"; tools/lab/lab_agg3 0 5 & sleep 2; tools/lab/$b B 1.5 | tail -4; wait
done
